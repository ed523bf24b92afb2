"""Long randomised parity run on a GPU box: the seeded sweeps of tests/test_random_gpu.py with
fresh seeds.  ``python tools/fuzz_gpu.py --seeds 20 [--start 1000]`` prints one line per failure
(seed + test) and a summary; exit code 1 if anything failed."""
import argparse
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tests import test_random_gpu as sweeps  # noqa: E402

CASES = [sweeps.test_gls_direct_random_cases, sweeps.test_gls_long_curves_on_short_grids_random,
         sweeps.test_gls_batch_random_ragged,
         sweeps.test_gls_fft_random_cases, sweeps.test_pdm_random_cases, sweeps.test_binned_scans_random_cases,
         sweeps.test_stringlength_random_cases, sweeps.test_gls_shared_time_axis_random,
         sweeps.test_supersmoother_random_cases, sweeps.test_streamed_samples_in_any_order_random,
         sweeps.test_bglst_random_cases]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=10)
    ap.add_argument("--start", type=int, default=1000)
    args = ap.parse_args()
    failures = 0
    for seed in range(args.start, args.start + args.seeds):
        for case in CASES:
            try:
                case(seed=seed)
            except Exception:  # report and keep going: the point is to collect every failing seed
                failures += 1
                print(f"FAIL seed={seed} {case.__name__}")
                traceback.print_exc(limit=3)
    print(f"fuzz: {args.seeds} seeds x {len(CASES)} sweeps, {failures} failure(s)")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
