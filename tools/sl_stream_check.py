"""The streamed StringLength kernels (sl_part / sl_sort / sl_link) against the C oracle and against the
other kernels, at sizes chosen on the command line; PDC_SL_STREAM_MIN=<n> routes smaller curves through
them (developer tool; tests/test_phase_gpu.py runs it in a child process).

    PDC_SL_STREAM_MIN=4096 python tools/sl_stream_check.py 70000x64 20000x32e
      NxP: N unevenly sampled points x P periods; suffix e = evenly sampled (clustered phases at commensurate
      periods: the bins overflow and the general kernel takes the period); suffix d = duplicate time stamps, two gaps
      of many periods and a negative start (slices mode: empty cells, cycles skipped); suffix u = samples in random
      order (slices mode must step aside: the lists are built whatever the order).
    SL_CHECK_SAVE=file.npz keeps every result (tests compare PDC_SL_SLICES=0 against the default bit for bit)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as co  # noqa: E402
from periodicity_amd import _cabi  # noqa: E402

worst = 0.0
kept = {}
for spec in sys.argv[1:]:
    even, dups, shuffled = spec.endswith("e"), spec.endswith("d"), spec.endswith("u")
    n, n_per = (int(float(v)) for v in spec.rstrip("edu").split("x"))
    rng = np.random.default_rng(n + n_per)
    t = np.arange(float(n)) if even else np.sort(rng.uniform(0, float(n), n))
    if dups:
        t[n // 3:] += 0.31 * n                              # a gap of many periods ...
        t[2 * n // 3:] += 0.07 * n                          # ... and another
        t[5:n:7] = t[4:n - 1:7]                             # duplicate time stamps
        t -= 0.4 * n                                        # a negative start
    y = np.sin(2 * np.pi * t / 13.7) + 0.2 * rng.standard_normal(n)
    m = (y - y.max()) / (2 * (y.max() - y.min())) + 0.25
    df = 0.1 / (t[-1] - t[0])
    periods = 1 / np.linspace(n_per * df * 40, df, n_per)
    if even:
        periods[::3] = np.round(periods[::3])          # exactly commensurate: a handful of distinct phases
        periods[1::7] = 10.0 + 1e-9 * np.arange(periods[1::7].size)
    if shuffled:
        order = rng.permutation(n)
        t, m = t[order], m[order]
    got = _cabi.stringlength_scan(t, m, periods)
    again = _cabi.stringlength_scan(t, m, periods)
    pick = np.unique(np.concatenate([[0, n_per - 1], rng.integers(0, n_per, min(n_per, 24))]))
    co.set_threads(os.cpu_count() or 1)
    want = co.stringlength_scan(t, m, periods[pick])
    rel = np.max(np.abs(got[pick] - want) / np.abs(want))
    worst = max(worst, rel)
    print(f"{spec}: max rel err vs oracle {rel:.2e} over {pick.size} periods; bitwise repeatable: "
          f"{np.array_equal(got, again)}; finite: {bool(np.all(np.isfinite(got)))}")
    assert rel <= 1e-9 and np.array_equal(got, again)
    kept[spec] = got
    q0, q1 = t.min() / periods, t.max() / periods                   # (summed as the samples stand: less than one cycle)

    kept[spec + ":one_cycle"] = (periods > 0) & ((np.floor(q1) == np.floor(q0)) | ((np.floor(q1) - np.floor(q0) == 1) & (q1 % 1 < q0 % 1)))
if os.environ.get("SL_CHECK_SAVE"):
    np.savez(os.environ["SL_CHECK_SAVE"], **kept)
print("ok", worst)
