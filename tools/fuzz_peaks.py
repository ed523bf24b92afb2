"""Randomised parity run of pdc_peaks_topk against scipy.signal.find_peaks / peak_prominences and the
host half-maximum lookups: row lengths around the chunk / block sizes, smooth, noisy, quantised (flat
tops, tied heights and prominences), monotone, random-walk, NaN- and inf-holding rows, batches.
``python tools/fuzz_peaks.py --cases 300 [--seed 1]``; exit code 1 on any mismatch."""
import argparse
import os
import sys

import numpy as np
from scipy.signal import find_peaks

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from periodicity_amd import _cabi  # noqa: E402


def ranked(x, k, by_prominence):
    idx, res = find_peaks(x, prominence=0.0)
    prom = res["prominences"]
    key = prom if by_prominence else x[idx]
    order = np.lexsort((idx, -key))[:k]
    pad = k - order.size
    return (idx.size, np.concatenate([idx[order], -np.ones(pad, dtype=np.int64)]),
            np.concatenate([x[idx][order], np.full(pad, np.nan)]),
            np.concatenate([prom[order], np.full(pad, np.nan)]))


def half_max(x, idmax, key):
    half = x[idmax] - key / 2
    with np.errstate(invalid="ignore"):
        left = np.where(np.diff(np.signbit(x[:idmax] - half)))[0]
        right = np.where(np.diff(np.signbit(x[idmax:] - half)))[0]
    return (int(idmax + right[0]) if right.size else -1), (int(left[-1]) if left.size else -1)


def make_row(rng, n):
    kind = rng.integers(0, 9)
    if kind == 0:
        x = rng.standard_normal(n)
    elif kind == 1:
        w = int(rng.choice([3, 9, 31, 101]))
        x = np.convolve(rng.standard_normal(n + w), np.ones(w) / w, mode="same")[:n]
    elif kind == 2:
        x = rng.standard_normal(n).round(int(rng.integers(0, 2)))                # flat tops, ties
    elif kind == 3:
        x = rng.standard_normal(n).cumsum()                                      # long walks
    elif kind == 4:
        saw = np.where(np.arange(n) % 2 == 1, 1.0, -1.0)
        x = saw + rng.choice([1e-3, -1e-3, 0.0]) * np.arange(n)                  # records / ties everywhere
    elif kind == 5:
        x = np.abs(np.sin(np.arange(n) * rng.uniform(0.001, 0.3))) * np.linspace(1, rng.uniform(0.5, 3), n)
    elif kind == 6:
        x = np.tile(rng.integers(0, 4, int(rng.integers(3, 12))).astype(float), n // 3 + 1)[:n]
    elif kind == 7:
        x = rng.standard_normal(n) * (rng.random(n) < 0.02) * 10                 # sparse spikes on a flat floor
    else:
        x = np.resize(np.convolve(rng.standard_normal(n), np.ones(5) / 5, mode="same"), n) + 2 * np.exp(-0.5 * ((np.arange(n) - n / 3) / 40.0) ** 2)
    x = np.resize(np.asarray(x, dtype=float), n).copy()
    if rng.integers(0, 5) == 0 and n > 4:
        x[rng.integers(0, n, max(1, n // 400))] = np.nan
    if rng.integers(0, 8) == 0 and n > 4:
        x[rng.integers(0, n, 2)] = rng.choice([np.inf, -np.inf])
    return x


def one_case(rng):
    n = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 2048, 4097, 10_000, 50_000,
                        262_145, 1_100_000],
                       p=[.02, .02, .02, .03, .03, .03, .03, .04, .04, .04, .17, .05, .05, .05, .05, .09, .12, .09, .02, .01]))
    # (round 6: ranks beyond 16 and beyond one launch of 128 too - the SORT instance's rankings, the quarter-wave
    # half-maximum searches, the seams between launches)
    k = int([rng.integers(1, 17), rng.integers(1, 17), rng.integers(17, 129), rng.integers(129, 400)][int(rng.integers(0, 4))])
    by_prominence = bool(rng.integers(0, 2))
    rows = int(rng.choice([1, 1, 3])) if n <= 50_000 else 1
    x = np.stack([make_row(rng, n) for _ in range(rows)])
    got = _cabi.peaks_topk(x, k=k, by_prominence=by_prominence)
    for b in range(rows):
        count, idx, height, prom = ranked(x[b], k, by_prominence)
        ok = (got["count"][b] == count and np.array_equal(got["indices"][b], idx)
              and np.array_equal(got["heights"][b], height, equal_nan=True)
              and np.array_equal(got["prominences"][b], prom, equal_nan=True))
        if ok:
            for r in range(k):
                key = prom[r] if by_prominence else height[r]
                if idx[r] >= 0 and not np.isfinite(key):
                    continue   # half-maximum level inf - inf/2 = NaN: the crossings depend on the NaN's sign bit
                want = (-1, -1) if idx[r] < 0 else half_max(x[b], idx[r], key)
                ok = ok and (got["half_lo"][b, r], got["half_hi"][b, r]) == want
        if not ok:
            wh = [(-1, -1) if idx[r] < 0 else half_max(x[b], idx[r], prom[r] if by_prominence else height[r]) for r in range(k)]
            return False, dict(n=n, k=k, by_prominence=by_prominence, row=b, want=(count, idx, height, prom, wh),
                               got=(got["count"][b], got["indices"][b], got["heights"][b], got["prominences"][b],
                                    list(zip(got["half_lo"][b].tolist(), got["half_hi"][b].tolist()))))
    return True, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for c in range(args.cases):
        ok, info = one_case(rng)
        if not ok:
            bad += 1
            print(f"case {c}: MISMATCH {info}", flush=True)
    print(f"fuzz_peaks: {args.cases} cases, {bad} failure(s)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
