"""Randomised parity run of the StringLength kernels against the C oracle: sizes around the fast
path's capacity, clustered / tied / offset / non-finite time stamps, extreme and special periods.
``python tools/fuzz_sl.py --cases 200 [--seed 1]``; exit code 1 on any mismatch.  ``PDC_SL_STREAM_MIN=4096``
sends everything from 4096 samples on through the streamed kernels (default: from 262 144), ``PDC_SL_STREAM=0`` nothing."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import c_oracle as co  # noqa: E402
from oracle import scan_oracle as so  # noqa: E402
from periodicity_amd import _cabi  # noqa: E402


def one_case(rng):
    n = int(rng.choice([1, 2, 5, 64, 255, 256, 257, 1000, 4096, 9000, 20481, 36865, 50000, 52112, 52113, 70000,
                        131071, 208000, 383617],
                       p=[.04, .04, .04, .05, .05, .05, .05, .15, .1, .1, .08, .06, .06, .04, .03, .03, .01, .01, .01]))
    kind = rng.integers(0, 6)
    if kind == 0:
        t = np.sort(rng.uniform(0, float(n), n))
    elif kind == 1:
        t = np.arange(n, dtype=float) * rng.choice([1.0, 0.5, 0.1, 3.0])               # evenly sampled: ties
    elif kind == 2:
        t = np.sort(rng.uniform(0, 30.0, n)) + rng.choice([2454953.5, -1e5, 1e9])       # large offsets
    elif kind == 3:
        t = np.sort(np.round(rng.uniform(0, 400.0, n), 1))                              # repeated stamps
    elif kind == 4:
        t = np.sort(rng.normal(0, 1e-3, n))                                             # tiny, both signs
    else:
        t = np.sort(rng.uniform(0, float(n), n))
        t[rng.integers(0, n, max(1, n // 500))] = rng.choice([np.nan, np.inf, -np.inf])  # non-finite stamps
    y = np.sin(2 * np.pi * np.nan_to_num(t) / rng.uniform(2.0, 50.0)) + 0.3 * rng.standard_normal(n)
    if rng.integers(0, 8) == 0:
        y[rng.integers(0, n)] = np.nan
    m = so.stringlength_scale(y) if np.nanmax(y) > np.nanmin(y) else np.zeros_like(y)
    n_per = int(rng.choice([1, 3, 17, 40]))
    periods = np.concatenate([
        10.0 ** rng.uniform(-3, 7, n_per),
        rng.choice([1.0, 2.0, 0.1, 0.25, 12.5, np.nextafter(2.0, 0.0), np.nextafter(1.0, 0.0), -3.7, 1e-200, 1e200,
                    float(n), 3.0], 3)])
    with np.errstate(all="ignore"):
        want = co.stringlength_scan(t, m, periods) if n > 300 else so.stringlength_scan(t, m, periods)
    got = _cabi.stringlength_scan(t, m, periods)
    again = _cabi.stringlength_scan(t, m, periods)
    ok = np.allclose(got, want, rtol=1e-9, atol=1e-12, equal_nan=True) and np.array_equal(got, again, equal_nan=True)
    return ok, dict(n=n, kind=int(kind), periods=periods, got=got, want=want)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for c in range(args.cases):
        ok, info = one_case(rng)
        if not ok:
            bad += 1
            w = np.argmax(~np.isclose(info["got"], info["want"], rtol=1e-9, atol=1e-12, equal_nan=True))
            print(f"FAIL case {c}: n={info['n']} kind={info['kind']} period={info['periods'][w]!r} "
                  f"got={info['got'][w]!r} want={info['want'][w]!r}")
    print(f"fuzz_sl: {args.cases} cases, {bad} failure(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
