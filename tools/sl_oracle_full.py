"""StringLength through the C ABI against the C oracle on EVERY period of the reference's grid, at the sizes the
several-slice and streamed kernels serve (developer tool; tests/test_phase_gpu.py runs it in child processes because the
library reads PDC_SL_SLICES / PDC_SL_STREAM_GROUPS once per process).

    python tools/sl_oracle_full.py 300000x4096 1000000x384 262144x384d 300000x320o
      NxP: N unevenly sampled points x the P periods of StringLength's own grid (phase.py:67-68, dphi = 0.1);
      suffix d = duplicate time stamps, two gaps of many periods, a negative start; suffix o = Julian-date offset
      (t + 2454953.5: a cycle boundary inside the samples for the longest periods); suffix s = a grid of SHORT periods
      (40x the frequencies: cells under four samples -> the lists mode even for time-ordered samples); suffix u = the
      samples handed over in a random order (the device orders them by time first - csrc/timesort.inc; the oracle gets
      what TSeries would make of them: a stable sort by time); suffix z = with u: a few time stamps are -0.0 / +0.0 and
      the times straddle zero.
    SL_FULL_DEV=1: through pdc_stringlength_scan_dev (the host has not looked at the samples: the time sort's launches
    are enqueued and return at once when the samples are in order).
The oracle sorts every period on all host threads (oracle/scan_oracle.c: OpenMP over periods)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as co  # noqa: E402
from periodicity_amd import _cabi  # noqa: E402

co.set_threads(os.cpu_count() or 1)


def scan(t, m, periods):
    if not os.environ.get("SL_FULL_DEV"):
        return _cabi.stringlength_scan(t, m, periods)
    lib, DB = _cabi.lib(), _cabi.DeviceBuffer
    wb = lib.pdc_stringlength_work_bytes(t.size, periods.size)
    bufs = [DB.from_array(t, 0), DB.from_array(m, 0), DB.from_array(periods, 0), DB(periods.size * 8, 0), DB(wb, 0)]
    _cabi.check(lib.pdc_stringlength_scan_dev(0, None, bufs[0].ptr, bufs[1].ptr, t.size, bufs[2].ptr, periods.size,
                                              bufs[3].ptr, bufs[4].ptr, wb))
    _cabi.check(lib.pdc_device_sync(0))
    out = bufs[3].to_array(np.float64, periods.size)
    for b in bufs:
        b.free()
    return out


worst = 0.0
for spec in sys.argv[1:]:
    flags = spec.lstrip("0123456789x")
    n, n_per = (int(v) for v in spec[:len(spec) - len(flags)].split("x"))
    rng = np.random.default_rng(n + 3 * n_per)
    t = np.sort(rng.uniform(0, float(n), n))
    if "d" in flags:
        t[n // 3:] += 0.31 * n
        t[2 * n // 3:] += 0.07 * n
        t[5:n:7] = t[4:n - 1:7]
        t -= 0.4 * n
    if "o" in flags:
        t += 2454953.5
    y = np.sin(2 * np.pi * t / 13.7) + 0.2 * rng.standard_normal(n)
    m = (y - y.max()) / (2 * (y.max() - y.min())) + 0.25
    df = 0.1 / (t[-1] - t[0])
    periods = 1 / np.linspace(n_per * df * (40 if "s" in flags else 1), df, n_per)
    gt, gm = t, m
    if "u" in flags:
        if "z" in flags:
            t = t - t[n // 2]
            t[n // 2 - 2:n // 2 + 3] = [-0.0, 0.0, -0.0, 0.0, 0.0]
        order = rng.permutation(n)
        gt, gm = t[order], m[order]
        back = np.argsort(gt, kind="stable")
        t, m = gt[back], gm[back]
    got = scan(gt, gm, periods)
    again = scan(gt, gm, periods)
    t0 = time.time()
    want = co.stringlength_scan(t, m, periods)
    rel = np.abs(got - want) / np.abs(want)
    worst = max(worst, float(rel.max()))
    print(f"{spec}: ALL {n_per} periods, max rel err vs oracle {rel.max():.2e} (period #{int(rel.argmax())}), "
          f"{int((rel > 1e-9).sum())} over 1e-9; bitwise repeatable: {np.array_equal(got, again)}; "
          f"oracle {time.time() - t0:.1f} s", flush=True)
    assert rel.max() <= 1e-9 and np.array_equal(got, again), spec
    assert int(np.argmin(got)) == int(np.argmin(want)), spec
print("ok", worst)
