"""StringLength kernel time over a few (N, n_periods) shapes (developer tool; PDC_SL_GENERAL=1 for the
general kernel; SHUFFLE=1: the samples in a random order - the device orders them by time first, timesort.inc)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from periodicity_amd import _cabi  # noqa: E402

lib = _cabi.lib()
sp = C.c_void_p()
_cabi.check(lib.pdc_stream_create(0, C.byref(sp)))
tm = bench.EventTimer(lib, _cabi, 0, sp.value)
DB = _cabi.DeviceBuffer
SHAPES = ((500, 100_000), (2000, 100_000), (2000, 1000), (10_000, 20_000), (25_000, 100_000), (50_000, 100_000))
if os.environ.get("LARGE"):
    SHAPES = ((74_326, 20_000), (200_000, 2048), (200_000, 20_000), (400_000, 2048), (1_000_000, 256), (1_000_000, 2048))
if os.environ.get("SHAPES"):      # SHAPES="300000x2048,600000x1024"
    SHAPES = tuple(tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(","))
for n, n_per in SHAPES:
    t, y, _ = bench.synth_curve(n, 5, period=13.7)
    m = (y - y.max()) / (2 * (y.max() - y.min())) + 0.25
    df = 0.1 / (t[-1] - t[0])
    if os.environ.get("SHUFFLE"):
        order = np.random.default_rng(n).permutation(n)
        t, m = t[order], m[order]
    periods = 1 / np.linspace(n_per * df, df, n_per)
    bt, bm, bp, be = DB.from_array(t, 0), DB.from_array(m, 0), DB.from_array(periods, 0), DB(n_per * 8, 0)
    wb = lib.pdc_stringlength_work_bytes(n, n_per)
    w = DB(wb, 0)
    ms = tm.ms(lambda: _cabi.check(lib.pdc_stringlength_scan_dev(0, sp.value, bt.ptr, bm.ptr, n, bp.ptr, n_per, be.ptr,
                                                                 w.ptr, wb)), reps=3)
    ell = be.to_array(np.float64, n_per)
    print(f"N={n:6d} periods={n_per:6d}: {ms:8.3f} ms  {n * n_per / ms / 1e6:7.1f} Gpair/s   sum(ell)={ell.sum():.12e} min at {int(np.argmin(ell))}")
    for b in (bt, bm, bp, be, w):
        b.free()
