"""Supersmoother scan kernel time over a few shapes (developer tool)."""
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
SHAPES = ((2000, 10_000), (50_000, 4096), (74_326, 2048), (1_000_000, 96))
if os.environ.get("SHAPES"):
    SHAPES = tuple(tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(","))
for n, n_per in SHAPES:
    t, y, _ = bench.synth_curve(n, 5, period=13.7)
    periods = np.linspace(1.0, 100.0, n_per)
    bt, by, bp, be = DB.from_array(t, 0), DB.from_array(y, 0), DB.from_array(periods, 0), DB(n_per * 8, 0)
    wb = lib.pdc_supersmoother_work_bytes(n, n_per)
    w = DB(wb, 0)
    ms = tm.ms(lambda: _cabi.check(lib.pdc_supersmoother_scan_dev(0, sp.value, bt.ptr, by.ptr, n, bp.ptr, n_per, 0.0, be.ptr,
                                                                  w.ptr, wb)), reps=3)
    st = be.to_array(np.float64, n_per)
    print(f"N={n:7d} periods={n_per:6d}: {ms:9.3f} ms  {n * n_per / ms / 1e6:7.2f} Gpair/s  workspace {wb / 1e6:8.1f} MB  "
          f"argmin period {periods[int(np.argmin(st))]:.3f}")
    for b in (bt, by, bp, be, w):
        b.free()
