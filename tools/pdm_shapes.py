"""PDM kernel time over a few (N, n_periods) shapes (developer tool)."""
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
shapes = os.environ.get("SHAPES", "100x1000000,1000x100000,2000x1000,50000x100000,50000x1000,1000000x1000,1000000x64,200000x20000")
NB, NC = int(os.environ.get("NB", "5")), int(os.environ.get("NC", "2"))
for spec in shapes.split(","):
    n, n_per = (int(v) for v in spec.split("x"))
    t, y, _ = bench.synth_curve(n, 5, period=13.7)
    periods = np.linspace(1.0, 100.0, n_per)
    bt, bx, bp, bo = DB.from_array(t, 0), DB.from_array(y, 0), DB.from_array(periods, 0), DB(n_per * 8, 0)
    sigma = float(np.var(y, ddof=1))
    ms = tm.ms(lambda: _cabi.check(lib.pdc_pdm_scan_dev(0, sp.value, bt.ptr, bx.ptr, n, bp.ptr, n_per, NB, NC, sigma, bo.ptr)), reps=3)
    print(f"N={n:8d} periods={n_per:8d}: {ms:9.3f} ms  {n * n_per / ms / 1e6:8.1f} Gpair/s")
    for b in (bt, bx, bp, bo):
        b.free()
