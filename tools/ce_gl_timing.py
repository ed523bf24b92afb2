"""Kernel time of the counts-only phase scans (conditional entropy, Gregory-Loredo) at C5 (developer tool)."""
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
t5, y5, m, periods, _ = bench.c5_inputs()
n, n_per = t5.size, periods.size
mag = np.minimum(np.floor((y5 - y5.min()) / (y5.max() - y5.min()) * 5), 4).astype(np.float64)
bt, bmag, bp, bo = DB.from_array(t5), DB.from_array(mag), DB.from_array(periods), DB(n_per * 8)
ms = tm.ms(lambda: _cabi.check(lib.pdc_cond_entropy_scan_dev(0, sp.value, bt.ptr, bmag.ptr, n, bp.ptr, n_per, 10, 5, bo.ptr)), reps=3)
print(f"cond_entropy 10x5: {ms:.3f} ms  checksum {bo.to_array(np.float64, n_per).sum():.12e}")
for mm, noff in ((2, 8), (6, 8), (12, 8)):
    ms = tm.ms(lambda: _cabi.check(lib.pdc_gl_scan_dev(0, sp.value, bt.ptr, n, bp.ptr, n_per, mm, noff, bo.ptr)), reps=3)
    print(f"gregory_loredo m={mm} offsets={noff}: {ms:.3f} ms  checksum {bo.to_array(np.float64, n_per).sum():.12e}")
