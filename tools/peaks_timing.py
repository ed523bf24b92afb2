"""Timing of the device-side peak kernels on C3-shaped spectra resident in HBM (developer tool)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from periodicity_amd import _cabi  # noqa: E402

lib = _cabi.lib()
dev = 0
B, n, nf = int(os.environ.get("B", 4096)), 2000, 50_000
rng = np.random.default_rng(20241008 + 3)
tt = np.sort(rng.uniform(0, float(n), (B, n)), axis=1)
dd = rng.uniform(0.05, 0.2, (B, n))
yy = 1.0 + 0.5 * np.sin(2 * np.pi * tt / (5.0 + 0.01 * np.arange(B))[:, None]) + dd * rng.standard_normal((B, n))
offsets = np.arange(B + 1, dtype=np.int64) * n
df = 1.0 / n / 5
f = np.arange(0.5 * df, 0.5 * df + (nf - 1.5) * df + df, df)
g0, gd, _ = _cabi.grid_params(f)
DB = _cabi.DeviceBuffer
bt, by, bdy, boff = (DB.from_array(a, dev) for a in (tt, yy, dd, offsets))
wb = lib.pdc_gls_work_bytes(B * n, B, nf)
work, power = DB(wb, dev), DB(B * nf * 8, dev)
sp = C.c_void_p()
_cabi.check(lib.pdc_stream_create(dev, C.byref(sp)))
tm = bench.EventTimer(lib, _cabi, dev, sp.value)
_cabi.check(lib.pdc_gls_scan_dev(dev, sp.value, bt.ptr, by.ptr, bdy.ptr, boff.ptr, B * n, B, 0, g0, gd, 0, nf, 1, 0,
                                 power.ptr, None, None, work.ptr, wb))
out = DB(B * (1 + 5 * 256) * 8, dev)
print("highest_peak ms", tm.ms(lambda: _cabi.check(lib.pdc_highest_peak_dev(dev, sp.value, power.ptr, B, nf, out.ptr, out.ptr + B * 8))))
for k in (1, 4, 8, 64, 128, 256):
    for bp in (0, 1):
        p = out.ptr
        ms = tm.ms(lambda: _cabi.check(lib.pdc_peaks_topk_dev(dev, sp.value, power.ptr, B, nf, k, bp, p, p + B * 8, p + B * 8 * (1 + 3 * k),
                                                              p + B * 8 * (1 + 4 * k), p + B * 8 * (1 + k), p + B * 8 * (1 + 2 * k))))
        print(f"topk k={k} by_prominence={bp}: {ms:.3f} ms")
cnt = out.to_array(np.int64, B)
print("peaks per spectrum: mean", cnt.mean(), "max", cnt.max())
