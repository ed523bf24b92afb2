"""Direct-sum GLS kernel time over (N, nf) shapes, single curves and batches (developer tool).
``SHAPES="1000x1000000,100000x1000" BATCH="4096x200x50000" python tools/gls_shapes.py``"""
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
shapes = os.environ.get("SHAPES", "100x1000000,1000x1000000,10000x1000000,100000x1000000,1000000x100000,1000000x10000,100000x10000,1000x10000")
for spec in shapes.split(","):
    n, nf = (int(v) for v in spec.split("x"))
    t, y, dy = bench.synth_curve(n)
    freq, df, fmin = bench.throughput_grid(t, nf)
    f0, delta, _ = _cabi.grid_params(freq)
    bt, by, bdy, bp = DB.from_array(t, 0), DB.from_array(y, 0), DB.from_array(dy, 0), DB(nf * 8, 0)
    wb = lib.pdc_gls_work_bytes(n, 1, nf)
    w = DB(wb, 0)
    ms = tm.ms(lambda: _cabi.check(lib.pdc_gls_scan_dev(0, sp.value, bt.ptr, by.ptr, bdy.ptr, None, n, 1, 0, f0, delta, 0, nf, 1, 0,
                                                        bp.ptr, None, None, w.ptr, wb)), reps=3)
    print(f"N={n:8d} nf={nf:8d}: {ms:9.3f} ms  {n * nf / ms / 1e6:8.1f} Gpair/s")
    for b in (bt, by, bdy, bp, w):
        b.free()
for spec in os.environ.get("BATCH", "4096x2000x50000,4096x200x50000,65536x100x5000,256x20000x50000").split(","):
    B, n, nf = (int(v) for v in spec.split("x"))
    rng = np.random.default_rng(5)
    tt = np.sort(rng.uniform(0, float(n), (B, n)), axis=1)
    dd = rng.uniform(0.05, 0.2, (B, n))
    yy = 1.0 + 0.5 * np.sin(2 * np.pi * tt / 7.3) + dd * rng.standard_normal((B, n))
    offsets = np.arange(B + 1, dtype=np.int64) * n
    df = 1.0 / n / 5
    f = np.arange(0.5 * df, 0.5 * df + (nf - 1.5) * df + df, df)
    g0, gd, _ = _cabi.grid_params(f)
    bt, by, bdy, boff = (DB.from_array(a, 0) for a in (tt, yy, dd, offsets))
    wb = lib.pdc_gls_work_bytes(B * n, B, nf)
    w, amax, arg = DB(wb, 0), DB(B * 8, 0), DB(B * 8, 0)
    ms = tm.ms(lambda: _cabi.check(lib.pdc_gls_scan_dev(0, sp.value, bt.ptr, by.ptr, bdy.ptr, boff.ptr, B * n, B, 0, g0, gd, 0, nf, 1, 0,
                                                        None, amax.ptr, arg.ptr, w.ptr, wb)), reps=3)
    print(f"batch B={B:6d} N={n:6d} nf={nf:6d} (peaks only): {ms:9.3f} ms  {B * n * nf / ms / 1e6:8.1f} Gpair/s")
    for b in (bt, by, bdy, boff, w, amax, arg):
        b.free()
