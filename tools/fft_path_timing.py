"""Kernel time of the device FFT-extirpolation path at C1, C2 and C4 (developer tool)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from periodicity_amd import _cabi
lib = _cabi.lib()
sp = C.c_void_p(); _cabi.check(lib.pdc_stream_create(0, C.byref(sp)))
tm = bench.EventTimer(lib, _cabi, 0, sp.value)
DB = _cabi.DeviceBuffer
for n, nf, k in ((1000, 1000, 1), (100_000, 1_000_000, 2), (1_000_000, 10_000_000, 4)):
    t, y, dy = bench.synth_curve(n, k)
    f, df, fmin = bench.throughput_grid(t, nf)
    b = [DB.from_array(a, 0) for a in (t, y, dy)]
    wb = lib.pdc_gls_fft_work_bytes(n, nf)
    w, out = DB(wb, 0), DB(nf * 8, 0)
    ms = tm.ms(lambda: _cabi.check(lib.pdc_gls_scan_fft_dev(0, sp.value, b[0].ptr, b[1].ptr, b[2].ptr, n, fmin, df, nf, 1, 0, out.ptr, w.ptr, wb)), reps=5)
    print(f"FFT path N={n} nf={nf}: {ms:.4f} ms")
    for x in b + [w, out]: x.free()
