"""Where a row of peaks_topk_kernel spends its time (developer tool): s_memrealtime stamps of every row of a C3-shaped
batch from a PDC_PK_DBG build -
    tools/ab_build.sh pkdbg "-DPDC_PK_DBG=1" peaks.hip
    PDC_LIBRARY=periodicity_amd/libpdc_ab_pkdbg.so python tools/peaks_stamps.py [k] [by_prominence]
Phases: sweep 1 | ranking | walks | (by prominence: ranking + tau | sweep 2 | walks | ranking) | outputs | crossings."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from periodicity_amd import _cabi  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = _cabi.lib()
B, n, nf = 4096, 2000, 50_000
# the C3 batch of tools/peaks_timing.py: GLS spectra of 4096 noisy sinusoids (~6300 maxima a row)
rng = np.random.default_rng(20241008 + 3)
tt = np.sort(rng.uniform(0, float(n), (B, n)), axis=1)
dd = rng.uniform(0.05, 0.2, (B, n))
yy = 1.0 + 0.5 * np.sin(2 * np.pi * tt / (5.0 + 0.01 * np.arange(B))[:, None]) + dd * rng.standard_normal((B, n))
df = 1.0 / n / 5
fgrid = np.arange(0.5 * df, 0.5 * df + (nf - 1.5) * df + df, df)
g0, gd, _ = _cabi.grid_params(fgrid)
x = _cabi.gls_scan_batch(tt.ravel(), yy.ravel(), dd.ravel(), np.arange(B + 1, dtype=np.int64) * n, g0, gd, nf)[0]
DB = _cabi.DeviceBuffer
bx, out = DB.from_array(x, 0), DB(B * (1 + 5 * k) * 8, 0)
p = out.ptr
for _ in range(3):
    _cabi.check(lib.pdc_peaks_topk_dev(0, None, bx.ptr, B, nf, k, bp, p, p + B * 8, p + B * 8 * (1 + 3 * k), p + B * 8 * (1 + 4 * k),
                                       p + B * 8 * (1 + k), p + B * 8 * (1 + 2 * k)))
    _cabi.check(lib.pdc_device_sync(0))
raw = (C.c_ulonglong * (8192 * 16))()
C.CDLL(os.environ["PDC_LIBRARY"]).pdc_debug_peaks_stamps(raw)
st = np.frombuffer(raw, dtype=np.uint64).reshape(8192, 16)[:B].astype(np.int64)
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9] if bp else [0, 1, 2, 3, 8, 9]
names = (["sweep 1", "ranking", "walks", "ranking + tau", "sweep 2", "walks", "ranking", "outputs", "crossings"] if bp
         else ["sweep 1", "ranking", "walks", "outputs", "crossings"])
t0 = st[:, 0].min()
print(f"k={k} by_prominence={bp}: kernel {(st[:, 9].max() - t0) / 100:.1f} us over {B} rows")
d = np.diff(st[:, idx], axis=1) / 100.0
for n_, col in zip(names, d.T):
    print(f"  {n_:14s} mean {col.mean():7.2f} us   median {np.median(col):7.2f}   p90 {np.percentile(col, 90):7.2f}")
print(f"  row total      mean {(st[:, 9] - st[:, 0]).mean() / 100:7.2f} us")
starts = np.sort(st[:, 0] - t0) / 100.0
print("  row starts (us) at ranks 0, 1535, 1536, 3071, 3072, 4095:", [float(starts[i]) for i in (0, 1535, 1536, 3071, 3072, 4095)])
