"""GLS.bootstrap end to end (numpy in, maxima out), by index (pdc_gls_bootstrap: the curve + int32 picks are
uploaded, the device gathers) against the round-3 way (values[picks], err[picks] built on the host and
shipped through pdc_gls_scan_batch).  Developer tool: python tools/bootstrap_e2e.py [N] [B]"""
import os
import sys
import time
import tracemalloc

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from periodicity_amd import _cabi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t, y, dy = bench.synth_curve(n, 7)
df = 1.0 / (t[-1] - t[0]) / 5
freq = np.arange(0.5 * df, 0.5 / np.median(np.diff(t)) + df, df)
f0, delta, nf = _cabi.grid_params(freq)
rng = np.random.default_rng(1)
picks = np.empty((B, n), dtype=np.int32)
for i in range(B):
    picks[i] = rng.integers(0, n, n)
offsets = np.arange(B + 1, dtype=np.int64) * n


def by_index():
    return _cabi.gls_bootstrap(t, y, dy, picks, f0, delta, nf)[0]


def expanded():
    return _cabi.gls_scan_batch(t, y[picks].ravel(), dy[picks].ravel(), offsets, f0, delta, nf, shared_t=True,
                                want_power=False, want_peaks=True)[1]


out = {}
for name, fn in (("by_index", by_index), ("expanded_on_host", expanded)):
    fn()                                             # sizes the cached device buffers
    tracemalloc.start()
    t0 = time.perf_counter()
    res = fn()
    dt = time.perf_counter() - t0
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    out[name] = res
    print(f"{name:18s} N={n} B={B} nf={nf}: {dt * 1e3:9.1f} ms end to end, host allocations peak {peak / 1e6:8.1f} MB, "
          f"H2D {(3 * n * 8 + B * n * 4 if name == 'by_index' else n * 8 + 2 * B * n * 8) / 1e6:8.1f} MB")
print("identical maxima:", np.array_equal(out["by_index"], out["expanded_on_host"]))
