"""Host-API (PCIe-inclusive) timings: numpy arrays in, numpy arrays out, through the ctypes C ABI."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from periodicity_amd import _cabi
from periodicity_amd.core import TSeries
from periodicity_amd.spectral import GLS
from periodicity_amd.phase import PDM, StringLength
import bench

def best(fn, reps=5):
    fn()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); out.append(time.perf_counter() - t0)
    return min(out)

t, y, dy = bench.synth_curve(100_000)
freq, df, fmin = bench.throughput_grid(t, 1_000_000)
f0, delta, nf = _cabi.grid_params(freq)
r = {"C2 pdc_gls_scan (direct) ms": best(lambda: _cabi.gls_scan(t, y, dy, f0, delta, nf)) * 1e3,
     "C2 pdc_gls_scan_fft ms": best(lambda: _cabi.gls_scan_fft(t, y, dy, fmin, df, nf)) * 1e3}
sig = TSeries(t, y)
r["C2 GLS(fmin,fmax) class call ms"] = best(lambda: GLS(fmin=freq[0], fmax=freq[-1] - 0.5 * df)(sig, err=dy)) * 1e3
t5, y5, _ = bench.synth_curve(50_000, k=5, period=13.7)
s5 = TSeries(t5, y5)
r["C5 PDM class call ms"] = best(lambda: PDM(p_min=1.0, p_max=100.0, n_periods=100_000)(s5), 3) * 1e3
r["C5 StringLength class call ms"] = best(lambda: StringLength(n_periods=100_000)(s5), 3) * 1e3
print(json.dumps({k: round(v, 2) for k, v in r.items()}))

# bootstrap false-alarm levels: 1000 replicates of a 2000-sample curve on its default grid
tb, yb, dyb = bench.synth_curve(2000, k=6, period=11.0)
sb = TSeries(tb, yb)
rb = {}
for method in ("direct", "fft"):
    g = GLS(method=method)
    g(sb, err=dyb)
    rb[f"bootstrap(1000) N=2000 nf={g.frequency.size} method={method} ms"] = round(
        best(lambda: g.bootstrap(1000, random_seed=1), 2) * 1e3, 1)
print(json.dumps(rb))
