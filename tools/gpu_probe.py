"""Ad-hoc first contact with the GPU: parity at small size + a timing at C2 scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from periodicity_amd import _cabi
from oracle import scan_oracle as so

print(_cabi.device_count(), _cabi.device_info(0))

def synth(N, k, P0=37.3):
    rng = np.random.default_rng(20241008 + k)
    T = float(N)
    t = np.sort(rng.uniform(0, T, N)); dy = rng.uniform(0.05, 0.2, N)
    y = 1.0 + 0.5*np.sin(2*np.pi*t/P0) + dy*rng.standard_normal(N)
    return t, y, dy

def grid(t, nf):
    df = 1/(t[-1]-t[0])/5; fmin = 0.5*df; fmax = fmin+(nf-1.5)*df
    f = np.arange(fmin, fmax+df, df); assert f.size == nf, f.size
    return f, df, fmin

t, y, dy = synth(1000, 1); f, df, fmin = grid(t, 1000)
f0, delta, nf = _cabi.grid_params(f)
for fm in (True, False):
    for psd in (False, True):
        p = _cabi.gls_scan(t, y, dy, f0, delta, nf, fm, psd)
        pe = so.gls_power(t, y, dy, f, df, fmin, fm, psd, sums="exact")
        pr = so.gls_power(t, y, dy, f, df, fmin, fm, psd, sums="fft")
        print("C1", fm, psd, "max rel vs exact", np.max(np.abs(p-pe)/np.abs(pe)), "argmax", p.argmax(), pe.argmax(), pr.argmax())
S, C = _cabi.trig_sums(t+2.45e6, dy**-2, f0, delta, nf)
Se, Ce = so.trig_sum_exact(t+2.45e6, dy**-2, f)
print("trig_sums offset: max abs err / scale", np.max(np.abs(S-Se))/np.max(np.abs(Se)), np.max(np.abs(C-Ce))/np.max(np.abs(Ce)))

for N, nf in ((10000, 100000), (100000, 1000000)):
    t, y, dy = synth(N, 2); f, df, fmin = grid(t, nf); f0, delta, nf = _cabi.grid_params(f)
    for rep in range(3):
        t0 = time.perf_counter(); p = _cabi.gls_scan(t, y, dy, f0, delta, nf); dtm = time.perf_counter()-t0
        print(f"N={N} nf={nf} K={os.environ.get('PDC_GLS_K','8')}: {dtm*1e3:.2f} ms end-to-end -> {N*nf/dtm/1e9:.1f} Gpair/s; argmax {p.argmax()} peak P={1/f[p.argmax()]:.4f}")
