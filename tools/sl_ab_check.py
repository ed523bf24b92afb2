"""StringLength over a mix of shapes (random, clustered, tied, negative and NaN times), results saved to an
.npz; tests/test_phase_gpu.py runs it in child processes with PDC_SL_DUO=1 / 0 (the switch is read once per
process) and compares the two kernels."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import scan_oracle as so  # noqa: E402
from periodicity_amd import _cabi  # noqa: E402

rng = np.random.default_rng(2024)
out = {}
for n, n_per in ((1, 5), (2, 5), (63, 40), (500, 300), (4096, 200), (4097, 130), (12_345, 700), (26_048, 300), (26_049, 64)):
    t = np.sort(rng.uniform(0, float(n), n)) - 0.3 * n
    y = np.sin(2 * np.pi * t / 13.7) + 0.1 * rng.standard_normal(n)
    m = so.stringlength_scale(y) if n > 1 else np.zeros(n)
    periods = np.concatenate([np.linspace(0.7, 0.4 * n + 2.0, n_per), [13.7, 27.4, 1e-3, 1e6]])
    out[f"random_{n}"] = _cabi.stringlength_scan(t, m, periods)
    out[f"random_{n}_again"] = _cabi.stringlength_scan(t, m, periods)
te = np.arange(20_000.0)                                   # clusters: deferred ranges of every size, ties
me = so.stringlength_scale(np.sin(2 * np.pi * te / 12.5) + 0.05 * np.cos(0.37 * te))
pe = np.array([1.0, 2.0, 2.5, 4.0, 12.5, 3.0000000000000004, 7.3, 20000.0, 1e-3, 0.3, 1 / 3, 100.0, 128.0])
out["clustered"] = _cabi.stringlength_scan(te, me, pe)
ti = np.repeat(np.arange(3000.0), 3)[:8000]               # every time stamp three times: exact ties everywhere
mi = so.stringlength_scale(np.cos(ti / 7.0) + 0.01 * np.arange(ti.size) % 5)
out["tied"] = _cabi.stringlength_scan(ti, mi, np.linspace(0.9, 300.0, 257))
tn = np.sort(rng.uniform(0, 5000.0, 5000))
tn[[7, 1000, 4999]] = np.nan                                # NaN phases sort last, in time order
mn = so.stringlength_scale(np.sin(tn / 3.0) + 0.2)
mn[np.isnan(mn)] = 0.0
out["nan_times"] = _cabi.stringlength_scan(tn, mn, np.linspace(0.9, 300.0, 100))
np.savez(sys.argv[1], **out)
