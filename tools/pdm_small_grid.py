"""PDM / StringLength kernel time at the reference's DEFAULT grid size (1000 trial periods)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from periodicity_amd import _cabi
rng = np.random.default_rng(2)
for n in (2000, 50_000, 1_000_000):
    t = np.sort(rng.uniform(0, float(n), n)); x = np.sin(t / 5.0) + 0.1 * rng.standard_normal(n)
    periods = np.linspace(1.0, 100.0, 1000)
    sig = np.var(x, ddof=1)
    out = {}
    for name, fn in (("PDM", lambda: _cabi.pdm_scan(t, x, periods, 5, 2, sig)), ("SL", lambda: _cabi.stringlength_scan(t, x, periods))):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        out[name + "_call_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    print(json.dumps({"N": n, "periods": 1000, **out}))
