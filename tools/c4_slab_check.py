import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from periodicity_amd import _cabi
from oracle import c_oracle as co
import bench
n, nf_total, world = 1_000_000, 10_000_000, 8
t, y, dy = bench.synth_curve(n, k=4)
freq, df, fmin = bench.throughput_grid(t, nf_total)
f0, delta, _ = _cabi.grid_params(freq)
slab = nf_total // world
for rank in (0, 7):
    t0 = time.perf_counter()
    p = _cabi.gls_scan(t, y, dy, f0, delta, slab, j_begin=rank * slab)
    dtm = time.perf_counter() - t0
    rng = np.random.default_rng(rank)
    pick = np.unique(rng.integers(0, slab, 12))
    exact = co.gls_power_exact(t, y, dy, freq[rank * slab + pick])
    rel = np.max(np.abs(p[pick] - exact) / np.abs(exact))
    print(json.dumps({"config": f"C4 slab rank {rank}/8: N=1e6 x 1.25e6", "ms_end_to_end": round(dtm * 1e3, 1),
                      "Gpair_per_s": round(n * slab / dtm / 1e9, 1), "max_rel_vs_exact_12bins": float(rel),
                      "finite": bool(np.all(np.isfinite(p)))}))
