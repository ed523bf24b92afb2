"""CPU baselines for the phase scans on this machine's host cores: the numpy
restatement of the reference (oracle/scan_oracle.py, one core, exactly upstream's per-period work)
and the C restatement under OpenMP (all cores), on a subsample of the C5 period grid, scaled
linearly to the full 1e5 periods.  Reported next to the GPU numbers in DESIGN.md; never a target."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle as co  # noqa: E402
from oracle import scan_oracle as so  # noqa: E402

rng = np.random.default_rng(20241008 + 5)
n, n_full = 50_000, 100_000
t = np.sort(rng.uniform(0, float(n), n))
y = 1.0 + 0.5 * np.sin(2 * np.pi * t / 13.7) + 0.1 * rng.standard_normal(n)
m = so.stringlength_scale(y)
cores = len(os.sched_getaffinity(0))
co.set_threads(cores)
out = {"cores": cores, "n_samples": n}
for name, np_fn, c_fn, sub_np, sub_c in (
        ("pdm", lambda p: so.pdm_scan(t, y, p, 5, 2), lambda p: co.pdm_scan(t, y, p, 5, 2), 300, 4 * cores),
        ("stringlength", lambda p: so.stringlength_scan(t, m, p), lambda p: co.stringlength_scan(t, m, p), 200, 4 * cores)):
    periods = np.linspace(1.0, 100.0, n_full)
    t0 = time.perf_counter(); np_fn(periods[:: n_full // sub_np][:sub_np]); dt = time.perf_counter() - t0
    out[name + "_numpy_1core"] = {"ms_per_period": round(dt / sub_np * 1e3, 3),
                                  "full_C5_core_seconds": round(dt / sub_np * n_full, 1),
                                  "sample": f"{sub_np} of {n_full} periods"}
    c_fn(periods[:cores])  # warm OpenMP
    t0 = time.perf_counter(); c_fn(periods[:: n_full // sub_c][:sub_c]); dt = time.perf_counter() - t0
    out[name + "_c_openmp_all_cores"] = {"full_C5_seconds": round(dt / sub_c * n_full, 2),
                                         "Gpair_per_s": round(n * sub_c / dt / 1e9, 3),
                                         "sample": f"{sub_c} of {n_full} periods"}
print(json.dumps(out))
