"""The phase scans the way upstream runs them: ``multiprocessing.Pool(cores).map`` of the per-period
function over the trial periods (/root/reference/src/periodicity/phase.py:69-70,185-186), with the
numpy restatement of ``_pdm`` / ``_stringlength`` (oracle/scan_oracle.py) as the worker, on a subsample
of the C5 period grid.  bench.py starts this as a CHILD process (it never touches the GPU and forks
its workers itself) and scales the result linearly; SURVEY.md 8d (iii).

    python tools/cpu_pool_baseline.py [n_periods_sampled] [cores]  ->  one JSON line
    python tools/cpu_pool_baseline.py defaults N [cores]           ->  the classes' DEFAULT calls at N samples
        (StringLength(): dphi = 0.1, 1000 periods, phase.py:38,67-68; PDM(): nb = 5, nc = 2, 1000 periods between
        2 median dt and the baseline, phase.py:108-118,167-180), whole grids, one Pool per call as upstream
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import scan_oracle as so  # noqa: E402


class PdmWorker:
    """Pickled to the workers chunk by chunk like upstream's bound method (which carries t, x)."""

    def __init__(self, t, x, nb, nc):
        self.t, self.x, self.nb, self.nc = t, x, nb, nc
        self.sigma = np.var(x, ddof=1)

    def __call__(self, period):
        return so.pdm_theta(self.t, self.x, period, self.nb, self.nc, self.sigma)


class StringWorker:
    def __init__(self, t, m):
        self.t, self.m = t, m

    def __call__(self, period):
        return so.stringlength_one(self.t, self.m, period)


def defaults_main():
    n = int(sys.argv[2])
    cores = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
    rng = np.random.default_rng(20241008 + 5)                 # bench.synth_curve(n, 5, 13.7)
    t = np.sort(rng.uniform(0, float(n), n))
    dy = rng.uniform(0.05, 0.2, n)
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / 13.7) + dy * rng.standard_normal(n)
    m = so.stringlength_scale(y)
    grids = {"pdm": so.pdm_periods(t)[0], "stringlength": so.stringlength_periods(t[-1] - t[0])}
    workers = {"pdm": PdmWorker(t, y, 5, 2), "stringlength": StringWorker(t, m)}
    out = {"cores": cores, "n_samples": n, "start_method": mp.get_start_method()}
    for name in ("pdm", "stringlength"):
        t0 = time.perf_counter()
        with mp.Pool(cores) as pool:
            values = pool.map(workers[name], grids[name])
        wall = time.perf_counter() - t0
        assert len(values) == grids[name].size
        out[name] = {"wall_s": round(wall, 3), "n_periods": int(grids[name].size)}
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "defaults":
        return defaults_main()
    sub = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cores = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    n, n_full = 50_000, 100_000
    rng = np.random.default_rng(20241008 + 5)                 # bench.synth_curve(n, 5, 13.7)
    t = np.sort(rng.uniform(0, float(n), n))
    dy = rng.uniform(0.05, 0.2, n)
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / 13.7) + dy * rng.standard_normal(n)
    m = so.stringlength_scale(y)
    pick = np.linspace(0, n_full - 1, sub).astype(int)
    grids = {"pdm": np.linspace(1.0, 100.0, n_full)[pick],
             "stringlength": so.stringlength_periods(t[-1] - t[0], 0.1, n_full)[pick]}
    workers = {"pdm": PdmWorker(t, y, 5, 2), "stringlength": StringWorker(t, m)}
    out = {"cores": cores, "periods_sampled": sub, "start_method": mp.get_start_method()}
    for name in ("pdm", "stringlength"):
        t0 = time.perf_counter()
        with mp.Pool(cores) as pool:                          # as phase.py:69,185: the pool lives for one call
            t1 = time.perf_counter()
            values = pool.map(workers[name], grids[name])
            t2 = time.perf_counter()
        t3 = time.perf_counter()
        assert len(values) == sub and np.all(np.isfinite(values))
        out[name] = {"wall_s": round(t3 - t0, 3), "map_s": round(t2 - t1, 3), "pool_start_s": round(t1 - t0, 3),
                     "full_grid_s_scaled": round((t2 - t1) * n_full / sub + (t1 - t0) + (t3 - t2), 2),
                     "Gpair_per_s": round(n * sub / (t2 - t1) / 1e9, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
