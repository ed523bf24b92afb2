"""Runs the PDM / AoV scans over a sequence of shapes that change between back-to-back calls and saves
every result; tests/test_multi_gpu.py runs it twice - PDC_PDM_POISON=1 (split mode, scratch poisoned
before every call) and PDC_PDM_SPLIT=0 (unsplit kernel) - and compares."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from periodicity_amd import _cabi  # noqa: E402

rng = np.random.default_rng(77)
n_max = 120_000
t_all = np.sort(rng.uniform(0, 5000.0, n_max)) - 100.0
x_all = np.sin(2 * np.pi * t_all / 6.3) + 0.3 * rng.standard_normal(n_max)
out = {}
# (n, n_periods, nb, nc): grows, shrinks, grows again; device lists exercise the per-slot workspaces too
shapes = [(50_001, 1000, 5, 2), (120_000, 64, 5, 2), (4_100, 1, 5, 2), (120_000, 2900, 10, 3), (9_999, 7, 10, 3),
          (60_000, 300, 4, 1), (120_000, 64, 5, 2), (20_000, 130, 4, 1)]
for i, (n, n_per, nb, nc) in enumerate(shapes):
    t, x = t_all[:n], x_all[:n]
    periods = np.linspace(0.9, 40.0, n_per)
    sigma = float(np.var(x, ddof=1))
    out[f"pdm{i}"] = _cabi.pdm_scan(t, x, periods, nb, nc, sigma)
    out[f"aov{i}"] = _cabi.aov_scan(t, x, periods, nb)
    if i % 3 == 0:
        out[f"pdm_multi{i}"] = _cabi.pdm_scan(t, x, periods, nb, nc, sigma, devices=(0, 0, 0))
np.savez(sys.argv[1], **out)
