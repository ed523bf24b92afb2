import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from periodicity_amd import _cabi
from oracle import c_oracle as co, scan_oracle as so
rng = np.random.default_rng(1)
n, n_per = int(sys.argv[1]), int(sys.argv[2])
t = np.sort(rng.uniform(0, float(n), n))
y = np.sin(2 * np.pi * t / 13.7) + 0.1 * rng.standard_normal(n)
m = so.stringlength_scale(y)
periods = np.linspace(1.0, 100.0, n_per)
sig = np.var(y, ddof=1)
a = _cabi.pdm_scan(t, y, periods, 5, 2, sig)
_cabi.stringlength_scan(t, m, periods[:8])
b = _cabi.pdm_scan(t, y, periods, 5, 2, sig)
_cabi.stringlength_scan(t, m, periods)
c = _cabi.pdm_scan(t, y, periods, 5, 2, sig)
d = _cabi.pdm_scan(t, y, periods, 5, 2, sig)
print("b==a", np.array_equal(a, b), "c==a", np.array_equal(a, c), "d==a", np.array_equal(a, d))
for name, v in (("b", b), ("c", c), ("d", d)):
    bad = np.where(v != a)[0]
    if bad.size: print(name, "differs at", bad[:10], bad.size, "max rel", np.max(np.abs(v[bad] - a[bad]) / a[bad]))
pick = np.array([0, n_per // 3, n_per - 1])
print("vs oracle", np.max(np.abs(a[pick] - co.pdm_scan(t, y, periods[pick], 5, 2)) / a[pick]))
