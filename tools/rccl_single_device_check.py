import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from periodicity_amd import _cabi
rng = np.random.default_rng(0)
t = np.sort(rng.uniform(0, 500, 500)); dy = rng.uniform(.05,.2,500); y = np.sin(t/3) + dy*rng.standard_normal(500)
f0, delta, nf = 0.001, 0.0007, 3001
a = _cabi.gls_scan(t, y, dy, f0, delta, nf)
b = _cabi.gls_scan_multi(t, y, dy, f0, delta, nf, devices=(0,))
print("forced RCCL single-device all-gather equal:", np.array_equal(a, b))
