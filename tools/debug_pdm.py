import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from periodicity_amd import _cabi
from oracle import scan_oracle as so
from test_random_gpu import random_curve
rng = np.random.default_rng(17)
t, y, dy = random_curve(rng, 5000)
t = np.sort(rng.uniform(0, 5000.0, 5000))
m = so.stringlength_scale(y)
periods = np.linspace(0.8, 90.0, 700)
sig = np.var(y, ddof=1)
print("y mean", y.mean(), "finite", np.isfinite(y).all(), np.isfinite(t).all())
seq = sys.argv[1] if len(sys.argv) > 1 else "psgpsgp"
ref = None
for c in seq:
    if c == "p":
        r = _cabi.pdm_scan(t, y, periods, 5, 2, sig)
        if ref is None: ref = r
        print("pdm", r[:3], "same" if np.array_equal(r, ref) else "DIFFERENT")
    elif c == "s":
        r = _cabi.stringlength_scan(t, m, periods); print("sl", r[:2])
    elif c == "S":
        r = _cabi.stringlength_scan(t, m, periods[:5]); print("sl5", r[:2])
    elif c == "g":
        r = _cabi.gls_scan(t, y, dy, 0.0004, 0.00013, 20000); print("gls", r[:2])
    elif c == "r":
        _cabi.check(_cabi.lib().pdc_release()); print("release")
