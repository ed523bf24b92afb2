import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from periodicity_amd import _cabi
from scipy.signal import find_peaks
rng = np.random.default_rng(3)
x = rng.standard_normal(4000)
idx, res = find_peaks(x, prominence=0.0)
prom = dict(zip(idx, res["prominences"]))
for k in (1, 3, 8):
    for bp in (False, True):
        g = _cabi.peaks_topk(x, k=k, by_prominence=bp)
        print(k, bp, "idx", g["indices"][0], "\n   prom", g["prominences"][0], "\n   want", [prom.get(i) for i in g["indices"][0]], "\n   h", g["heights"][0], [x[i] for i in g["indices"][0]])
