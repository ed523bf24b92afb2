"""StringLength of bench.synth_curve(N) over the reference's period grid, saved to a file: the streamed kernels' two
ways of fetching a bin's records (slices of t[] / m[] against the partition kernel's lists, PDC_SL_SLICES=0) must agree
bit for bit at the sizes the streamed path serves (developer tool; tests/test_phase_gpu.py runs it in child processes).

    python tools/sl_slices_ab.py out.npy 250000x4096 [1000000x512 ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from periodicity_amd import _cabi  # noqa: E402

out = {}
for spec in sys.argv[2:]:
    n, n_per = (int(v) for v in spec.split("x"))
    t, y, _ = bench.synth_curve(n, 5, period=13.7)
    m = (y - y.max()) / (2 * (y.max() - y.min())) + 0.25
    df = 0.1 / (t[-1] - t[0])
    periods = 1 / np.linspace(n_per * df, df, n_per)
    out[spec] = _cabi.stringlength_scan(t, m, periods)
    q0, q1 = t[0] / periods, t[-1] / periods                   # (summed as the samples stand: less than one cycle)

    out[spec + ":one_cycle"] = (periods > 0) & ((np.floor(q1) == np.floor(q0)) | ((np.floor(q1) - np.floor(q0) == 1) & (q1 % 1 < q0 % 1)))
np.savez(sys.argv[1], **out)
print("ok")
