"""Device-side peak picking against scipy.signal.find_peaks — the routine
FSeries.period_at_highest_peak is built on (core.py:283-317, 952-955)."""
import numpy as np
import pytest
from scipy.signal import find_peaks

from periodicity_amd import _cabi
from periodicity_amd.core import FSeries

pytestmark = pytest.mark.gpu


def scipy_highest(x):
    idx, _ = find_peaks(x, prominence=0.0)
    if idx.size == 0:
        return -1, np.nan
    j = idx[np.nanargmax(x[idx])]
    return int(j), float(x[j])


def test_matches_scipy_on_random_plateau_nan_and_edge_cases():
    rng = np.random.default_rng(0)
    rows = [rng.standard_normal(5000), rng.standard_normal(5000).round(1),      # many flat tops
            np.arange(50.0), -np.arange(50.0), np.zeros(40), np.array([1.0]), np.array([1.0, 2.0]),
            np.array([0.0, 1.0, 0.0]), np.array([0.0, 1.0, 1.0, 0.0]), np.array([0.0, 1.0, 1.0, 1.0, 0.0]),
            np.array([0.0, 2.0, 2.0, 3.0, 3.0, 1.0, 3.0, 3.0, 0.0]),              # equal maxima: first wins
            np.array([0.0, 1.0, 1.0]), np.array([1.0, 1.0, 0.0]),                  # flat top touching an edge
            np.array([5.0, 1.0, 2.0, 1.0, 9.0])]                                  # edges are never peaks
    nanrow = rng.standard_normal(300)
    nanrow[[10, 11, 150]] = np.nan
    rows.append(nanrow)
    for x in rows:
        got = _cabi.highest_peak(x)
        want = scipy_highest(x)
        assert got[0] == want[0], (x[:12], got, want)
        assert got[1] == want[1] or (np.isnan(got[1]) and np.isnan(want[1]))
    batch = rng.standard_normal((37, 1234)).round(2)
    idx, val = _cabi.highest_peak(batch)
    for b in range(37):
        j, v = scipy_highest(batch[b])
        assert idx[b] == j and val[b] == v


def test_batched_periodograms_reduced_on_device():
    rng = np.random.default_rng(5)
    lens = [300, 511, 1000, 64]
    ts, ys, dys = [], [], []
    for i, n in enumerate(lens):
        t = np.sort(rng.uniform(0, n, n))
        dy = rng.uniform(0.05, 0.2, n)
        ts.append(t)
        dys.append(dy)
        ys.append(np.sin(2 * np.pi * t / (7.0 + 3 * i)) + dy * rng.standard_normal(n))
    offsets = np.concatenate([[0], np.cumsum(lens)])
    freq = np.arange(0.003, 0.45, 0.0004)
    f0, delta, nf = _cabi.grid_params(freq)
    t, y, dy = map(np.concatenate, (ts, ys, dys))
    idx, val = _cabi.gls_batch_highest_peak(t, y, dy, offsets, f0, delta, nf)
    power, _, _ = _cabi.gls_scan_batch(t, y, dy, offsets, f0, delta, nf)
    for b in range(len(lens)):
        fs = FSeries(freq, power[b])
        assert fs.period_at_highest_peak == 1 / freq[idx[b]]        # the reference's consumer
        assert val[b] == power[b, idx[b]]
        assert abs(1 / freq[idx[b]] - (7.0 + 3 * b)) < 0.5
