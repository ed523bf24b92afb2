"""The Supersmoother period search (a one-line TODO upstream, spectral.py:8) through the C ABI against the oracle's
restatement of the published algorithm (Friedman 1984 `supsmu`, periodic; Reimann 1994).  PARITY UNPINNED BY THE
REFERENCE: the oracle is pinned to a literal restatement of the Fortran's updating formulas (CPU test below)."""
import numpy as np
import pytest

from oracle import scan_oracle as so
from periodicity_amd import _cabi
from periodicity_amd.core import TSeries
from periodicity_amd.phase import SuperSmoother

RTOL = 1e-9


def curve(n, seed, even=False, period=7.3):
    rng = np.random.default_rng(seed)
    t = np.arange(float(n)) * 0.1 if even else np.sort(rng.uniform(0, 0.1 * n, n))
    y = np.sin(2 * np.pi * t / period) + 0.3 * np.cos(4 * np.pi * t / period) + 0.2 * rng.standard_normal(n)
    return t, y


def test_oracle_window_sums_equal_the_literal_updating_formulas():
    """CPU: the vectorised smoother (window sums) against the literal restatement of Friedman's `smooth` (one point
    out, one point in), ties included, and the whole of `supsmu` built on either."""
    rng = np.random.default_rng(1)
    for n in (40, 101, 400):
        x = np.sort(rng.uniform(0, 1, n))
        y = np.sin(2 * np.pi * x) + 0.3 * rng.standard_normal(n)
        if n == 101:
            x[10] = x[11]
            x[50:53] = x[50]
        v = (1e-3 * (x[3 * (n // 4) - 1] - x[n // 4 - 1])) ** 2
        for span in so.SS_SPANS:
            a, ra = so.ss_smooth_incremental(x, y, span, v, True)
            b, rb = so.ss_smooth(x, y, span, v, True)
            np.testing.assert_allclose(b, a, rtol=0, atol=1e-12)
            np.testing.assert_allclose(rb, ra, rtol=0, atol=1e-12)
        for alpha in (0.0, 4.0):
            np.testing.assert_allclose(so.supersmoother(x, y, alpha), so.supersmoother(x, y, alpha, so.ss_smooth_incremental),
                                       rtol=0, atol=1e-12)
    t, y = curve(2000, 3)
    per = np.linspace(5.0, 10.0, 21)
    assert abs(per[np.argmin(so.supersmoother_scan(t, y, per))] - 7.3) < 0.26


def test_oracle_window_sums_stay_exact_for_phases_crowded_into_a_sliver_of_the_cycle():
    """CPU: a period far beyond the baseline folds the samples into a sliver of the cycle (spread 1e-5 ... 1e-7).  The
    oracle's window sums (long double, abscissae relative to the median, wrapped parts added through their own sums)
    equal a brute-force fit of every window in LOCALLY centred long-double arithmetic to rounding - where the literal
    double-precision updating formulas of the Fortran, and the round-4 oracle that prefixed over shifted abscissae, lose
    var = Sxx - fbw xm^2 altogether (errors of 1e-4 ... 0.2).  The device kernels are held to THIS oracle."""
    L = np.longdouble
    rng = np.random.default_rng(2)
    n = 300
    for spread in (1e-5, 1e-7):
        x = np.sort(2.45e-3 + spread * rng.uniform(0, 1, n))
        y = np.sin(2 * np.pi * (x - x[0]) / spread * 3) + 0.3 * rng.standard_normal(n)
        v = (1e-3 * (x[3 * (n // 4) - 1] - x[n // 4 - 1])) ** 2
        for span in so.SS_SPANS:
            ibw = so.ss_half_width(n, span)
            want = np.empty(n)
            for j in range(n):
                idx = np.arange(j - ibw, j + ibw + 1)
                xs = np.where(idx < 0, x[idx % n] - 1, np.where(idx >= n, x[idx % n] + 1, x[idx % n])).astype(L) - L(x[j])
                ys = y[idx % n].astype(L)
                xm, ym = xs.mean(), ys.mean()
                var, cvar = ((xs - xm) ** 2).sum(), ((xs - xm) * (ys - ym)).sum()
                want[j] = (cvar / var if var > v else 0) * (0 - xm) + ym
            got, _ = so.ss_smooth(x, y, span, v, False)
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("n,even,alpha", [(300, False, 0.0), (5000, False, 0.0), (5000, False, 5.0), (9000, False, 0.0),
                                          (4000, True, 0.0), (6000, True, 2.0)])
def test_supersmoother_scan_matches_the_oracle(n, even, alpha):
    """Below 4096 samples every period is sorted by the fallback kernel, above by the streamed kernels; even
    sampling at commensurate periods gives tied phases (averaged fits) and clustered bins (fallback sort)."""
    t, y = curve(n, n + 1, even)
    periods = np.concatenate([np.linspace(2.1, 40.0, 24), [7.3, 10.0, 0.5, 25.0, 3.7 * t[-1]]])
    got = _cabi.supersmoother_scan(t, y, periods, alpha)
    want = so.supersmoother_scan(t, y, periods, alpha)
    np.testing.assert_allclose(got, want, rtol=RTOL)
    assert np.array_equal(got, _cabi.supersmoother_scan(t, y, periods, alpha))      # bitwise repeatable


@pytest.mark.gpu
@pytest.mark.parametrize("n", [50_000, 74_326, 200_000])
def test_supersmoother_scan_matches_the_oracle_at_the_sizes_it_runs(n):
    """N = 5e4 is bench.py's shape, 74 326 the reference's SunSpots curve, 2e5 the class the workspace is sized for:
    ten periods each - a commensurate (tied phases) one on an evenly sampled copy, periods beyond the baseline, the
    bass control on - against the oracle's long-double window sums.  The observed error is printed per N so that the
    trend with N (the device forms windows from fp64 running sums) is on record."""
    worst = {}
    for even, alpha in ((False, 0.0), (False, 6.0), (True, 0.0)):
        t, y = curve(n, n + 7, even)
        periods = np.array([0.61, 2.1, 7.3, 10.0, 25.0, 33.3, 0.013 * t[-1], 0.31 * t[-1], 1.7 * t[-1], 12.0 * t[-1]])
        if even and n > 60_000:
            periods = periods[[1, 3, 6, 8]]        # (long tied runs are one oracle loop each: keep the CPU leg short)
        got = _cabi.supersmoother_scan(t, y, periods, alpha)
        want = so.supersmoother_scan(t, y, periods, alpha)
        rel = np.abs(got - want) / np.abs(want)
        worst[(even, alpha)] = float(rel.max())
        np.testing.assert_allclose(got, want, rtol=RTOL)
        assert np.array_equal(got, _cabi.supersmoother_scan(t, y, periods, alpha))
    print(f"supersmoother N={n}: max rel err vs oracle " + ", ".join(f"even={e} alpha={a}: {v:.2e}" for (e, a), v in worst.items()))


@pytest.mark.gpu
def test_supersmoother_samples_in_any_order_at_the_streamed_sizes():
    """N >= 262 144 handed over in a random order (the C ABI allows it; a TSeries never is): the device orders the
    samples by time first (csrc/timesort.inc, stable - duplicates of a time stamp keep the caller's order) and the
    result is the one for TSeries(t, y); periods beyond the baseline (written as the samples stand) included."""
    n = 300_000
    t, y = curve(n, 99)
    t[5:n:9] = t[4:n - 1:9]
    rng = np.random.default_rng(5)
    order = rng.permutation(n)
    ts, ys = t[order], y[order]
    back = np.argsort(ts, kind="stable")
    periods = np.array([2.1, 7.3, 33.3, 0.31 * t[-1], 1.7 * t[-1], 12.0 * t[-1]])
    got = _cabi.supersmoother_scan(ts, ys, periods, 0.0)
    np.testing.assert_allclose(got, so.supersmoother_scan(ts[back], ys[back], periods, 0.0), rtol=RTOL)
    assert np.array_equal(got, _cabi.supersmoother_scan(ts, ys, periods, 0.0))
    # in order, the same samples take the same kernels: the same bits
    assert np.array_equal(got, _cabi.supersmoother_scan(ts[back], ys[back], periods, 0.0))


@pytest.mark.gpu
@pytest.mark.parametrize("n,alpha", [(30_000, 0.0), (30_000, 7.0), (8_192, 0.0), (4_096, 3.0)])
def test_tiled_smoother_short_runs_of_equal_phases(n, alpha):
    """Duplicate time stamps give runs of two and three equal phases at every period: the tiled kernels (n >= 4096)
    average the fitted values over such runs from their halos - wherever the runs fall with respect to the tiles and
    the segments - and hand nothing back to the generic kernel; a tail of 300 equal stamps (a run longer than a halo)
    is the generic kernel's."""
    t, y = curve(n, 3 * n + 1)
    t[5:n:7] = t[4:n - 1:7]
    t[100:n:1001] = t[99:n - 1:1001]
    t[101:n:1001] = t[99:n - 2:1001]
    periods = np.array([0.37, 2.1, 7.3, 9.99, 31.0, 0.2 * t[-1], 2.5 * t[-1]])
    got = _cabi.supersmoother_scan(t, y, periods, alpha)
    np.testing.assert_allclose(got, so.supersmoother_scan(t, y, periods, alpha), rtol=RTOL)
    assert np.array_equal(got, _cabi.supersmoother_scan(t, y, periods, alpha))
    t[-300:] = t[-300]
    got = _cabi.supersmoother_scan(t, y, periods[:4], alpha)
    np.testing.assert_allclose(got, so.supersmoother_scan(t, y, periods[:4], alpha), rtol=RTOL)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5_000, 40_000])
def test_supersmoother_periods_that_outlast_the_samples(n):
    """p > baseline: less than one cycle, the phase order is the time order - or, with Julian-date stamps, the later
    samples first (a cycle boundary inside the samples): written as they stand by ss_direct_kernel, no sort (the
    bitonic fallback took ~200 passes over the padded curve for each such period).  Duplicated stamps keep their order."""
    t, y = curve(n, n + 3)
    t = t + 2454953.5
    t[7:n:11] = t[6:n - 1:11]
    base = t[-1] - t[0]
    periods = np.concatenate([base * np.array([0.999, 1.0, 1.001, 1.5, 2.0, 3.3, 9.99, 57.0]), [2454953.5, 2454953.5 / 2, 1e7, 1e9]])
    for alpha in (0.0, 4.0):
        got = _cabi.supersmoother_scan(t, y, periods, alpha)
        np.testing.assert_allclose(got, so.supersmoother_scan(t, y, periods, alpha), rtol=RTOL)


@pytest.mark.gpu
def test_supersmoother_class_finds_the_period_and_edges():
    t, y = curve(6000, 11)
    res = SuperSmoother(p_min=5.0, p_max=10.0, n_periods=201)(TSeries(t, y))
    assert res.size == 201 and abs(res.period[np.argmin(res.values)] - 7.3) < 0.03
    assert np.all(np.diff(res.frequency) > 0)
    with pytest.raises(ValueError):
        _cabi.supersmoother_scan(t[:4], y[:4], [1.0])                  # the woofer window needs five points
    with pytest.raises(ValueError):
        _cabi.supersmoother_scan(t, y, [1.0], alpha=11.0)
    assert _cabi.supersmoother_scan(t, y, []).size == 0
    five = _cabi.supersmoother_scan(t[:5], y[:5], [0.3, 1.7])
    np.testing.assert_allclose(five, so.supersmoother_scan(t[:5], y[:5], [0.3, 1.7]), rtol=RTOL)


@pytest.mark.gpu
def test_supersmoother_over_device_slots_equals_one_launch():
    """`devices=(...)` cuts the period grid into one slab per listed slot (the phase plan, kind 5): same values."""
    t, y = curve(5000, 21)
    periods = np.linspace(3.0, 30.0, 50)
    one = _cabi.supersmoother_scan(t, y, periods, 3.0)
    many = _cabi.supersmoother_scan(t, y, periods, 3.0, devices=(0, 0, 0))
    assert np.array_equal(one, many)
    res = SuperSmoother(alpha=3.0, p_min=3.0, p_max=30.0, n_periods=50, devices=(0, 0))(TSeries(t, y))
    assert np.array_equal(res.values[::-1], one)
