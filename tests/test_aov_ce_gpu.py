"""Analysis of Variance and conditional entropy (the TODO scans of the reference's phase.py:11-15) on
the PDM binning kernel, against oracle/scan_oracle.py's restatement of the published formulas
(parity unpinned by the reference: it has no implementation of either)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import scan_oracle as so
from periodicity_amd import _cabi, phase
from periodicity_amd.core import TSeries

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def curve(n, seed, period=13.7, t_shift=0.0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, float(n), n)) + t_shift
    dy = rng.uniform(0.05, 0.2, n)
    return t, 1.0 + 0.5 * np.sin(2 * np.pi * t / period) + dy * rng.standard_normal(n)


@pytest.mark.parametrize("n,n_periods,n_bins", [(2000, 200, 10), (777, 65, 5), (5000, 3000, 16), (40, 10, 8)])
def test_aov_matches_published_formula(n, n_periods, n_bins):
    t, x = curve(n, 100 + n)
    periods = np.linspace(1.0, 60.0, n_periods)
    got = _cabi.aov_scan(t, x, periods, n_bins)
    want = so.aov_scan(t, x, periods, n_bins)
    np.testing.assert_allclose(got, want, rtol=RTOL)
    assert np.argmax(got) == np.argmax(want)


def test_aov_is_scipy_one_way_anova_over_the_phase_bins():
    """An independent, third-party pin: Theta_AoV IS the one-way ANOVA F statistic of the samples grouped by
    phase bin (Schwarzenberg-Czerny 1989 section 2) - scipy.stats.f_oneway over the groups numpy's own
    ``(t / P) % 1`` and ``floor(phi * r)`` produce, for the kernel AND for the oracle's restatement."""
    from scipy.stats import f_oneway
    t, x = curve(4000, 321, t_shift=-77.25)
    periods = np.concatenate([np.linspace(1.5, 50.0, 40), [13.7, 27.4, 6.85]])
    for r in (4, 10, 16):
        got = _cabi.aov_scan(t, x, periods, r)
        mine = so.aov_scan(t, x, periods, r)
        for i, period in enumerate(periods):
            phi = (t / period) % 1
            # bin membership by the doubles k / r exactly as phase.py:137 compares them
            edges = np.arange(r + 1) / r
            k = np.clip(np.searchsorted(edges, phi, side="right") - 1, 0, r - 1)
            groups = [x[k == j] for j in range(r)]
            assert all(g.size > 0 for g in groups)
            f_stat = f_oneway(*groups).statistic
            assert abs(got[i] / f_stat - 1) < 1e-9 and abs(mine[i] / f_stat - 1) < 1e-9


def test_aov_agrees_with_the_reference_pinned_pdm_kernel():
    """Non-overlapping bins: Theta_AoV = ((N - 1) / theta_PDM - (N - r)) / (r - 1) - the AoV scan against the
    PDM scan whose parity is pinned by the reference's own goldens (G7)."""
    t, x = curve(6000, 77, t_shift=-250.0)
    n = t.size
    periods = np.linspace(2.0, 60.0, 500)
    sigma = float(np.var(x, ddof=1))
    for r in (5, 10, 16):
        theta = _cabi.pdm_scan(t, x, periods, r, 1, sigma)
        np.testing.assert_allclose(_cabi.aov_scan(t, x, periods, r), ((n - 1) / theta - (n - r)) / (r - 1), rtol=1e-9)


def test_aov_split_mode_negative_times_and_edges():
    t, x = curve(40_000, 5, t_shift=-12345.5)                   # few periods x many samples: split mode
    periods = np.linspace(3.0, 40.0, 90)
    np.testing.assert_allclose(_cabi.aov_scan(t, x, periods, 10), so.aov_scan(t, x, periods, 10), rtol=RTOL)
    te = np.arange(300.0)                                        # evenly sampled, commensurate periods: phases
    xe = np.sin(2 * np.pi * te / 12.5) + 0.1 * np.cos(te)        # sit on bin edges and in few bins
    pe = np.array([1.0, 2.0, 2.5, 4.0, 5.0, 12.5, 25.0, 50.0])
    np.testing.assert_allclose(_cabi.aov_scan(te, xe, pe, 5), so.aov_scan(te, xe, pe, 5), rtol=RTOL)
    assert _cabi.aov_scan(t, x, np.empty(0), 10).size == 0
    assert np.all(np.isnan(_cabi.aov_scan(t[:6], x[:6], [3.0, 7.0], 10)))   # n <= r: undefined
    with pytest.raises(ValueError):
        _cabi.aov_scan(t, x[:-1], [1.0], 10)
    with pytest.raises(ValueError):
        _cabi.aov_scan(t, x, [1.0], 500)


@pytest.mark.parametrize("n,n_periods,n_phase,n_mag", [(2000, 200, 10, 5), (513, 70, 4, 3), (5000, 2100, 12, 8)])
def test_conditional_entropy_matches_published_formula(n, n_periods, n_phase, n_mag):
    t, x = curve(n, 300 + n)
    mag = so.magnitude_bins(x, n_mag)
    periods = np.linspace(1.0, 60.0, n_periods)
    got = _cabi.cond_entropy_scan(t, mag, periods, n_phase, n_mag)
    want = so.cond_entropy_scan(t, mag, periods, n_phase, n_mag)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-13)
    assert np.argmin(got) == np.argmin(want)
    with pytest.raises(ValueError):
        _cabi.cond_entropy_scan(t, mag + n_mag, periods, n_phase, n_mag)


def test_classes_find_the_period():
    t, x = curve(3000, 9, period=13.7)
    sig = TSeries(t, x)
    aov = phase.AOV(n_bins=10, p_min=2.0, p_max=40.0, n_periods=4000)(sig)
    ce = phase.ConditionalEntropy(p_min=2.0, p_max=40.0, n_periods=4000)(sig)
    assert abs(aov.period[np.nanargmax(aov.values)] - 13.7) < 0.1
    assert abs(ce.period[np.nanargmin(ce.values)] - 13.7) < 0.1
    assert np.all(np.diff(aov.frequency) > 0) and aov.size == 4000     # FSeries order, as PDM's output


def test_full_size_c5_aov_and_entropy():
    """BASELINE configs[4] shape (N=5e4 x 1e5 trial periods) through the scans the reference only names: EVERY
    period against the C restatement of the published formulas (`oracle/scan_oracle.c`, itself tied to the numpy
    form in tests/test_oracle_golden.py) - round 6, was 25 periods - and the optimum's index identical."""
    co.tune_threads()        # the host is shared: the thread count the C checker runs fastest with, measured once
    n, n_per = 50_000, 100_000
    t, x = curve(n, 20241012)
    periods = np.linspace(1.0, 100.0, n_per)
    pick = np.random.default_rng(0).integers(0, n_per, 25)
    aov = _cabi.aov_scan(t, x, periods, 10)
    np.testing.assert_allclose(co.aov_scan(t, x, periods[pick], 10), so.aov_scan(t, x, periods[pick], 10), rtol=1e-11)
    want = co.aov_scan(t, x, periods, 10)
    np.testing.assert_allclose(aov, want, rtol=RTOL)
    assert int(np.argmax(aov)) == int(np.argmax(want))
    mag = so.magnitude_bins(x, 5)
    ce = _cabi.cond_entropy_scan(t, mag, periods, 10, 5)
    np.testing.assert_allclose(co.cond_entropy_scan(t, mag, periods[pick], 10, 5),
                               so.cond_entropy_scan(t, mag, periods[pick], 10, 5), rtol=1e-12)
    want = co.cond_entropy_scan(t, mag, periods, 10, 5)
    np.testing.assert_allclose(ce, want, rtol=RTOL)
    assert int(np.argmin(ce)) == int(np.argmin(want))
    assert abs(periods[np.argmax(aov)] - 13.7) < 0.05 and abs(periods[np.argmin(ce)] - 13.7) < 0.05
    # Gregory-Loredo on the same stamps (the statistic of arrival times: only `t` enters), m = 6 bins x 4 offsets
    gl = _cabi.gl_scan(t, periods, 6, 4)
    np.testing.assert_allclose(co.gl_scan(t, periods[pick], 6, 4), so.gl_scan(t, periods[pick], 6, 4), rtol=1e-11, atol=1e-9)
    want = co.gl_scan(t, periods, 6, 4)
    np.testing.assert_allclose(gl, want, rtol=RTOL, atol=1e-9)
    assert int(np.argmax(gl)) == int(np.argmax(want))


# ---- Gregory-Loredo (phase.py:13, TODO upstream): counts-only histogram path -----------------------------------
def arrival_times(n, period, depth, seed, t_shift=0.0):
    """Events of a periodic Poisson process: uniform times thinned by 1 + depth * cos(2 pi t / P)."""
    rng = np.random.default_rng(seed)
    t = rng.uniform(0, 50.0 * period, 3 * n)
    keep = rng.uniform(0, 1 + depth, t.size) < 1 + depth * np.cos(2 * np.pi * t / period)
    return np.sort(t[keep][:n]) + t_shift


@pytest.mark.parametrize("n,n_periods,m,n_off", [(800, 150, 4, 8), (3000, 70, 12, 8), (257, 300, 2, 1), (5000, 4200, 7, 5),
                                                 (40, 9, 19, 10)])
def test_gregory_loredo_matches_published_formula(n, n_periods, m, n_off):
    t = arrival_times(n, 3.7, 0.8, n + m)
    periods = np.linspace(0.5, 12.0, n_periods)
    got = _cabi.gl_scan(t, periods, m, n_off)
    want = so.gl_scan(t, periods, m, n_off)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-9)
    assert np.argmax(got) == np.argmax(want)


def test_gregory_loredo_edges_devices_and_class():
    t = arrival_times(2500, 5.3, 0.9, 7, t_shift=-400.0)             # negative times
    periods = np.linspace(1.0, 20.0, 400)
    want = so.gl_scan(t, periods, 6, 4)
    np.testing.assert_allclose(_cabi.gl_scan(t, periods, 6, 4), want, rtol=RTOL, atol=1e-9)
    assert np.array_equal(_cabi.gl_scan(t, periods, 6, 4, devices=(0, 0, 0)), _cabi.gl_scan(t, periods, 6, 4))
    tn = t.copy()
    tn[[3, 77]] = np.nan                                              # NaN stamps are events of no bin: N drops
    np.testing.assert_allclose(_cabi.gl_scan(tn, periods, 6, 4), so.gl_scan(tn, periods, 6, 4), rtol=RTOL, atol=1e-9)
    te = np.arange(600.0)                                             # phases on the bin edges, phi == 1.0 cases
    pe = np.array([1.0, 2.0, 2.5, 4.0, 5.0, 12.5, 25.0, 50.0, 3.0000000000000004])
    np.testing.assert_allclose(_cabi.gl_scan(te, pe, 5, 2), so.gl_scan(te, pe, 5, 2), rtol=RTOL, atol=1e-9)
    tiny = np.concatenate([[-1e-300, -1e-18], t[:500] + 401.0])       # (t / P) % 1 == 1.0 exactly
    np.testing.assert_allclose(_cabi.gl_scan(tiny, periods[:50], 4, 4), so.gl_scan(tiny, periods[:50], 4, 4),
                               rtol=RTOL, atol=1e-9)
    assert _cabi.gl_scan(t, np.empty(0), 4, 4).size == 0
    with pytest.raises(ValueError):
        _cabi.gl_scan(t, periods, 20, 10)                             # 200 fine bins do not fit
    # the class: log odds summed over m = 2 .. m_max peak at the injected period (or its harmonics' base)
    gl = phase.GregoryLoredo(m_max=8, n_offsets=6, p_min=2.0, p_max=12.0, n_periods=600)
    res = gl(TSeries(arrival_times(4000, 5.3, 0.9, 11)))
    assert abs(res.period[np.argmax(res.values)] - 5.3) < 0.05 and res.values.max() > 10.0
    assert sorted(gl.log_s) == list(range(2, 9))
    plan = _cabi.PhasePlan((0, 0))
    plan.upload(t, t)
    plan.scan("gregory_loredo", periods, 6 * 4, 6)
    assert np.array_equal(plan.download(), _cabi.gl_scan(t, periods, 6, 4))
    plan.close()


def test_counts_only_kinds_cut_long_curves_into_slices_of_16_bit_cells():
    """The conditional-entropy / Gregory-Loredo histograms keep two 16-bit cells per LDS word, so a workgroup
    bins at most 65 280 samples: longer curves are cut into sample slices whose partial histograms are added
    (unpacked) by the finishing launch - whatever the period count, and also with PDC_PDM_SPLIT=0 (checked in
    a child process)."""
    import os
    import subprocess
    import sys
    t, x = curve(140_000, 17, period=9.1)
    mag = so.magnitude_bins(x, 4)
    for periods in (np.linspace(2.0, 30.0, 33), np.linspace(2.0, 30.0, 5000)):
        pick = np.unique(np.linspace(0, periods.size - 1, 25).astype(int))
        got = _cabi.cond_entropy_scan(t, mag, periods, 8, 4)
        np.testing.assert_allclose(got[pick], so.cond_entropy_scan(t, mag, periods[pick], 8, 4), rtol=RTOL)
        got = _cabi.gl_scan(t, periods, 5, 6)
        np.testing.assert_allclose(got[pick], so.gl_scan(t, periods[pick], 5, 6), rtol=RTOL, atol=1e-9)
    same = np.full(70_000, 3.25)                      # every sample in ONE cell: a count of 70 000 > 65 535
    np.testing.assert_allclose(_cabi.gl_scan(same, [2.0, 7.0], 4, 2), so.gl_scan(same, np.array([2.0, 7.0]), 4, 2),
                               rtol=RTOL, atol=1e-9)
    code = ("import numpy as np; from periodicity_amd import _cabi; from oracle import scan_oracle as so;"
            "same = np.full(70_000, 3.25);"
            "a = _cabi.gl_scan(same, [2.0, 7.0], 4, 2); b = so.gl_scan(same, np.array([2.0, 7.0]), 4, 2);"
            "assert np.allclose(a, b, rtol=1e-9, atol=1e-9), (a, b); print('ok')")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PDC_PDM_SPLIT="0"), cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]
