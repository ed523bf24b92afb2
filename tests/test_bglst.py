"""BGLST - the Bayesian generalised Lomb-Scargle periodogram with linear trend the reference exports as an empty
class (/root/reference/src/periodicity/spectral.py:7,207-208): PARITY UNPINNED BY THE REFERENCE.  The oracle
(oracle/scan_oracle.py:bglst_loglik, 80-bit, 4 x 4 marginalisation) is pinned to a third party - scipy's multivariate
normal on the dense n x n covariance the model implies - in the CPU tests; the GPU tests compare the HIP kernel
(gls_scan_kernel<K, MODE_TREND, 1> + bglst_loglik epilogue, through the C ABI) with the oracle."""
import inspect

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import scan_oracle as so
from periodicity_amd import _cabi
from periodicity_amd.core import FSeries, TSeries
from periodicity_amd.spectral import BGLST, GLS

RTOL = 1e-9   # of max(|log-likelihood|, W yy): the likelihood is a difference of two sums of that size


def curve(n, seed, period=6.3, slope=0.05, t_offset=0.0, baseline=None):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, float(baseline or n), n)) + t_offset
    err = rng.uniform(0.1, 0.3, n)
    y = 2.0 + slope * (t - t[0]) + 0.7 * np.sin(2 * np.pi * t / period + 0.4) + err * rng.standard_normal(n)
    return t, y, err


# ---- CPU: the oracle against scipy, properties of the statistic, the host class -----------------------------------
@pytest.mark.parametrize("priors,t_ref,with_err", [((1.0, 2.0, 3.0), 120.0, True), ((0.3, 0.1, 10.0), 100.0, True),
                                                   ((5.0, 5.0, 5.0), 140.0, False)])
def test_oracle_equals_the_dense_gaussian_marginal(priors, t_ref, with_err):
    t, y, err = curve(70, 2, t_offset=100.0, baseline=40.0)
    err = err if with_err else np.ones_like(y)
    f = np.array([0.01, 0.1, 1 / 6.3, 0.3, 1.7])
    got = so.bglst_loglik(t, y, err if with_err else None, f, *priors, t_ref)
    want = np.array([so.bglst_loglik_dense(t, y, err, x, *priors, t_ref) for x in f])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-8 * np.abs(want).max())


def test_double_precision_checker_is_pinned_to_the_80_bit_oracle():
    """`co.bglst_loglik_f64` (the C direct-sum checker's sums, float64 marginalisation) is what the GPU suite checks
    all 1e6 bins of the C2-shaped run with: here against the 80-bit oracle, Julian-date stamps, with and without
    uncertainties, and - one frequency - against scipy's dense Gaussian."""
    t, y, err = curve(3000, 5, t_offset=2454900.5, baseline=300.0)
    f = 0.003 + 0.0007 * np.arange(400)
    for e in (err, None):
        fast = co.bglst_loglik_f64(t, y, e, f, 0.8, 1.5, 4.0, t[0] + 100.0)
        want = so.bglst_loglik(t, y, e, f[::20], 0.8, 1.5, 4.0, t[0] + 100.0)
        scale = max(float(np.abs(want).max()), float(np.sum((y / (err if e is not None else 1.0)) ** 2)))
        assert np.max(np.abs(fast[::20] - want)) <= 1e-12 * scale
    small = slice(0, 80)
    dense = so.bglst_loglik_dense(t[small], y[small], err[small], f[37], 0.8, 1.5, 4.0, t[0] + 10.0)
    assert abs(co.bglst_loglik_f64(t[small], y[small], err[small], f[37:38], 0.8, 1.5, 4.0, t[0] + 10.0)[0] - dense) <= 1e-8 * abs(dense)


def test_oracle_properties_of_the_statistic():
    t, y, err = curve(400, 5, baseline=120.0)
    f = np.linspace(0.005, 0.5, 600)
    ll = so.bglst_loglik(t, y, err, f, 1.0, 1.0, 3.0, t_ref=60.0)
    assert abs(1 / f[np.argmax(ll)] - 6.3) < 0.1                       # the injected period, trend and all
    # the trigonometric basis may be rotated (a shift of the time origin of cos / sin only): same likelihood - the
    # invariance the device kernel (sums over t - t[0]) and the oracle's own default origin rely on, shown with the
    # DENSE form, which knows nothing of either
    for x in (f[10], f[300], f[599]):
        a = so.bglst_loglik_dense(t[:90], y[:90], err[:90], x, 1.0, 1.0, 3.0, 60.0)
        phi0 = so.bglst_design(t[:90], x, 60.0, trig_origin=0.0).astype(float)
        phi1 = so.bglst_design(t[:90], x, 60.0, trig_origin=17.25).astype(float)
        cov = [p @ np.diag([1.0, 1.0, 1.0, 9.0]) @ p.T + np.diag(err[:90] ** 2) for p in (phi0, phi1)]
        np.testing.assert_allclose(cov[0], cov[1], rtol=0, atol=1e-12)      # the covariance itself is the same matrix
        assert abs(a - so.bglst_loglik(t[:90], y[:90], err[:90], [x], 1.0, 1.0, 3.0, 60.0)[0]) < 1e-8 * abs(a)
    shifted = so.bglst_loglik(t + 1000.0, y, err, f, 1.0, 1.0, 3.0, t_ref=1060.0)
    np.testing.assert_allclose(shifted, ll, rtol=0, atol=1e-7)
    # a tighter prior on the trend than the data's slope costs likelihood everywhere
    assert np.all(so.bglst_loglik(t, y, err, f[:50], 1.0, 1e-3, 3.0, 60.0) < ll[:50])


def test_host_class_mirrors_gls_and_validates():
    sig = inspect.signature(BGLST.__init__)
    assert list(sig.parameters)[:4] == ["self", "fmin", "fmax", "n"]             # GLS's grid arguments, same order
    assert all(sig.parameters[k].kind is inspect.Parameter.KEYWORD_ONLY for k in ("sigma_A", "sigma_alpha", "sigma_beta", "t_ref"))
    assert issubclass(BGLST, GLS)
    t, y, err = curve(50, 1, baseline=30.0)
    b = BGLST()
    sA, sa, sb, t_ref = b.priors(TSeries(t, y))
    assert sA == pytest.approx(np.std(y)) and sa == sA and t_ref == pytest.approx(0.5 * (t[0] + t[-1]))
    assert sb == pytest.approx(np.sqrt(np.var(y) + np.mean(y) ** 2))
    sc = BGLST._scalars(t, y, err, 1.0, 2.0, 3.0, t_ref)
    assert sc.shape == (12,) and sc[0] == pytest.approx(np.sum(err ** -2.0)) and sc[7] == pytest.approx(1 / (t[-1] - t[0]))
    with pytest.raises(NotImplementedError):
        b.bootstrap(10)
    if _cabi.device_count() == 0:
        with pytest.raises((RuntimeError, ValueError)):      # no GPU: the class fails loudly, never a CPU answer
            b(TSeries(t, y), err)


# ---- GPU ------------------------------------------------------------------------------------------------------------
def assert_close(got, want, scale):
    tol = RTOL * max(float(np.abs(want).max()), scale)
    assert np.max(np.abs(got - want)) <= tol, (np.max(np.abs(got - want)), tol)


@pytest.mark.gpu
@pytest.mark.parametrize("n,nf,with_err,t_offset", [(200, 700, True, 0.0), (3000, 5000, True, 2454900.5), (3000, 2049, False, -50.0),
                                                    (20000, 300, True, 0.0)])
def test_device_matches_the_oracle(n, nf, with_err, t_offset):
    t, y, err = curve(n, n + nf, t_offset=t_offset, baseline=n / 10.0)
    t_ref = 0.5 * (t[0] + t[-1])
    priors = (0.8, 1.5, 4.0)
    f0, delta = 0.002, 0.9 / nf
    freq = f0 + delta * np.arange(nf)
    e = err if with_err else np.ones_like(y)
    sc = BGLST._scalars(t, y, e, *priors, t_ref)
    got = _cabi.bglst_scan(t, y, err if with_err else None, f0, delta, nf, sc)
    pick = np.unique(np.concatenate([np.arange(0, nf, max(1, nf // 150)), [0, nf - 1, int(np.argmax(got))]]))
    want = so.bglst_loglik(t, y, err if with_err else None, freq[pick], *priors, t_ref)
    assert_close(got[pick], want, sc[0] * sc[1])
    assert int(np.argmax(got)) == pick[int(np.argmax(want))]
    # a slab of the grid (j_begin) reproduces the same bins (another tile phase: to rounding)
    part = _cabi.bglst_scan(t, y, err if with_err else None, f0, delta, min(nf, 257), sc, j_begin=nf // 3)
    assert_close(part[: nf - nf // 3], got[nf // 3:nf // 3 + 257][: part.size], sc[0] * sc[1])


@pytest.mark.gpu
def test_class_call_finds_the_period_under_a_trend_and_posterior_mean():
    t, y, err = curve(1500, 11, period=9.1, slope=0.02, baseline=300.0)
    b = BGLST(fmax=0.5)
    ll = b(TSeries(t, y), err)
    assert isinstance(ll, FSeries) and ll.size == b.frequency.size
    assert abs(ll.period_at_highest_peak - 9.1) < 0.05
    pick = np.linspace(0, ll.size - 1, 80).astype(int)
    want = so.bglst_loglik(t, y, err, ll.frequency[pick], *b.priors(TSeries(t, y)))
    assert_close(ll.values[pick], want, float(np.sum(y * y / err ** 2)))
    a, bb, alpha, beta = b.posterior_mean(1 / 9.1)
    assert abs(np.hypot(a, bb) - 0.7) < 0.05 and abs(alpha - 0.02) < 0.002
    # raw arrays are wrapped like GLS does (spectral.py:86-87), unit uncertainties by default
    raw = BGLST(fmax=0.4)(y)
    assert raw.size == BGLST(fmax=0.4)._grid(TSeries(values=y)).size
    with pytest.raises(ValueError):
        BGLST()(TSeries(t, y), err[:-1])
    with pytest.raises(ValueError):
        BGLST()(TSeries(t[:3], y[:3]))


@pytest.mark.gpu
def test_full_size_c2_shape_every_bin():
    """BASELINE configs[1]'s shape (N = 1e5 x nf = 1e6) through the trend kernel: a sample of bins against the 80-bit
    oracle, EVERY bin against the double-precision one, the peak at the injected period."""
    n, nf = 100_000, 1_000_000
    rng = np.random.default_rng(20241010)
    t = np.sort(rng.uniform(0, float(n), n))
    err = rng.uniform(0.05, 0.2, n)
    y = 1.0 + 1e-5 * t + 0.5 * np.sin(2 * np.pi * t / 37.3) + err * rng.standard_normal(n)
    df = 1.0 / (t[-1] - t[0]) / 5
    f0 = 0.5 * df
    t_ref = 0.5 * (t[0] + t[-1])
    sc = BGLST._scalars(t, y, err, 1.0, 1.0, 2.0, t_ref)
    got = _cabi.bglst_scan(t, y, err, f0, df, nf, sc)
    assert np.all(np.isfinite(got))
    peak = int(np.argmax(got))
    assert abs(1 / (f0 + peak * df) - 37.3) < 0.01
    pick = np.unique(np.concatenate([rng.integers(0, nf, 40), [0, nf - 1, peak, 2047, 2048]]))
    want = so.bglst_loglik(t, y, err, f0 + df * pick, 1.0, 1.0, 2.0, t_ref)
    assert_close(got[pick], want, sc[0] * sc[1])
    # EVERY bin: the double-precision checker (the direct-sum C oracle's sums + the same marginalisation in numpy),
    # first re-proved on the sampled bins against the 80-bit oracle, then trusted on all 1e6
    co.tune_threads()
    fast = co.bglst_loglik_f64(t, y, err, f0 + df * pick, 1.0, 1.0, 2.0, t_ref)
    assert np.max(np.abs(fast - want)) <= 1e-11 * max(float(np.abs(want).max()), sc[0] * sc[1])
    full = co.bglst_loglik_f64(t, y, err, f0 + df * np.arange(nf), 1.0, 1.0, 2.0, t_ref)
    assert_close(got, full, sc[0] * sc[1])
    assert int(np.argmax(full)) == peak
