"""Seeded randomised sweeps of all scans against the oracles: odd sizes, ragged batches, time
offsets, tied phases, extreme periods — the corners the hand-written cases might miss."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import scan_oracle as so
from periodicity_amd import _cabi

pytestmark = pytest.mark.gpu


def exact_power_and_sensitivity(t, y, err, freq, fit_mean, psd, eps=1e-12, trials=4):
    """Long-double-sum power through the reference epilogue, and per bin the largest change of that
    power when each of the six sums moves by ``eps`` (sums are O(1): weights add up to one)."""
    w, yc, e = so.gls_weights(y, err, fit_mean)
    f = np.ascontiguousarray(freq, dtype=float)
    Sh, Ch = co.trig_sums_exact(t, w * yc, f)
    S2, C2 = co.trig_sums_exact(t, w, 2 * f)
    S, C = co.trig_sums_exact(t, w, f) if fit_mean else (None, None)
    YY = np.dot(w, yc ** 2)
    want = so.gls_epilogue(Sh, Ch, S2, C2, S, C, YY, fit_mean, psd, e)
    rng = np.random.default_rng(0)
    wobble = np.zeros_like(want)
    # rank-deficient bins (e.g. two distinct time stamps under a three-parameter fit): the variance of
    # one rotated basis function, CC or SS of spectral.py:124-127, vanishes together with its
    # numerator and the term is 0/0 - whatever an implementation returns there is rounding noise
    with np.errstate(divide="ignore", invalid="ignore"):
        tan2 = (S2 - 2 * S * C) / (C2 - (C * C - S * S)) if fit_mean else S2 / C2
        c2w = 1 / np.sqrt(1 + tan2 * tan2)
        s2w = tan2 * c2w
        CC = 0.5 * (1 + C2 * c2w + S2 * s2w)
        SS = 0.5 * (1 - C2 * c2w - S2 * s2w)
        if fit_mean:
            cw = np.sqrt(0.5) * np.sqrt(1 + c2w)
            sw = np.sqrt(0.5) * np.sign(s2w) * np.sqrt(1 - c2w)
            CC = CC - (C * cw + S * sw) ** 2
            SS = SS - (S * cw - C * sw) ** 2
        singular = ~(np.minimum(np.abs(CC), np.abs(SS)) > 1e-10)
    for _ in range(trials):
        moved = [None if a is None else a + eps * rng.choice([-1.0, 1.0], a.shape)
                 for a in (Sh, Ch, S2, C2, S, C)]
        other = so.gls_epilogue(*moved, YY, fit_mean, psd, e)
        with np.errstate(invalid="ignore"):
            wobble = np.fmax(wobble, np.abs(other - want))
    wobble[singular] = np.inf
    # the slack is for ill-conditioned bins only: where both rotated variances are comfortably away
    # from zero the plain 1e-6 gate applies and nothing below it is hidden
    wobble[np.minimum(np.abs(CC), np.abs(SS)) > 1e-3] = 0.0
    return want, wobble


def assert_close_spectrum(got, want, rtol, afloor, singular_ok=0.0, extra=None):
    """Bins where either side is non-finite are 0/0 singularities of the epilogue (CC or SS -> 0,
    e.g. two samples, or integer times at Nyquist); allow a fraction ``singular_ok`` of them."""
    want = np.asarray(want)
    fin = np.isfinite(want) & np.isfinite(got)
    mismatch = np.isfinite(want) != np.isfinite(got)
    if extra is not None:  # a bin the oracle itself calls singular may come out as anything
        mismatch &= np.isfinite(want) & np.isfinite(extra)
    assert np.mean(mismatch) <= singular_ok
    if fin.any():
        scale = np.max(np.abs(want[fin]))
        slack = 0.0 if extra is None else np.nan_to_num(extra[fin], nan=np.inf)
        assert np.all(np.abs(got[fin] - want[fin]) <= rtol * np.abs(want[fin]) + afloor * scale + slack)


def random_curve(rng, n):
    kind = rng.integers(0, 4)
    if kind == 0:
        t = np.sort(rng.uniform(0, 50.0, n))
    elif kind == 1:
        t = np.arange(n, dtype=float) * rng.choice([1.0, 0.25, 3.0])          # evenly sampled
    elif kind == 2:
        t = np.sort(rng.uniform(0, 30.0, n)) + rng.choice([2454953.5, -1000.25, 1e5])
    else:
        t = np.sort(np.round(rng.uniform(0, 40.0, n), 1))                     # repeated times
    y = np.sin(2 * np.pi * t / rng.uniform(2.0, 9.0)) + 0.3 * rng.standard_normal(n) + rng.uniform(-5, 5)
    dy = rng.uniform(0.05, 0.5, n)
    return t, y, dy


def test_gls_direct_random_cases(seed=2024):
    rng = np.random.default_rng(seed)
    for case in range(60):
        n = int(rng.choice([3, 5, 17, 64, 255, 256, 257, 300, 513, 777]))
        t, y, dy = random_curve(rng, n)
        nf = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 1025, 2049]))
        span = max(t[-1] - t[0], 1.0)
        f0, delta = rng.uniform(0.05, 2.0) / span, rng.uniform(0.01, 0.3) / span
        freq = f0 + delta * np.arange(nf)
        fit_mean, psd = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        err = dy if rng.integers(0, 3) else None
        got = _cabi.gls_scan(t, y, err, f0, delta, nf, fit_mean, psd)
        want, wobble = exact_power_and_sensitivity(t, y, err, freq, fit_mean, psd)
        # tiny-N / evenly sampled spectra have genuinely (near-)singular bins (CC or SS -> 0), where
        # the epilogue amplifies rounding in the sums by 1/CC: the gate is 1e-6 relative plus what
        # the oracle's own epilogue does with sums moved by 1e-12 (the kernel walks the grid as
        # f_tile + j*delta exactly, numpy rounds every f_j: up to 1 ulp(f) x time span of phase)
        assert_close_spectrum(got, want, 1e-6, 1e-9, extra=wobble), case


def test_gls_long_curves_on_short_grids_random(seed=77):
    """n >= 16384 samples on grids short enough for the scan to cut the samples into parts as well."""
    rng = np.random.default_rng(seed)
    for case in range(10):
        n = int(rng.choice([16384, 20001, 50000, 131072]))
        t, y, dy = random_curve(rng, n)
        nf = int(rng.choice([1, 5, 64, 300, 1000, 5000, 40000]))
        span = max(t[-1] - t[0], 1.0)
        f0, delta = rng.uniform(0.05, 2.0) / span, rng.uniform(0.01, 0.3) / span
        fit_mean, psd = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        err = dy if rng.integers(0, 3) else None
        got = _cabi.gls_scan(t, y, err, f0, delta, nf, fit_mean, psd)
        assert np.array_equal(got, _cabi.gls_scan(t, y, err, f0, delta, nf, fit_mean, psd), equal_nan=True), case
        pick = np.unique(np.linspace(0, nf - 1, 64).astype(int))
        want, wobble = exact_power_and_sensitivity(t, y, err, f0 + delta * pick, fit_mean, psd)
        assert_close_spectrum(got[pick], want, 1e-6, 1e-9, extra=wobble), case


def test_gls_batch_random_ragged(seed=7):
    rng = np.random.default_rng(seed)
    for case in range(8):
        lens = rng.integers(1, 700, size=int(rng.integers(2, 12)))
        curves = [random_curve(rng, int(n)) for n in lens]
        offsets = np.concatenate([[0], np.cumsum(lens)])
        t, y, dy = (np.concatenate([c[i] for c in curves]) for i in range(3))
        nf = int(rng.choice([10, 300, 1500]))
        f0, delta = 0.01, 0.003
        power, amax, argmax = _cabi.gls_scan_batch(t, y, dy, offsets, f0, delta, nf, want_peaks=True)
        for b, (tb, yb, dyb) in enumerate(curves):
            single = _cabi.gls_scan(tb, yb, dyb, f0, delta, nf)
            assert np.array_equal(single, power[b], equal_nan=True)
            if np.any(np.isfinite(single)):
                assert argmax[b] == np.nanargmax(single)


def test_gls_fft_random_cases(seed=99):
    rng = np.random.default_rng(seed)
    for case in range(40):
        n = int(rng.choice([9, 30, 100, 333, 1000]))   # (2-3 samples: every bin is a 0/0 singularity)
        t, y, dy = random_curve(rng, n)
        if t[-1] == t[0]:
            continue
        kw = dict(n=float(rng.choice([1, 2.5, 5])))
        if rng.integers(0, 2):
            span = t[-1] - t[0]
            kw.update(fmin=rng.uniform(0.1, 1.0) / span, fmax=rng.uniform(5.0, 40.0) / span)
        fit_mean = bool(rng.integers(0, 2))
        err = dy if rng.integers(0, 2) else None
        dt_med = np.median(np.diff(t))
        if "fmax" not in kw and (dt_med <= 0 or (t[-1] - t[0]) / dt_med * kw["n"] > 2e5):
            continue                                   # repeated times: upstream's Nyquist is infinite
        with np.errstate(all="ignore"):
            freq, want = so.gls(t, y, err, fit_mean=fit_mean, **kw)
        if freq.size == 0:
            continue
        df = 1.0 / (t[-1] - t[0]) / kw["n"]
        fmin = kw.get("fmin", 0.5 * df)
        got = _cabi.gls_scan_fft(t, y, err, fmin, df, freq.size, fit_mean)
        assert_close_spectrum(got, want, 1e-7, 1e-9, singular_ok=0.02), case


def test_pdm_random_cases(seed=11):
    rng = np.random.default_rng(seed)
    for case in range(60):
        n = int(rng.choice([2, 3, 10, 100, 511, 512, 513, 1500]))
        t, y, _ = random_curve(rng, n)
        if rng.integers(0, 3) == 0:
            t = t - t.mean()                                                    # negative times
        nb, nc = int(rng.integers(1, 12)), int(rng.integers(1, 6))
        n_per = int(rng.choice([1, 7, 64, 65, 200]))
        periods = rng.uniform(0.05, 80.0, n_per)
        periods[: min(3, n_per)] = [1.0, 0.25, 3.0][: min(3, n_per)]          # commensurate
        sigma = np.var(y, ddof=1)
        got = _cabi.pdm_scan(t, y, periods, nb, nc, sigma)
        with np.errstate(all="ignore"):
            want = so.pdm_scan(t, y, periods, nb, nc)
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12, equal_nan=True, err_msg=str(case))


def test_binned_scans_random_cases(seed=17):
    """PDM / AoV / conditional entropy / Gregory-Loredo at the sizes where the launcher changes mode: few
    periods on long curves (sample slices + finishing launch), more than 65 280 samples (16-bit count cells
    force slices), one period, periods spread over device slots of one GPU."""
    rng = np.random.default_rng(seed)
    for case in range(10):
        n = int(rng.choice([700, 5000, 16_385, 33_000, 65_281, 70_001]))
        n_per = int(rng.choice([1, 2, 9, 40, 130]))
        t, y, _ = random_curve(rng, n)
        if rng.integers(0, 3) == 0:
            t = t - t.mean()
        periods = rng.uniform(0.3, 90.0, n_per)
        periods[0] = 2.0
        devices = (0, 0, 0) if rng.integers(0, 4) == 0 else None
        tag = f"case {case}: n={n} periods={n_per} devices={devices}"
        nb, nc = int(rng.integers(2, 9)), int(rng.integers(1, 4))
        with np.errstate(all="ignore"):
            np.testing.assert_allclose(_cabi.pdm_scan(t, y, periods, nb, nc, np.var(y, ddof=1), devices=devices),
                                       so.pdm_scan(t, y, periods, nb, nc), rtol=1e-9, atol=1e-12, err_msg=tag)
            r = int(rng.integers(2, 25))
            np.testing.assert_allclose(_cabi.aov_scan(t, y, periods, r, devices=devices),
                                       so.aov_scan(t, y, periods, r), rtol=1e-8, atol=1e-12, err_msg=tag)
            n_phase, n_mag = int(rng.integers(2, 16)), int(rng.integers(1, 9))
            mb = so.magnitude_bins(y, n_mag)
            np.testing.assert_allclose(_cabi.cond_entropy_scan(t, mb, periods, n_phase, n_mag, devices=devices),
                                       so.cond_entropy_scan(t, mb, periods, n_phase, n_mag), rtol=1e-10, atol=1e-12,
                                       err_msg=tag)
            m, n_off = int(rng.integers(2, 13)), int(rng.choice([1, 4, 8]))
            np.testing.assert_allclose(_cabi.gl_scan(t, periods, m, n_off, devices=devices),
                                       so.gl_scan(t, periods, m, n_off), rtol=1e-10, atol=1e-9, err_msg=tag)


def test_stringlength_random_cases(seed=13):
    rng = np.random.default_rng(seed)
    for case in range(50):
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 191, 193, 1000, 4097, 6000]))
        t, y, _ = random_curve(rng, n)
        if rng.integers(0, 3) == 0:
            t = t - t.mean()
        m = so.stringlength_scale(y) if np.ptp(y) > 0 else np.zeros_like(y)
        n_per = int(rng.choice([1, 5, 33]))
        periods = rng.uniform(0.05, 80.0, n_per)
        periods[: min(3, n_per)] = [1.0, 0.25, 3.0][: min(3, n_per)]
        got = _cabi.stringlength_scan(t, m, periods)
        want = so.stringlength_scan(t, m, periods)
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12, err_msg=str(case))


def test_streamed_samples_in_any_order_random(seed=29):
    """N >= 262 144 handed over out of order (shuffled, reversed, two ordered halves swapped, a few strays): the device
    orders the samples by time first (csrc/timesort.inc, stable) - the result is the oracle's for the time-ordered
    series, duplicates of a time stamp in the caller's order.  Time stamps: uniform, integer (many duplicates), with
    gaps, negative, straddling zero, a narrow range on a large offset."""
    rng = np.random.default_rng(seed)
    for case in range(6):
        n = int(rng.choice([262_144, 263_001, 300_000, 420_000]))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            t = rng.uniform(0.0, float(n), n)
        elif kind == 1:
            t = rng.integers(0, n // 3, n).astype(float)                  # every stamp ~3 times
        elif kind == 2:
            t = rng.uniform(0.0, float(n), n)
            t[t > 0.4 * n] += 0.8 * n                                      # a gap of many periods
            t -= 0.7 * n
        elif kind == 3:
            t = rng.uniform(-0.5 * n, 0.5 * n, n)
            t[rng.integers(0, n, 8)] = rng.choice([0.0, -0.0], 8)
        else:
            t = 2454953.5 + rng.uniform(0.0, 400.0, n)
        how = int(rng.integers(0, 4))
        ts = np.sort(t)
        if how == 0:
            t = ts[rng.permutation(n)]
        elif how == 1:
            t = ts[::-1].copy()
        elif how == 2:
            t = np.concatenate([ts[n // 2:], ts[:n // 2]])
        else:
            t = ts.copy()
            stray = rng.integers(0, n, 5)
            t[stray] = ts[rng.integers(0, n, 5)]
        y = np.sin(2 * np.pi * t / 13.7) + 0.2 * rng.standard_normal(n)
        m = so.stringlength_scale(y)
        base = ts[-1] - ts[0]
        periods = np.concatenate([rng.uniform(base / 180.0, base / 3.0, 4), [base * 1.7, 13.7]])
        back = np.argsort(t, kind="stable")
        got = _cabi.stringlength_scan(t, m, periods)
        want = so.stringlength_scan(t[back], m[back], periods)
        np.testing.assert_allclose(got, want, rtol=1e-9, err_msg=f"case {case} kind {kind} how {how}")


def test_nan_and_inf_inputs_propagate_like_numpy():
    rng = np.random.default_rng(3)
    t, y, dy = random_curve(rng, 200)
    t = np.sort(rng.uniform(0, 50.0, 200))
    periods = np.array([0.7, 3.0, 11.0])
    m = so.stringlength_scale(y)
    for where in ("y", "t"):
        tt, yy, mm = t.copy(), y.copy(), m.copy()
        if where == "y":
            yy[17] = np.nan
            mm[17] = np.nan
        else:
            tt[17] = np.nan
        with np.errstate(all="ignore"):
            want_pdm = so.pdm_scan(tt, yy, periods, 5, 2)
            want_sl = so.stringlength_scan(tt, mm, periods)
            sigma = np.var(yy, ddof=1)
        got_pdm = _cabi.pdm_scan(tt, yy, periods, 5, 2, sigma)
        got_sl = _cabi.stringlength_scan(tt, mm, periods)
        np.testing.assert_allclose(got_pdm, want_pdm, rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(got_sl, want_sl, rtol=1e-9, equal_nan=True)
        got = _cabi.gls_scan(tt, yy, dy, 0.01, 0.003, 50)
        assert np.all(np.isnan(got))
    with np.errstate(all="ignore"):
        odd = np.array([np.inf, np.nan, 0.0, -2.5])
        np.testing.assert_allclose(_cabi.pdm_scan(t, y, odd, 5, 2, np.var(y, ddof=1)),
                                   so.pdm_scan(t, y, odd, 5, 2), rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(_cabi.stringlength_scan(t, m, odd),
                                   so.stringlength_scan(t, m, odd), rtol=1e-9, equal_nan=True)


def test_class_level_calls_random():
    """The three callables end to end (grids, ordering, attributes) against the restated calls."""
    from periodicity_amd.core import TSeries
    from periodicity_amd.phase import PDM, StringLength
    from periodicity_amd.spectral import GLS
    rng = np.random.default_rng(21)
    for case in range(12):
        n = int(rng.choice([40, 150, 600]))
        t = np.sort(rng.uniform(0, 3.0 * n, n))
        dy = rng.uniform(0.05, 0.4, n)
        y = np.sin(2 * np.pi * t / rng.uniform(5.0, 25.0)) + dy * rng.standard_normal(n)
        sig = TSeries(t, y)
        kw = dict(n=float(rng.choice([2, 5])), psd=bool(rng.integers(0, 2)))
        fit_mean = bool(rng.integers(0, 2))
        freq, ref = so.gls(t, y, dy, fit_mean=fit_mean, **kw)
        fft = GLS(method="fft", **kw)(sig, err=dy, fit_mean=fit_mean)
        assert np.array_equal(fft.frequency, freq)
        assert_close_spectrum(fft.values, ref, 1e-7, 1e-10)
        direct = GLS(**kw)(sig, err=dy, fit_mean=fit_mean)
        exact = co.gls_power_exact(t, y, dy, freq, fit_mean, kw["psd"])
        assert_close_spectrum(direct.values, exact, 1e-6, 1e-11)
        assert direct.argmax() == int(np.nanargmax(ref))                       # tier R

        nper = int(rng.choice([33, 200]))
        sub = bool(rng.integers(0, 2))
        pdm = PDM(nb=int(rng.integers(3, 8)), nc=int(rng.integers(1, 4)), n_periods=nper,
                  do_subharmonic=sub)
        got = pdm(sig)
        f_ref, th_ref = so.pdm(t, y, pdm.nb, pdm.nc, n_periods=nper, do_subharmonic=sub)
        assert np.array_equal(got.frequency, f_ref)
        np.testing.assert_allclose(got.values, th_ref, rtol=1e-9)
        dphi = float(rng.choice([0.1, 0.05]))
        sl = StringLength(dphi=dphi, n_periods=nper)(sig)
        f_ref, ell_ref = so.stringlength(t, y, dphi=dphi, n_periods=nper)
        assert np.array_equal(sl.frequency, f_ref)
        np.testing.assert_allclose(sl.values, ell_ref, rtol=1e-9)


def test_scans_are_bitwise_reproducible_run_to_run():
    """No result depends on timing: reductions use fixed orders, LDS atomics only touch private
    counters or feed a total order.  (The FFT path's grid is filled with global fp64 atomics, so it
    is reproducible only to rounding and is not part of this check.)"""
    rng = np.random.default_rng(17)
    t, y, dy = random_curve(rng, 5000)
    t = np.sort(rng.uniform(0, 5000.0, 5000))
    m = so.stringlength_scale(y)
    periods = np.linspace(0.8, 90.0, 700)
    f0, delta, nf = 0.0004, 0.00013, 20000
    ref = (_cabi.gls_scan(t, y, dy, f0, delta, nf), _cabi.pdm_scan(t, y, periods, 5, 2, np.var(y, ddof=1)),
           _cabi.stringlength_scan(t, m, periods))
    for _ in range(4):
        assert np.array_equal(_cabi.gls_scan(t, y, dy, f0, delta, nf), ref[0])
        assert np.array_equal(_cabi.pdm_scan(t, y, periods, 5, 2, np.var(y, ddof=1)), ref[1])
        assert np.array_equal(_cabi.stringlength_scan(t, m, periods), ref[2])


def test_stringlength_fuzz_around_the_fast_path(seed=5):
    """tools/fuzz_sl.py's generator (sizes around the fast kernel's capacity, tied / clustered / offset /
    non-finite time stamps, extreme and special periods: all-ones significands, 1e±200, negative),
    60 cases: parity with the oracle and bitwise repeatability."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_sl
    rng = np.random.default_rng(seed)
    with np.errstate(all="ignore"):
        for case in range(60):
            ok, info = fuzz_sl.one_case(rng)
            assert ok, (case, info["n"], info["kind"], info["periods"], info["got"], info["want"])


def test_gls_shared_time_axis_random(seed=23):
    """Batches on ONE time axis (the bootstrap shape) with random sizes: the two-frequencies-per-lane kernel's tile
    edges (nf mod 128 anywhere), curve counts that leave waves and groups partly empty, odd sample counts (the
    padding row), individual and equal weights, fit_mean / psd: rows against per-curve calls and the peaks."""
    rng = np.random.default_rng(seed)
    for _ in range(4):
        B, n, nf = int(rng.integers(96, 260)), int(rng.integers(40, 500)), int(rng.integers(1, 900))
        t = np.sort(rng.uniform(0, rng.uniform(5, 500), n))
        y = np.sin(2 * np.pi * t / rng.uniform(0.5, 30))[None, :] + rng.standard_normal((B, n))
        dy = rng.uniform(0.1, 0.5, (B, n)) if rng.random() < 0.6 else None
        fit_mean, psd = bool(rng.integers(2)), bool(rng.integers(2))
        f0, delta = rng.uniform(0.001, 0.05), rng.uniform(0.0005, 0.01)
        offsets = np.arange(B + 1) * n
        power, amax, argmax = _cabi.gls_scan_batch(t, y.ravel(), None if dy is None else dy.ravel(), offsets, f0, delta, nf,
                                                   fit_mean, psd, shared_t=True, want_peaks=True)
        # bins with less than half a cycle over the baseline are ill-conditioned (the sinusoid is nearly the constant
        # / a line: the normal equations cancel to ~1e-7 of their terms) - there two kernels that sum in different
        # orders agree to Tier E's 1e-6, not to 1e-9 (seed 20004 of tools/fuzz_gpu.py: f T = 0.015, 9e-9 apart, each
        # within 1e-8 of the long-double sums)
        well = (f0 + delta * np.arange(nf)) * (t[-1] - t[0]) >= 0.5
        for b in rng.choice(B, 6, replace=False):
            single = _cabi.gls_scan(t, y[b], None if dy is None else dy[b], f0, delta, nf, fit_mean, psd)
            atol = 1e-12 * np.nanmax(np.abs(single))
            np.testing.assert_allclose(power[b][well], single[well], rtol=1e-9, atol=atol)
            np.testing.assert_allclose(power[b][~well], single[~well], rtol=1e-6, atol=atol)
            assert argmax[b] == np.nanargmax(power[b]) and amax[b] == np.nanmax(power[b])


def test_supersmoother_random_cases(seed=29):
    """The Supersmoother through both smoothers (generic below 4096 samples, tiled above) on random curves: uneven and
    even sampling, duplicated stamps, Julian offsets, periods from a fraction of the cadence to thousands of baselines,
    the bass control anywhere in [0, 10]."""
    rng = np.random.default_rng(seed)
    for n in (int(rng.integers(5, 60)), int(rng.integers(60, 4096)), int(rng.integers(4096, 9000)), int(rng.integers(9000, 30000))):
        even = rng.random() < 0.25
        t = np.arange(float(n)) * 0.1 if even else np.sort(rng.uniform(0, 0.1 * n, n))
        if rng.random() < 0.5:
            t = t + 2454953.5
        if n > 20 and rng.random() < 0.5:
            t[5:n:7] = t[4:n - 1:7]
        y = np.sin(2 * np.pi * t / 7.3) + 0.3 * rng.standard_normal(n)
        base = max(t[-1] - t[0], 1.0)
        periods = np.concatenate([rng.uniform(0.05, 40.0, 5), base * rng.uniform(0.3, 3000.0, 3), [10.0, 7.3]])
        alpha = float(rng.choice([0.0, rng.uniform(0.1, 10.0)]))
        got = _cabi.supersmoother_scan(t, y, periods, alpha)
        np.testing.assert_allclose(got, so.supersmoother_scan(t, y, periods, alpha), rtol=1e-9, err_msg=f"n={n} even={even} alpha={alpha}")


def test_bglst_random_cases(seed=31):
    """BGLST (parity unpinned by the reference: the oracle is the published marginal likelihood, pinned to scipy's dense
    Gaussian in tests/test_bglst.py): random sizes, grids, priors, reference times, offsets, with and without
    uncertainties, through the C ABI."""
    from periodicity_amd.spectral import BGLST
    rng = np.random.default_rng(seed)
    for case in range(40):
        n = int(rng.choice([4, 5, 17, 100, 511, 1500, 6000]))
        t, y, err = random_curve(rng, n)
        t = t + float(rng.choice([0.0, -37.0, 2454900.5]))
        y = y + rng.uniform(-3, 3) + rng.uniform(-0.05, 0.05) * (t - t[0])
        with_err = bool(rng.integers(0, 2))
        e = err if with_err else np.ones_like(y)
        priors = tuple(float(v) for v in rng.uniform(0.2, 5.0, 3))
        t_ref = float(rng.choice([t[0], t[-1], 0.5 * (t[0] + t[-1]), t[0] - 10.0]))
        nf = int(rng.choice([1, 7, 256, 2048, 2049, 5000]))
        span = max(t[-1] - t[0], 1e-3)
        f0, delta = float(rng.uniform(0.05, 2.0)) / span, float(rng.uniform(0.02, 0.5)) / span
        sc = BGLST._scalars(t, y, e, *priors, t_ref)
        got = _cabi.bglst_scan(t, y, err if with_err else None, f0, delta, nf, sc)
        pick = np.unique(np.concatenate([rng.integers(0, nf, min(nf, 12)), [0, nf - 1]]))
        want = so.bglst_loglik(t, y, err if with_err else None, f0 + delta * pick, *priors, t_ref)
        tol = 1e-9 * max(float(np.abs(want).max()), float(sc[0] * sc[1]))
        assert np.max(np.abs(got[pick] - want)) <= tol, (case, n, nf, np.max(np.abs(got[pick] - want)), tol)
