"""Live pinning of the oracle against the reference's source, executed verbatim.  Runs only
where /root/reference exists (the build container); skipped on the GPU box."""
import numpy as np
import pytest

from oracle import refstub
from oracle import scan_oracle as so
from periodicity_amd.core import TSeries

pytestmark = pytest.mark.skipif(not refstub.available(), reason="reference sources not present")


def curve(n, seed):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, n, n))
    dy = rng.uniform(0.05, 0.2, n)
    return t, 1 + 0.5 * np.sin(2 * np.pi * t / 17.0) + dy * rng.standard_normal(n), dy


def test_reference_tests_pass_through_the_stub():
    spectral, _ = refstub.load()
    t0, ts = 2.5, 0.1
    ls = spectral.GLS(n=1)(TSeries(np.arange(0, t0 + ts, ts)))
    assert ls.frequency[0] == (1 / t0) / 2 and np.round(ls.frequency[-1], 6) == (1 / ts) / 2
    sine = TSeries(values=np.sin((np.arange(100) / 100) * 20 * np.pi))
    assert spectral.GLS()(sine).period_at_highest_peak == 10.0


@pytest.mark.parametrize("kw", [dict(), dict(n=3.5), dict(fmin=0.01, fmax=0.4), dict(psd=True)])
@pytest.mark.parametrize("fit_mean", [True, False])
def test_gls_fft_path_matches_reference_bitwise(kw, fit_mean):
    spectral, _ = refstub.load()
    t, y, dy = curve(700, 3)
    ref = spectral.GLS(**kw)(TSeries(t, y), err=dy, fit_mean=fit_mean)
    freq, power = so.gls(t, y, dy, fit_mean=fit_mean, **kw)
    assert np.array_equal(freq, ref.frequency) and np.array_equal(power, ref.values)


def test_trig_sum_seam_matches_reference_bitwise():
    spectral, _ = refstub.load()
    t, y, dy = curve(500, 4)
    for tt in (t, t + 1234.5):
        S, C = spectral._trig_sum(tt, dy, 0.001, 300, 0.0005)
        S2, C2 = so.trig_sum_fft(tt, dy, 0.001, 300, 0.0005)
        assert np.array_equal(S, S2) and np.array_equal(C, C2)


def test_phase_seams_match_reference_bitwise():
    _, phase = refstub.load()
    t, y, _ = curve(600, 5)
    pdm = phase.PDM(nb=4, nc=3, n_periods=40, cores=1)
    res = pdm(TSeries(t, y))
    freq, theta = so.pdm(t, y, nb=4, nc=3, n_periods=40)
    assert np.array_equal(freq, res.frequency) and np.array_equal(theta, res.values)
    sl = phase.StringLength(cores=1)
    m = so.stringlength_scale(y)
    sl.m = TSeries(t, m)
    for p in (0.7, 3.3, 17.0, 250.0):
        assert sl._stringlength(p) == so.stringlength_one(t, m, p)
