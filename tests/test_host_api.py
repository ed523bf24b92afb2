"""Host-side logic on CPU: the numpy containers and the three callables' grid / attribute /
ordering rules, with the native scan replaced by the oracle (so that only the Python around the
C ABI is under test here — the kernels are tested on the GPU)."""
import numpy as np
import pytest

from oracle import scan_oracle as so
from periodicity_amd import _cabi, phase, spectral
from periodicity_amd.core import FSeries, TSeries


# ---- containers (mirrors /root/reference/tests/test_core.py:7-30) ---------------------------------
def test_time_array_is_always_sorted():
    sig = TSeries([3, 2, 1], [3, 5, 7])
    assert all(sig.time == [1, 2, 3]) and all(sig.values == [7, 5, 3])


def test_input_arrays_with_different_sizes():
    with pytest.raises(ValueError):
        TSeries([1, 2], [1, 2, 3])
    with pytest.raises(ValueError):
        FSeries([1, 2], [1, 2, 3])


def test_dt_baseline_defaults_and_fold():
    sig = TSeries([1, 3, 4], [1, 1, 1])
    assert sig.median_dt == 1.5
    with pytest.raises(AttributeError):
        sig.dt
    assert TSeries(np.arange(10)).baseline == 9
    assert np.array_equal(TSeries(values=[4.0, 5.0]).time, [0, 1])
    assert np.array_equal(TSeries(np.arange(3)).values, np.ones(3))
    folded = TSeries([0.0, 1.5, 2.25, 4.0], [1.0, 2.0, 3.0, 4.0]).fold(2.0)
    assert np.array_equal(folded.time, [0.0, 0.0, 0.125, 0.75])
    assert np.array_equal(folded.values, [1.0, 4.0, 3.0, 2.0])          # stable for equal phases
    sig = TSeries([0.0, 1.0, 2.0], [1.0, np.nan, 3.0])
    assert sig.amax() == 3.0 and sig.argmax() == 2 and sig.max().time[0] == 2.0
    scaled = 0.0 * sig + 1.0
    assert isinstance(scaled, TSeries) and np.array_equal(scaled.time, sig.time)
    cp = sig.copy()
    cp.values = np.zeros(3)
    assert sig.values[0] == 1.0


def test_fseries_sorts_and_finds_peaks():
    fs = FSeries([0.3, 0.1, 0.2, 0.4, 0.5], [1.0, 0.0, 5.0, 2.0, 3.0])
    assert np.array_equal(fs.frequency, [0.1, 0.2, 0.3, 0.4, 0.5])
    assert np.array_equal(fs.values, [0.0, 5.0, 1.0, 2.0, 3.0])
    assert np.allclose(fs.period, 1 / fs.frequency)
    assert fs.pmax() == 1 / 0.2 and fs.fmax() == 0.2 and fs.period_at_highest_peak == 1 / 0.2
    assert fs.find_peaks().attrs["indices"].tolist() == [1]
    assert fs[1:3].frequency.tolist() == [0.2, 0.3] and fs[1] == 5.0
    with np.errstate(divide="ignore"):
        assert np.isinf(FSeries([0.0, 1.0], [1.0, 2.0]).period[0])


def test_fseries_peak_pickers_follow_scipy():
    from scipy.signal import find_peaks
    rng = np.random.default_rng(0)
    f = np.linspace(0.01, 2.0, 400)
    v = np.exp(-0.5 * ((f - 0.5) / 0.03) ** 2) + 0.6 * np.exp(-0.5 * ((f - 1.2) / 0.05) ** 2)
    v += 0.01 * rng.standard_normal(f.size)
    fs = FSeries(f, v)
    idx, props = find_peaks(v, prominence=0.0)
    peaks = fs.find_peaks()
    assert np.array_equal(peaks.attrs["indices"], idx)
    assert np.array_equal(peaks.attrs["prominences"], props["prominences"])
    assert fs.period_at_highest_peak == 1 / f[idx[np.argmax(v[idx])]]
    assert fs.period_at_highest_prominence == 1 / f[idx[np.argmax(props["prominences"])]]
    assert np.array_equal(fs.psort_by_peak(), (1 / f[idx])[np.argsort(v[idx])[::-1]])
    assert np.array_equal(fs.psort_by_prominence(), (1 / f[idx])[np.argsort(props["prominences"])[::-1]])
    lower, upper = fs.periods_at_half_max()
    assert lower < 1 / 0.5 < upper and upper - lower < 0.5
    with_edges = fs.find_peaks(include_edges=True)
    assert with_edges.attrs["indices"][0] == 0 and with_edges.attrs["indices"][-1] == -1
    dips = (-fs).find_peaks()
    assert np.array_equal(fs.find_dips().attrs["indices"], dips.attrs["indices"])
    assert fs.median_df == np.median(np.diff(f)) and abs(fs.df - fs.median_df) < 1e-12


# ---- callables with the C ABI swapped for the oracle -------------------------------------------------
@pytest.fixture
def oracle_backend(monkeypatch):
    def gls_scan(t, y, dy, f0, delta, nf, fit_mean=True, psd=False, j_begin=0, device=None):
        freq = f0 + delta * (j_begin + np.arange(nf))
        return so.gls_power(t, y, dy, freq, delta, f0, fit_mean, psd, sums="exact")

    def gls_scan_batch(t, y, dy, offsets, f0, delta, nf, fit_mean=True, psd=False, shared_t=False,
                       want_power=True, want_peaks=False, j_begin=0, device=None, devices=None):
        rows = [gls_scan(t, y[a:b], dy[a:b], f0, delta, nf, fit_mean, psd)
                for a, b in zip(offsets[:-1], offsets[1:])]
        return None, np.array([np.nanmax(r) for r in rows]), None

    def gls_bootstrap(t, y, dy, picks, f0, delta, nf, fit_mean=True, psd=False, method="direct", device=None,
                      devices=None):
        assert picks.dtype == np.int32 and picks.shape[1:] == t.shape       # indices only cross the ABI
        rows = [gls_scan(t, y[p], None if dy is None else dy[p], f0, delta, nf, fit_mean, psd) for p in picks]
        return np.array([np.nanmax(r) for r in rows]), np.array([np.nanargmax(r) for r in rows])

    monkeypatch.setattr(_cabi, "gls_scan", gls_scan)
    monkeypatch.setattr(_cabi, "gls_scan_batch", gls_scan_batch)
    monkeypatch.setattr(_cabi, "gls_bootstrap", gls_bootstrap)
    monkeypatch.setattr(_cabi, "pdm_scan",
                        lambda t, x, p, nb, nc, sigma, device=None, devices=None: so.pdm_scan(t, x, np.asarray(p), nb, nc))
    monkeypatch.setattr(_cabi, "stringlength_scan",
                        lambda t, m, p, device=None, devices=None: so.stringlength_scan(t, m, np.asarray(p)))
    monkeypatch.setattr(_cabi, "aov_scan",
                        lambda t, x, p, n_bins, device=None, devices=None: so.aov_scan(t, x, np.asarray(p), n_bins))
    monkeypatch.setattr(_cabi, "gl_scan",
                        lambda t, p, m, n_off, device=None, devices=None: so.gl_scan(t, np.asarray(p), m, n_off))
    monkeypatch.setattr(_cabi, "cond_entropy_scan",
                        lambda t, mb, p, n_phase, n_mag, device=None, devices=None: so.cond_entropy_scan(t, mb, np.asarray(p), n_phase, n_mag))


def curve(n=400, seed=2):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, n, n))
    dy = rng.uniform(0.05, 0.2, n)
    return t, 1 + 0.5 * np.sin(2 * np.pi * t / 17.0) + dy * rng.standard_normal(n), dy


def test_gls_grid_attributes_and_aliases(oracle_backend, golden_dir):
    assert spectral.LombScargle is spectral.GLS
    t, y, dy = curve()
    for kw in (dict(), dict(n=3.5), dict(fmin=0.01, fmax=0.4), dict(n=1)):
        gls = spectral.GLS(**kw)
        ls = gls(TSeries(t, y), err=dy)
        freq, _, _ = so.gls_grid(t, kw.get("n", 5), kw.get("fmin"), kw.get("fmax"))
        assert np.array_equal(ls.frequency, freq) and np.array_equal(gls.frequency, freq)
        assert gls.periodogram is ls and gls.signal.size == t.size and gls.err is dy
    g = np.load(f"{golden_dir}/g2_sine100.npz")
    ls = spectral.GLS()(g["values"])                    # raw array -> TSeries(values=...)
    assert np.array_equal(ls.frequency, g["frequency"]) and ls.period_at_highest_peak == 10.0
    assert np.array_equal(spectral.GLS()(TSeries(t, y)).frequency, so.gls_grid(t)[0])
    copy = gls.copy()
    assert copy is not gls and copy.n == gls.n


def test_gls_bootstrap_reproduces_the_reference_draws(oracle_backend, golden_dir):
    g = np.load(f"{golden_dir}/g6_bootstrap.npz")
    gls = spectral.GLS()
    gls(TSeries(g["t"], g["y"]), err=g["dy"])
    reps = gls.bootstrap(20, random_seed=42)            # same rng.integers(0, n, n) stream
    np.testing.assert_allclose(reps, g["replicates_exact"], rtol=1e-8)
    assert gls.fap(0.3) == float(g["fap_at_0p3_exact"])
    win = gls.window()
    assert win.size == gls.frequency.size


def test_pdm_and_stringlength_host_rules(oracle_backend, golden_dir):
    g = np.load(f"{golden_dir}/g7_pdm.npz")
    sig = TSeries(g["t"], g["y"])
    res = phase.PDM(p_min=1.0, p_max=60.0, n_periods=200, do_subharmonic=True)(sig)
    assert np.array_equal(res.frequency, g["frequency_sub"])
    assert np.array_equal(res.values, g["theta_call_sub"])
    pdm = phase.PDM(n_periods=None, p_min=2.0, p_max=50.0)
    res = pdm(sig)
    assert res.size == int((1 / 2.0 - 1 / 50.0) * sig.baseline + 1)
    assert pdm.cores is None and phase.StringLength(cores=10 ** 6).cores == phase.MAX_CORES
    g = np.load(f"{golden_dir}/g8_stringlength.npz")
    sl = phase.StringLength(n_periods=200)
    res = sl(TSeries(g["t"], g["y"]))
    assert np.array_equal(sl.m.values, g["m"])
    assert np.array_equal(res.values[::-1], g["ell"])
    assert np.all(np.diff(res.frequency) > 0)


def test_no_silent_cpu_fallback():
    if _cabi.device_count() > 0:
        pytest.skip("a GPU is present")
    t, y, dy = curve(50)
    with pytest.raises(RuntimeError):
        spectral.GLS()(TSeries(t, y), err=dy)
    with pytest.raises(RuntimeError):
        phase.PDM(n_periods=8)(TSeries(t, y))
    with pytest.raises(RuntimeError):
        phase.StringLength(n_periods=8)(TSeries(t, y))


# ---- GLS.model against the reference's own output (golden G9, spectral.py:169-204) -----------------
def test_gls_model_matches_reference_golden(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "g9_model.npz"))
    for tag, err in (("err", g["dy"]), ("noerr", None)):
        gls = spectral.GLS()
        gls.signal = TSeries(g["t"], g["y"])                      # what __call__ leaves behind
        gls.err = np.ones_like(g["y"]) if err is None else err    # (spectral.py:99-101)
        for f0, want in zip(g["f0_" + tag], g["yf_" + tag]):
            fit = gls.model(g["tf"], float(f0))
            assert isinstance(fit, TSeries)
            np.testing.assert_array_equal(fit.time, g["tf_sorted_" + tag])
            np.testing.assert_allclose(fit.values, want, rtol=1e-10, atol=1e-12)


def test_aov_and_entropy_host_rules(oracle_backend):
    """The two TODO scans of phase.py:11-15 behind PDM-shaped classes: grid, ordering, attributes."""
    t, y, _ = curve(600, 4)
    sig = TSeries(t, y)
    aov = phase.AOV(n_bins=8, p_min=5.0, p_max=40.0, n_periods=300)
    res = aov(sig)
    assert np.array_equal(aov.periods, np.linspace(5.0, 40.0, 300)) and res.size == 300
    assert np.all(np.diff(res.frequency) > 0)                                   # FSeries sorts ascending
    assert abs(res.period[np.argmax(res.values)] - 17.0) < 0.5
    ce = phase.ConditionalEntropy(n_phase=8, n_mag=4, p_min=5.0, p_max=40.0, n_periods=300)
    res = ce(sig)
    assert set(np.unique(ce.mag_bin)) <= {0.0, 1.0, 2.0, 3.0}
    assert np.array_equal(ce.mag_bin, so.magnitude_bins(y, 4))
    assert abs(res.period[np.argmin(res.values)] - 17.0) < 0.5
    # an all-in-one-bin fold has zero conditional entropy only if the magnitudes are constant there
    assert so.cond_entropy(t, np.zeros_like(t), 7.0, 8, 4) == 0.0


def test_one_precedence_rule_for_device_and_devices():
    """`devices`, when given, wins over `device` everywhere (GLS.__call__ and bootstrap used to disagree)."""
    assert _cabi.pick_device(None, None) is None
    assert _cabi.pick_device(3, None) == 3
    assert _cabi.pick_device(0, (1,)) == 1 and _cabi.pick_device(None, (2, 5)) == 2
    assert _cabi.pick_device(4, ()) == 4


def test_package_imports_and_runs_host_logic_without_torch():
    """Nothing in the package needs torch (the one-rank-per-GPU launcher glue lives in tools/torchrun_sharded.py):
    the callables, the containers and the ctypes binding import with torch made unimportable."""
    import subprocess
    import sys
    code = ("import sys; sys.modules['torch'] = None\n"
            "import periodicity_amd\n"
            "from periodicity_amd import _cabi, core, phase, spectral\n"
            "from tools import torchrun_sharded as distributed\n"
            "assert 'torch' not in [m for m in sys.modules if sys.modules[m] is not None]\n"
            "print(distributed.slab_bounds(10, 3, 2), spectral.LombScargle.__name__, phase.AOV.__name__)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         cwd=__import__("os").path.dirname(__import__("os").path.dirname(__file__)))
    assert out.returncode == 0, out.stderr
    assert "(8, 10, 4) GLS AOV" in out.stdout
