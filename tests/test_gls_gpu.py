"""GPU parity tests of the Lomb-Scargle path, all through the C ABI (ctypes).

Tier E: power within 1e-6 relative of the long-double exact-sum oracle at every well-conditioned
bin.  Tier R: peak bin identical to the unmodified reference (golden ``power_ref``)."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import scan_oracle as so
from periodicity_amd import _cabi
from periodicity_amd.core import FSeries, TSeries
from periodicity_amd.spectral import GLS, LombScargle

pytestmark = pytest.mark.gpu
RTOL = 1e-6  # BASELINE.json north_star: "<= 1e-6 relative in fp64"


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def assert_tier_e(power, exact, rtol=RTOL, floor=1e-13):
    """Relative parity on every bin whose exact power is above ``floor`` x the spectrum's maximum
    (bins that are exact zeros / 0-over-0 singularities carry only rounding noise in any
    implementation, the oracle's included)."""
    exact = np.asarray(exact)
    ok = np.abs(exact) > floor * np.nanmax(np.abs(exact))
    assert ok.mean() > 0.99
    rel = np.abs(power[ok] - exact[ok]) / np.abs(exact[ok])
    assert rel.max() <= rtol, rel.max()
    return rel.max()


def synth(n, seed, period=37.3, t_offset=0.0):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, float(n), n)) + t_offset
    dy = rng.uniform(0.05, 0.2, n)
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / period) + dy * rng.standard_normal(n)
    return t, y, dy


def test_device_is_gfx950():
    info = _cabi.device_info(0)
    assert "gfx950" in info["name"] and info["cu_count"] == 256


def test_reference_test_default_grid():
    # /root/reference/tests/test_spectral.py:7-24, verbatim assertions, through the GPU class
    t0, ts = 2.5, 0.1
    fs, f0 = 1 / ts, 1 / t0
    ls = GLS(n=1)(TSeries(np.arange(0, t0 + ts, ts)))
    freq = ls.frequency
    assert sorted(freq) == list(freq)
    assert freq[0] == f0 / 2
    assert np.round(freq[-1], 6) == fs / 2
    assert np.max(np.abs(np.diff(freq) - f0)) < 1e-10
    assert ls.values.shape == freq.shape


def test_reference_test_can_find_periods(golden_dir):
    # /root/reference/tests/test_spectral.py:27-31
    sine = TSeries(values=np.sin((np.arange(100) / 100) * 20 * np.pi))
    ls = LombScargle()(sine)
    assert ls.period_at_highest_peak == 10.0
    g = load(golden_dir, "g2_sine100")
    assert np.array_equal(ls.frequency, g["frequency"])
    assert ls.argmax() == int(g["argmax"]) == 49
    ok = np.abs(g["power_exact"]) > 1e-20
    ok[-1] = False  # Nyquist bin of integer times: 0/0
    rel = np.abs(ls.values[ok] - g["power_exact"][ok]) / np.abs(g["power_exact"][ok])
    assert rel.max() <= RTOL


@pytest.mark.parametrize("fit_mean", [True, False])
@pytest.mark.parametrize("psd", [False, True])
def test_spotted_star_all_flags(golden_dir, fit_mean, psd):
    g = load(golden_dir, "g3_spotted_star")
    gls = GLS(psd=psd)
    ls = gls(TSeries(g["t"], g["y"]), err=g["dy"], fit_mean=fit_mean)
    tag = f"fm{int(fit_mean)}_psd{int(psd)}"
    assert np.array_equal(ls.frequency, g["frequency"])
    assert_tier_e(ls.values, g["power_exact_" + tag])
    ref = g["power_ref_" + tag]
    assert ls.argmax() == int(np.nanargmax(ref))                       # tier R: same peak bin
    assert ls.pmax() == FSeries(g["frequency"], ref).pmax()            # identical double
    assert gls.periodogram is ls and gls.signal.size == g["t"].size
    assert np.array_equal(gls.err, g["dy"]) and np.array_equal(gls.frequency, g["frequency"])


def test_spotted_star_without_errors(golden_dir):
    g = load(golden_dir, "g3_spotted_star")
    gls = GLS()
    ls = gls(TSeries(g["t"], g["y"]))
    assert_tier_e(ls.values, g["power_exact_noerr"])
    assert ls.argmax() == int(np.nanargmax(g["power_ref_noerr"]))
    assert np.array_equal(gls.err, np.ones_like(g["y"]))               # spectral.py:99-101


@pytest.mark.parametrize("n", [1000, 5000])
def test_synthetic_power_and_seams(golden_dir, n):
    g = load(golden_dir, f"g4_synth{n}")
    t, y, dy, f = g["t"], g["y"], g["dy"], g["frequency"]
    ls = GLS()(TSeries(t, y), err=dy)
    assert np.array_equal(ls.frequency, f)
    assert_tier_e(ls.values, g["power_exact"])
    assert ls.argmax() == int(np.nanargmax(g["power_ref"]))
    # seam level: _trig_sum(t, w*y, df, nf, fmin) and _trig_sum(t, w, 2df, nf, 2fmin)
    w, yc, _ = so.gls_weights(y, dy, True)
    f0, delta, nf = _cabi.grid_params(f)
    for h, sname, cname, scale in ((w * yc, "Sh_exact", "Ch_exact", 1), (w, "S_exact", "C_exact", 1),
                                   (w, "S2_exact", "C2_exact", 2)):
        S, C = _cabi.trig_sums(t, h, scale * f0, scale * delta, nf)
        amp = np.max(np.hypot(g[sname], g[cname]))
        assert np.max(np.abs(S - g[sname])) <= 1e-9 * amp
        assert np.max(np.abs(C - g[cname])) <= 1e-9 * amp


def test_window_and_model(golden_dir):
    g = load(golden_dir, "g5_window")
    gls = GLS()
    gls(TSeries(g["t"], g["y"]), err=g["dy"])
    win = gls.window()
    assert np.array_equal(win.frequency, g["frequency"])
    assert_tier_e(win.values, g["power_exact"], floor=1e-9)
    assert win.argmax() == int(np.nanargmax(g["power_ref"]))
    fit = gls.model(g["t"], 1 / 37.3)
    assert fit.size == g["t"].size and abs(np.std(fit.values) - 0.5 / np.sqrt(2)) < 0.05


def test_bootstrap_fap_fal(golden_dir):
    g = load(golden_dir, "g6_bootstrap")
    gls = GLS()
    gls(TSeries(g["t"], g["y"]), err=g["dy"])
    reps = gls.bootstrap(20, random_seed=42)
    np.testing.assert_allclose(reps, g["replicates_exact"], rtol=RTOL)
    np.testing.assert_allclose(reps, g["replicates_ref"], rtol=0, atol=5e-3)  # FFT-path error at N=300
    assert gls.fap(0.3) == float(g["fap_at_0p3_exact"])
    np.testing.assert_allclose(gls.fal(0.1), float(g["fal_at_0p1_exact"]), rtol=RTOL)


@pytest.mark.parametrize("n_boot,with_dy", [(5, True), (130, True), (200, False)])
def test_bootstrap_by_index_equals_the_expanded_batch(n_boot, with_dy):
    """pdc_gls_bootstrap gathers y[picks], dy[picks] on the device (spectral.py:146-148): bit for bit what the
    batched scan gives on the resampled arrays built on the host - below and above the 96 curves from which the
    shared-trigonometry kernel takes over, with individual and with equal weights, and through the FFT path."""
    t, y, dy = synth(700, 61, period=9.0)
    dy = dy if with_dy else None
    rng = np.random.default_rng(3)
    picks = rng.integers(0, t.size, (n_boot, t.size)).astype(np.int32)
    freq = np.arange(0.002, 0.6, 0.00043)
    f0, delta, nf = _cabi.grid_params(freq)
    offsets = np.arange(n_boot + 1, dtype=np.int64) * t.size
    _, want, want_arg = _cabi.gls_scan_batch(t, y[picks].ravel(), None if dy is None else dy[picks].ravel(), offsets,
                                             f0, delta, nf, shared_t=True, want_power=False, want_peaks=True)
    amax, arg = _cabi.gls_bootstrap(t, y, dy, picks, f0, delta, nf)
    assert np.array_equal(amax, want) and np.array_equal(arg, want_arg)
    # three device slots: a slot's group may fall below the 96 curves of the shared-trigonometry kernel and take
    # the per-curve one - same sums in another order
    amax2, arg2 = _cabi.gls_bootstrap(t, y, dy, picks, f0, delta, nf, devices=(0, 0, 0))
    np.testing.assert_allclose(amax2, want, rtol=1e-12)
    assert np.array_equal(arg2, want_arg)
    # one replicate against the long-double oracle on the resampled curve
    b = n_boot // 2
    exact = np.asarray(co.gls_power_exact(t, y[picks[b]], None if dy is None else dy[picks[b]], freq))
    assert arg[b] == int(np.argmax(exact)) and abs(amax[b] / exact.max() - 1) < RTOL
    _, fwant, _ = _cabi.gls_scan_fft_batch(t, y[picks].ravel(), None if dy is None else dy[picks].ravel(), offsets,
                                           f0, delta, nf, shared_t=True, want_power=False, want_peaks=True)
    famax, _ = _cabi.gls_bootstrap(t, y, dy, picks, f0, delta, nf, method="fft")
    assert np.array_equal(famax, fwant)


def test_bootstrap_by_index_rejects_bad_picks_and_handles_empty():
    t, y, dy = synth(50, 2)
    picks = np.zeros((3, 50), dtype=np.int32)
    picks[1, 7] = 50
    with pytest.raises(ValueError):
        _cabi.gls_bootstrap(t, y, dy, picks, 0.01, 0.01, 20)
    picks[1, 7] = -1
    with pytest.raises(ValueError):
        _cabi.gls_bootstrap(t, y, dy, picks, 0.01, 0.01, 20)
    with pytest.raises(ValueError):
        _cabi.gls_bootstrap(t, y, dy, np.zeros((3, 49), dtype=np.int32), 0.01, 0.01, 20)
    amax, arg = _cabi.gls_bootstrap(t, y, dy, np.zeros((0, 50), dtype=np.int32), 0.01, 0.01, 20)
    assert amax.size == 0 and arg.size == 0
    gls = GLS()
    gls(TSeries(t, y), err=dy)
    assert gls.bootstrap(0).size == 0


def test_batch_ragged_equals_single_and_peaks():
    rng = np.random.default_rng(11)
    lens = [1, 2, 255, 256, 257, 1000, 3, 777]
    curves = [synth(n, 100 + i, period=5.0 + i) for i, n in enumerate(lens)]
    offsets = np.concatenate([[0], np.cumsum(lens)])
    t = np.concatenate([c[0] for c in curves])
    y = np.concatenate([c[1] for c in curves])
    dy = np.concatenate([c[2] for c in curves])
    freq = np.arange(0.0007, 0.5, 0.00037)
    f0, delta, nf = _cabi.grid_params(freq)
    power, amax, argmax = _cabi.gls_scan_batch(t, y, dy, offsets, f0, delta, nf,
                                               want_power=True, want_peaks=True)
    for b, (tb, yb, dyb) in enumerate(curves):
        single = _cabi.gls_scan(tb, yb, dyb, f0, delta, nf)
        assert np.array_equal(single, power[b], equal_nan=True)
        if np.all(np.isnan(single)):
            assert argmax[b] == -1 and np.isnan(amax[b])
        else:
            assert argmax[b] == np.nanargmax(single) and amax[b] == np.nanmax(single)
    _, amax2, argmax2 = _cabi.gls_scan_batch(t, y, dy, offsets, f0, delta, nf,
                                             want_power=False, want_peaks=True)
    assert np.array_equal(amax, amax2, equal_nan=True) and np.array_equal(argmax, argmax2)
    big = curves[5]
    exact = co.gls_power_exact(big[0], big[1], big[2], freq)
    assert_tier_e(power[5], exact)
    del rng


def test_shared_time_axis_batch():
    t, y, dy = synth(400, 5)
    rng = np.random.default_rng(3)
    picks = rng.integers(0, 400, (6, 400))
    offsets = np.arange(7) * 400
    freq = np.arange(0.001, 0.4, 0.0009)
    f0, delta, nf = _cabi.grid_params(freq)
    power, _, _ = _cabi.gls_scan_batch(t, y[picks].ravel(), dy[picks].ravel(), offsets, f0, delta,
                                       nf, shared_t=True)
    for b in range(6):
        single = _cabi.gls_scan(t, y[picks[b]], dy[picks[b]], f0, delta, nf)
        assert np.array_equal(single, power[b])


@pytest.mark.parametrize("with_dy,fit_mean,psd", [(True, True, False), (True, False, True),
                                                  (False, True, False), (False, False, True)])
def test_shared_time_axis_many_curves_share_the_trigonometry(with_dy, fit_mean, psd):
    """>= 96 curves on one time axis run through gls_shared_kernel (sin/cos computed once per
    workgroup, per-curve weights from the scalar cache): rows agree with per-curve calls to
    rounding and with the long-double sums to Tier E; odd sample counts exercise the padding row."""
    rng = np.random.default_rng(11)
    B, n, nf = 131, 257, 700
    t = np.sort(rng.uniform(0, 40, n))
    y = np.sin(2 * np.pi * t / 3.3)[None, :] + rng.standard_normal((B, n))
    dy = rng.uniform(0.1, 0.5, (B, n)) if with_dy else None
    offsets = np.arange(B + 1) * n
    f0, delta = 0.01, 0.0021
    freq = f0 + delta * np.arange(nf)
    power, amax, argmax = _cabi.gls_scan_batch(t, y.ravel(), None if dy is None else dy.ravel(),
                                               offsets, f0, delta, nf, fit_mean, psd, shared_t=True,
                                               want_peaks=True)
    for b in range(B):
        single = _cabi.gls_scan(t, y[b], None if dy is None else dy[b], f0, delta, nf, fit_mean, psd)
        np.testing.assert_allclose(power[b], single, rtol=1e-10, atol=1e-13 * np.max(single))
        assert argmax[b] == np.nanargmax(power[b]) and amax[b] == np.nanmax(power[b])
    for b in (0, 77, B - 1):
        exact = co.gls_power_exact(t, y[b], None if dy is None else dy[b], freq, fit_mean, psd)
        assert_tier_e(power[b], exact)
    # peaks-only mode (what GLS.bootstrap asks for) gives the same maxima
    _, amax2, argmax2 = _cabi.gls_scan_batch(t, y.ravel(), None if dy is None else dy.ravel(), offsets,
                                             f0, delta, nf, fit_mean, psd, shared_t=True,
                                             want_power=False, want_peaks=True)
    assert np.array_equal(amax, amax2) and np.array_equal(argmax, argmax2)


def test_short_grid_over_a_long_curve_splits_the_samples():
    """Few frequencies x many samples (a zoom on one peak, a small slab of a sharded grid): the scan cuts the
    samples into parts as well, every part leaves six sums per frequency and a finishing kernel adds them in a
    fixed order.  Same answers as the exact sums, as an unsplit call (a two-curve batch never splits), the
    same bits on every call, and the device-side peak reduction follows."""
    t, y, dy = synth(60_000, 31, period=12.3)
    for nf, fit_mean, psd in ((3000, True, False), (700, False, True), (37, True, True), (20_000, True, False)):
        delta = 2e-7                                                  # a zoom on the peak
        f0 = 1 / 12.3 - (nf // 2) * delta
        freq = f0 + np.arange(nf) * delta                             # (the device's own fill rule)
        got = _cabi.gls_scan(t, y, dy, f0, delta, nf, fit_mean, psd)
        assert np.array_equal(got, _cabi.gls_scan(t, y, dy, f0, delta, nf, fit_mean, psd))
        pick = np.unique(np.linspace(0, nf - 1, 300).astype(int))        # (the exact sums cost N x nf in long double)
        assert_tier_e(got[pick], co.gls_power_exact(t, y, dy, freq[pick], fit_mean, psd))
        offsets = np.array([0, t.size, 2 * t.size])
        twice, amax, argmax = _cabi.gls_scan_batch(np.tile(t, 2), np.tile(y, 2), np.tile(dy, 2), offsets, f0, delta, nf,
                                                   fit_mean, psd, want_power=True, want_peaks=True)
        np.testing.assert_allclose(got, twice[0], rtol=1e-9, atol=1e-14 * np.nanmax(got))   # (other summation order)
        _, a1, j1 = _cabi.gls_scan_batch(t, y, dy, offsets[:2], f0, delta, nf, fit_mean, psd, want_power=False,
                                         want_peaks=True)
        assert j1[0] == int(np.nanargmax(got)) and a1[0] == got[j1[0]]


def test_slabs_tile_the_grid_bitwise():
    t, y, dy = synth(3000, 21)
    freq = np.arange(0.0001, 0.9, 0.00011)
    f0, delta, nf = _cabi.grid_params(freq)
    full = _cabi.gls_scan(t, y, dy, f0, delta, nf)
    cuts = [0, 1, 2047, 2048, 5000, nf]
    parts = [_cabi.gls_scan(t, y, dy, f0, delta, b - a, j_begin=a) for a, b in zip(cuts, cuts[1:])]
    stitched = np.concatenate(parts)
    # a slab starts its rotation recurrence at its own first frequency, so agreement is to
    # rounding, not bitwise; the first bin of each slab is seeded directly in both runs
    np.testing.assert_allclose(stitched, full, rtol=1e-9, atol=1e-15)
    multi = _cabi.gls_scan_multi(t, y, dy, f0, delta, nf, devices=(0,))
    assert np.array_equal(multi, full)
    ls = GLS(fmin=freq[0], fmax=freq[-1], n=1 / ((t[-1] - t[0]) * 0.00011), devices=(0,))
    assert ls(TSeries(t, y), err=dy).size > 0


def test_rccl_all_gather_path_on_one_device():
    # the collective of pdc_gls_scan_multi (ncclCommInitAll + grouped in-place all-gather) is
    # normally skipped for a single device; force it in a child process (the switch is read once)
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PDC_FORCE_RCCL="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_single_device_check.py")],
                         cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert "all-gather equal: True" in out.stdout, out.stdout[-400:] + out.stderr[-400:]


def test_plan_falls_back_to_copies_when_the_communicators_cannot_be_built():
    """plan_build had no fallback - a failing ncclCommInitAll killed the whole `--gpus 8` run.
    PDC_FORCE_RCCL_FAIL=1 injects the failure: the plan is built all the same, warns on stderr, keeps the reason
    (pdc_gls_plan_init_error) and exchanges by device-to-device copies; results unchanged."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PDC_FORCE_RCCL="1", PDC_FORCE_RCCL_FAIL="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_single_device_check.py"), "fail"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert "fallback after a failed ncclCommInitAll equal: True" in out.stdout, out.stdout[-600:] + out.stderr[-600:]
    assert "WARNING: ncclCommInitAll failed" in out.stderr


def test_persistent_plan_repeated_scans_and_validation():
    """pdc_gls_plan_*: buffers, streams and communicators created once; scans only enqueue, outputs are
    double-buffered; every scan of a sequence comes back bit-identical to the one-shot call."""
    t, y, dy = synth(3000, 77)
    grids = [np.arange(0.002, 0.9, 0.00031), np.arange(0.001, 0.5, 0.0007), np.arange(0.002, 0.9, 0.00031)]
    plan = _cabi.GlsPlan([0], n_max=4000, nf_max=max(g.size for g in grids))
    plan.upload(t, y, dy)
    for g in grids:                      # back-to-back enqueues, one wait at the end of each check
        f0, delta, nf = _cabi.grid_params(g)
        plan.scan(f0, delta, nf)
        plan.scan(f0, delta, nf)         # second generation while the first may still be in flight
        got = plan.download()
        assert np.array_equal(got, _cabi.gls_scan(t, y, dy, f0, delta, nf))
        assert plan.kernel_ms() > 0
    plan.upload(t[:1000], y[:1000], None)            # new light curve, same plan
    f0, delta, nf = _cabi.grid_params(grids[1])
    plan.scan(f0, delta, nf, fit_mean=False, psd=True)
    assert np.array_equal(plan.download(), _cabi.gls_scan(t[:1000], y[:1000], None, f0, delta, nf, False, True))
    with pytest.raises(ValueError):
        plan.upload(np.arange(5000.0), np.ones(5000))            # more samples than the plan holds
    with pytest.raises(ValueError):
        plan.scan(0.1, 0.1, 10 * max(g.size for g in grids))      # more frequencies than the plan holds
    plan.close()
    n_dev = _cabi.device_count()
    with pytest.raises(ValueError):
        _cabi.GlsPlan([n_dev], 10, 10)                           # not one of the visible devices
    with pytest.raises(ValueError):
        _cabi.GlsPlan([0, 0], 10, 10)                            # listed twice
    with pytest.raises(ValueError):
        _cabi.gls_scan_multi(t, y, dy, 0.1, 0.1, 8, devices=(n_dev + 3,))


def test_sharded_gls_gathers_the_device_slab_over_rccl():
    """tools/torchrun_sharded.py:sharded_gls under a one-rank nccl (RCCL) group: the slab is
    written by pdc_gls_scan_dev into a device tensor and all-gathered there."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_single_device_check.py"), "torch"],
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert "sharded_gls equal: True" in out.stdout, out.stdout[-400:] + out.stderr[-600:]


def test_large_time_offset_is_harmless():
    # Kepler-style barycentric dates: f*t spans ~1e7 cycles, the phase must not lose bits
    t, y, dy = synth(2000, 33)
    freq = np.arange(0.001, 2.0, 0.0021)
    f0, delta, nf = _cabi.grid_params(freq)
    p0 = _cabi.gls_scan(t, y, dy, f0, delta, nf)
    p1 = _cabi.gls_scan(t + 2454953.5, y, dy, f0, delta, nf)
    np.testing.assert_allclose(p1, p0, rtol=1e-7)
    to = t + 2454953.5
    S, C = _cabi.trig_sums(to, dy ** -2.0, f0, delta, nf)
    Se, Ce = co.trig_sums_exact(to, dy ** -2.0, freq)
    amp = np.max(np.hypot(Se, Ce))
    assert np.max(np.abs(S - Se)) <= 1e-9 * amp and np.max(np.abs(C - Ce)) <= 1e-9 * amp


def test_edge_cases():
    t, y, dy = synth(50, 1)
    assert _cabi.gls_scan(t, y, dy, 0.01, 0.01, 0).size == 0                # empty grid
    p = _cabi.gls_scan(t, y, dy, 0.3, 0.0, 1)                                  # one frequency
    assert p.shape == (1,) and np.isfinite(p[0])
    ynan = y.copy()
    ynan[7] = np.nan                                                           # NaN propagates
    assert np.all(np.isnan(_cabi.gls_scan(t, ynan, dy, 0.01, 0.01, 64)))
    with pytest.raises(ValueError):
        _cabi.gls_scan(t, y[:-1], dy, 0.01, 0.01, 8)
    with pytest.raises(ValueError):
        GLS()(TSeries(t, y), err=dy[:-1])
    with pytest.raises(ValueError):
        TSeries([1, 2], [1, 2, 3])
    raw = GLS()(y)                                                             # spectral.py:86-87
    assert np.array_equal(raw.frequency, GLS()(TSeries(values=y)).frequency)
    one = _cabi.gls_scan(t[:1], y[:1], dy[:1], 0.01, 0.01, 4)                # single sample: 0/0
    assert one.shape == (4,)


def test_entry_points_are_reentrant_across_python_threads():
    # ctypes drops the GIL during a call; calls on one device serialise on its workspace lock
    import threading
    cases = [synth(800 + 37 * i, 50 + i) for i in range(6)]
    freq = np.arange(0.002, 0.45, 0.0007)
    f0, delta, nf = _cabi.grid_params(freq)
    want = [_cabi.gls_scan(t, y, dy, f0, delta, nf) for t, y, dy in cases]
    got = [None] * len(cases)

    def work(i):
        for _ in range(5):
            t, y, dy = cases[i]
            got[i] = _cabi.gls_scan(t, y, dy, f0, delta, nf)
            _cabi.pdm_scan(t, y, np.linspace(1.0, 30.0, 70), 5, 2, np.var(y, ddof=1))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_full_size_c2_properties():
    """BASELINE configs[1] at full size (1e11 pairs): a random subset of bins against the exact
    oracle, plus invariances the domain offers (amplitude scaling, slab consistency)."""
    co.tune_threads()        # the host is shared: the thread count the C checker runs fastest with, measured once
    n, nf = 100_000, 1_000_000
    t, y, dy = synth(n, 20241010)
    df = 1.0 / (t[-1] - t[0]) / 5
    fmin = 0.5 * df
    freq = np.arange(fmin, fmin + (nf - 1.5) * df + df, df)
    assert freq.size == nf
    f0, delta, _ = _cabi.grid_params(freq)
    power = _cabi.gls_scan(t, y, dy, f0, delta, nf)
    assert np.all(np.isfinite(power)) and power.min() >= -1e-12 and power.max() <= 1 + 1e-9
    peak = int(np.argmax(power))
    assert abs(1 / freq[peak] - 37.3) < 0.01
    # >= 4096 bins against the long-double oracle, stratified over the kernel's tiles (2048 frequencies at
    # this shape; other tile shapes divide it): the first and last frequency of EVERY tile - where a thread's
    # recurrence starts and where it has run longest -, the last (partial) tile, and 7 random bins inside
    # each tile
    rng = np.random.default_rng(0)
    tile = 2048
    starts = np.arange(0, nf, tile)
    ends = np.minimum(starts + tile, nf) - 1
    inside = (starts[:, None] + rng.integers(0, tile, (starts.size, 7))).ravel()
    pick = np.unique(np.concatenate([starts, ends, inside[inside < nf], [0, nf - 1, peak, peak + 1],
                                     np.arange(starts[-1], nf, 37)]))
    assert pick.size >= 4096
    exact = co.gls_power_exact(t, y, dy, freq[pick])
    rel = np.abs(power[pick] - exact) / np.abs(exact)
    assert rel.max() <= RTOL, rel.max()
    assert peak == pick[np.argmax(exact)]                 # the oracle's maximum over the sampled bins is the peak
    # round 6 - EVERY bin: the double-precision direct sums (oracle_gls_sums_f64: pair-by-pair evaluation, exact
    # cycle reduction; pinned to the long-double sums on goldens in tests/test_oracle_golden.py) first re-proved
    # on the 4400 stratified bins against the long-double oracle, then trusted on all 1e6 (1e11 pairs on the host)
    fast = np.asarray(co.gls_power_f64(t, y, dy, freq[pick]))
    assert np.max(np.abs(fast - exact) / np.abs(exact)) <= 1e-10
    full = np.asarray(co.gls_power_f64(t, y, dy, freq))
    assert full.shape == power.shape and np.all(full > 0)
    rel = np.abs(power - full) / np.abs(full)
    assert rel.max() <= RTOL, (rel.max(), int(np.argmax(rel)))
    assert int(np.argmax(full)) == peak                   # peak-period index bit-exact over the whole grid
    del full, rel
    # normalised power is invariant under y -> a*y + b
    again = _cabi.gls_scan(t, 3.0 * y - 7.0, 3.0 * dy, f0, delta, nf)
    np.testing.assert_allclose(again, power, rtol=1e-7, atol=1e-14)
    # a slab cut out of the middle reproduces the same bins
    part = _cabi.gls_scan(t, y, dy, f0, delta, 4096, j_begin=500_000)
    # (a short slab picks a different tile shape, so agreement is to rounding, not bitwise)
    np.testing.assert_allclose(part, power[500_000:504_096], rtol=1e-8, atol=1e-15)


def test_full_size_c3_batch_peaks_only():
    """BASELINE configs[2]: 4096 curves x 2k samples on a shared 5e4 grid (4.1e11 pairs), reduced on
    the device to (amax, argmax, highest peak); spot-checked against single-curve calls."""
    co.tune_threads()        # the host is shared: the thread count the C checker runs fastest with, measured once
    B, n, nf = 4096, 2000, 50_000
    rng = np.random.default_rng(20241011)
    t = np.sort(rng.uniform(0, float(n), (B, n)), axis=1)
    dy = rng.uniform(0.05, 0.2, (B, n))
    per = 5.0 + 0.01 * np.arange(B)[:, None]
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / per) + dy * rng.standard_normal((B, n))
    offsets = np.arange(B + 1, dtype=np.int64) * n
    df = 1.0 / n / 5
    freq = np.arange(0.5 * df, 0.5 * df + (nf - 1.5) * df + df, df)
    assert freq.size == nf
    f0, delta, _ = _cabi.grid_params(freq)
    _, amax, argmax = _cabi.gls_scan_batch(t.ravel(), y.ravel(), dy.ravel(), offsets, f0, delta, nf,
                                           want_power=False, want_peaks=True)
    idx, val = _cabi.gls_batch_highest_peak(t.ravel(), y.ravel(), dy.ravel(), offsets, f0, delta, nf)
    assert np.all(argmax >= 0) and np.all(np.isfinite(amax))
    found = 1 / freq[argmax]
    # the injected periods are recovered to the grid resolution dP = P^2 df
    assert np.mean(np.abs(found - per[:, 0]) < 1.5 * per[:, 0] ** 2 * df) > 0.99
    # the highest find_peaks() maximum is the global maximum unless that sits on an edge bin (edges are never peaks)
    differ = idx != argmax
    assert np.all((argmax[differ] == 0) | (argmax[differ] == nf - 1))
    for b in rng.integers(0, B, 5):
        single = _cabi.gls_scan(t[b], y[b], dy[b], f0, delta, nf)
        # (a lone curve picks another tile shape than the 4096-curve batch: equal to rounding)
        assert argmax[b] == np.argmax(single) and abs(amax[b] / single.max() - 1) < 1e-12
        assert abs(val[b] / single[idx[b]] - 1) < 1e-12
    # the full-size batch against the ORACLE: the spectra of 8 random curves (first and last included) at all
    # 5e4 bins vs the long-double direct sums (1e8 pairs per curve), Tier E, and their argmax / amax
    power, amax2, argmax2 = _cabi.gls_scan_batch(t.ravel(), y.ravel(), dy.ravel(), offsets, f0, delta, nf,
                                                 want_power=True, want_peaks=True)
    assert np.array_equal(argmax2, argmax) and np.array_equal(amax2, amax)    # peaks-only == with the spectra written
    for b in np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, 6)])):
        exact = np.asarray(co.gls_power_exact(t[b], y[b], dy[b], freq))
        assert_tier_e(power[b], exact)
        assert argmax[b] == int(np.argmax(exact)) and abs(amax[b] / exact.max() - 1) < 1e-9
        assert argmax[b] == int(np.nanargmax(so.gls_power(t[b], y[b], dy[b], freq, delta, f0, sums="fft")))   # Tier R
    # round 6 - EVERY curve at EVERY bin: 4096 spectra x 5e4 bins against the double-precision direct sums (4.1e11
    # pairs on the host; the checker re-proved against the long-double oracle on the 8 curves above), and every
    # curve's argmax / amax as the peaks-only launch reported them
    worst = 0.0
    for b0 in range(0, B, 128):
        rows = slice(b0, min(B, b0 + 128))
        fast = co.gls_power_f64_batch(t[rows], y[rows], dy[rows], freq)
        for b in range(rows.start, rows.stop):
            row = fast[b - b0]
            if b in (0, B - 1):
                exact = np.asarray(co.gls_power_exact(t[b], y[b], dy[b], freq))
                ok = np.abs(exact) > 1e-13 * np.abs(exact).max()
                assert np.max(np.abs(row[ok] - exact[ok]) / np.abs(exact[ok])) <= 1e-10
            worst = max(worst, assert_tier_e(power[b], row))
            k = int(np.argmax(row))
            assert argmax[b] == k, (b, argmax[b], k)
            assert abs(amax[b] / row[k] - 1) < 1e-9
    assert worst <= RTOL
    del power
    # the same batch reduced to its 4 highest / most prominent find_peaks() maxima with prominences and
    # half-maximum crossings (core.py:283-317, 944-978): 1.64 GB of spectra stay in HBM
    from scipy.signal import find_peaks
    top = _cabi.gls_batch_peaks(t.ravel(), y.ravel(), dy.ravel(), offsets, f0, delta, nf, k=4)
    pro = _cabi.gls_batch_peaks(t.ravel(), y.ravel(), dy.ravel(), offsets, f0, delta, nf, k=4,
                                by_prominence=True)
    assert np.array_equal(top["indices"][:, 0], idx)
    assert np.all(top["count"] > 100) and np.all(pro["prominences"][:, 0] >= pro["prominences"][:, 1])
    # against the batch's OWN spectra (same kernel, same tile shape: no rounding slack) every figure is exact:
    # scipy on rows of the power array the batched scan wrote
    power, _, _ = _cabi.gls_scan_batch(t.ravel(), y.ravel(), dy.ravel(), offsets, f0, delta, nf,
                                       want_power=True, want_peaks=False)
    for b in np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, 200)])):
        pk, res = find_peaks(power[b], prominence=0.0)
        assert top["count"][b] == pk.size == pro["count"][b]
        order = np.lexsort((pk, -power[b][pk]))[:4]
        assert np.array_equal(top["indices"][b], pk[order]) and np.array_equal(top["heights"][b], power[b][pk[order]])
        order_p = np.lexsort((pk, -res["prominences"]))[:4]
        assert np.array_equal(pro["indices"][b], pk[order_p])
        assert np.array_equal(pro["prominences"][b], res["prominences"][order_p])
        lo, hi = top["half_lo"][b, 0], top["half_hi"][b, 0]
        assert 0 <= hi < idx[b] <= lo < nf and lo - hi < 40       # the half-maximum width of the main peak
    del power


def test_c1_exactly_as_baseline_defines_it():
    """BASELINE configs[0] (C1): 1k samples x 1k frequencies on SURVEY 8d's grid and seed.  Upstream it is the
    CPU-plumbing config (its CPU leg is in bench.py's cpu_baseline); here the same inputs go through the HIP
    path: Tier E against the long-double sums at every bin, Tier R (peak bin) against the FFT-path
    restatement of the reference, and the FFT path on the device against that restatement."""
    n = nf = 1000
    rng = np.random.default_rng(20241008 + 1)
    t = np.sort(rng.uniform(0, float(n), n))
    dy = rng.uniform(0.05, 0.2, n)
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / 37.3) + dy * rng.standard_normal(n)
    df = 1.0 / (t[-1] - t[0]) / 5
    fmin = 0.5 * df
    freq = np.arange(fmin, fmin + (nf - 1.5) * df + df, df)
    assert freq.size == nf
    ls = GLS(fmin=fmin, fmax=fmin + (nf - 1.5) * df)(TSeries(t, y), err=dy)
    assert np.array_equal(ls.frequency, freq)
    exact = np.asarray(co.gls_power_exact(t, y, dy, freq))
    assert_tier_e(ls.values, exact)
    ref = so.gls_power(t, y, dy, freq, df, fmin, True, False, sums="fft")
    assert ls.argmax() == int(np.argmax(exact)) == int(np.nanargmax(ref))
    assert ls.period_at_highest_peak == 1 / freq[int(np.nanargmax(ref))]
    fft = GLS(fmin=fmin, fmax=fmin + (nf - 1.5) * df, method="fft")(TSeries(t, y), err=dy)
    assert np.max(np.abs(fft.values - ref)) <= 1e-9 * np.abs(ref).max()


def test_full_size_c4_on_one_gpu_both_paths():
    """BASELINE configs[3] (N=1e6 x nf=1e7 = 1e13 pairs) on ONE GPU: the grid in 8 slabs exactly as
    the 8-GPU run shards it, cross-checked against the FFT path (which reproduces the reference's own
    algorithm) and, on a few bins, against the long-double oracle."""
    co.tune_threads()        # the host is shared: the thread count the C checker runs fastest with, measured once
    n, nf, world = 1_000_000, 10_000_000, 8
    t, y, dy = synth(n, 20241013)
    df = 1.0 / (t[-1] - t[0]) / 5
    fmin = 0.5 * df
    f0, delta = fmin, df
    slab = nf // world
    power = np.concatenate([_cabi.gls_scan(t, y, dy, f0, delta, slab, j_begin=r * slab)
                            for r in range(world)])
    assert power.shape == (nf,) and np.all(np.isfinite(power))
    # the same sharding through the persistent plan the 8-GPU run uses (8 logical slots on this one device,
    # the all-gather as device copies - what `bench.py --loopback 8` reports as c4_sharded): every slot
    # must end up holding exactly the slab-wise array
    plan = _cabi.GlsPlan([0], n, nf, loopback_slots=world)
    plan.upload(t, y, dy)
    plan.scan(f0, delta, nf)
    for which in (0, 3, world - 1):
        assert np.array_equal(plan.download(which), power), f"loopback-8 plan, slot {which}"
    assert len(plan.slot_ms()) == world and min(plan.slot_ms()) > 0
    plan.close()
    fft = _cabi.gls_scan_fft(t, y, dy, fmin, df, nf)
    peak = int(np.argmax(power))
    assert peak == int(np.argmax(fft))                                   # tier R at the largest config
    assert abs(1 / (f0 + peak * delta) - 37.3) < 0.01
    assert np.median(np.abs(fft - power)) < 1e-6 and np.max(np.abs(fft - power)) < 1e-3
    # >= 64 bins against the long-double oracle (1e6 pairs each): both edges of every slab (where the 8-GPU
    # run starts / ends a device's recurrences), the peak, and random bins in every slab
    rng = np.random.default_rng(5)
    edges = np.concatenate([[r * slab, (r + 1) * slab - 1] for r in range(world)])
    inner = (np.arange(world)[:, None] * slab + rng.integers(1, slab - 1, (world, 6))).ravel()
    pick = np.unique(np.concatenate([edges, inner, [peak]]))
    assert pick.size >= 64
    exact = co.gls_power_exact(t, y, dy, f0 + delta * pick)
    assert np.max(np.abs(power[pick] - exact) / np.abs(exact)) <= RTOL
    assert peak == pick[np.argmax(exact)]
    # round 6 - one FULL slab (the last: the highest frequencies, the longest phases) and the 8192 bins around the
    # peak against the double-precision direct sums (1.26e12 pairs on the host), the checker first re-proved on
    # the sampled bins against the long-double oracle
    fast = np.asarray(co.gls_power_f64(t, y, dy, f0 + delta * pick))
    assert np.max(np.abs(fast - exact) / np.abs(exact)) <= 1e-10
    for a, b in ((nf - slab, nf), (max(0, peak - 4096), peak + 4096)):
        full = np.asarray(co.gls_power_f64(t, y, dy, f0 + delta * np.arange(a, b)))
        assert_tier_e(power[a:b], full)
        assert int(np.argmax(full)) == int(np.argmax(power[a:b]))
    assert a + int(np.argmax(full)) == peak


_SHAPE_CHECK = """
import numpy as np
from oracle import c_oracle as co
from periodicity_amd import _cabi
rng = np.random.default_rng(5)
for n, nf, fit_mean, with_dy in [(777, 3001, True, True), (130, 517, False, True), (64, 2049, True, False)]:
    t = np.sort(rng.uniform(0, 80.0, n)) + 2454900.5
    y = np.sin(2 * np.pi * t / 7.7) + 0.3 * rng.standard_normal(n)
    dy = rng.uniform(0.1, 0.4, n) if with_dy else None
    f0, delta = 0.004, 0.00037
    freq = f0 + delta * np.arange(nf)
    got = _cabi.gls_scan(t, y, dy, f0, delta, nf, fit_mean)
    want = np.asarray(co.gls_power_exact(t, y, dy, freq, fit_mean))
    ok = np.abs(want) > 1e-13 * np.nanmax(np.abs(want))
    rel = np.max(np.abs(got[ok] - want[ok]) / np.abs(want[ok]))
    assert rel < 1e-6, (n, nf, rel)
    S, C = _cabi.trig_sums(t, y, f0, delta, nf)
    Se, Ce = co.trig_sums_exact(t, y, freq)
    assert np.max(np.abs(S - Se)) < 1e-9 * n and np.max(np.abs(C - Ce)) < 1e-9 * n
print("ok")
"""


@pytest.mark.parametrize("K,S", [(4, 1), (8, 1), (16, 1), (4, 2), (8, 4), (16, 4)])
def test_every_tile_shape_is_exact(K, S):
    """The launcher picks (K frequencies per thread, S waves per tile) from a cost model; every
    instantiation must give Tier-E results, including the ones the model rarely picks.  The
    override is read once per process, hence the child interpreter."""
    import subprocess
    import sys
    env = dict(os.environ, PDC_GLS_K=str(K), PDC_GLS_S=str(S))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _SHAPE_CHECK], env=env, cwd=root, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
