"""GPU parity tests of the phase-folding scans (PDM, StringLength) through the C ABI.

fp64 tolerance: the discrete decisions (bin membership, sort order) are reproduced exactly, so
what is left is summation order: 1e-9 relative is the gate (observed ~1e-13)."""
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import scan_oracle as so
from periodicity_amd import _cabi
from periodicity_amd.core import TSeries
from periodicity_amd.phase import PDM, StringLength

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def synth(n, seed, period=13.7):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, float(n), n))
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / period) + 0.1 * rng.standard_normal(n)
    return t, y


# ---- PDM -------------------------------------------------------------------------------------
@pytest.mark.parametrize("nb,nc", [(5, 2), (10, 3)])
def test_pdm_seam_and_call_golden(golden_dir, nb, nc):
    g = load(golden_dir, "g7_pdm")
    periods = g[f"periods_{nb}_{nc}"]
    theta = _cabi.pdm_scan(g["t"], g["y"], periods, nb, nc, float(g["sigma"]))
    np.testing.assert_allclose(theta, g[f"theta_seam_{nb}_{nc}"], rtol=RTOL)
    assert np.argmin(theta) == np.argmin(g[f"theta_seam_{nb}_{nc}"])      # the best period: same index
    pdm = PDM(nb=nb, nc=nc, p_min=1.0, p_max=60.0, n_periods=200, cores=1)
    res = pdm(TSeries(g["t"], g["y"]))
    assert np.array_equal(res.frequency, g[f"frequency_{nb}_{nc}"])
    np.testing.assert_allclose(res.values, g[f"theta_call_{nb}_{nc}"], rtol=RTOL)
    assert np.argmin(res.values) == np.argmin(g[f"theta_call_{nb}_{nc}"])
    assert np.array_equal(pdm.periods, periods) and pdm.sigma == float(g["sigma"])
    assert pdm.periodogram is res and pdm.signal.size == g["t"].size
    np.testing.assert_allclose(pdm._pdm(periods[17]), g[f"theta_seam_{nb}_{nc}"][17], rtol=RTOL)


def test_pdm_variants_golden(golden_dir):
    g = load(golden_dir, "g7_pdm")
    sig = TSeries(g["t"], g["y"])
    res = PDM(p_min=1.0, p_max=60.0, n_periods=200, do_subharmonic=True)(sig)
    assert np.array_equal(res.frequency, g["frequency_sub"])
    np.testing.assert_allclose(res.values, g["theta_call_sub"], rtol=RTOL)
    res = PDM(n_periods=150)(sig)                       # default p_min / p_max
    assert np.array_equal(res.frequency, g["frequency_default"])
    np.testing.assert_allclose(res.values, g["theta_call_default"], rtol=RTOL)
    res = PDM(p_min=1.0, p_max=60.0, n_periods=200)(TSeries(g["t_negative"], g["y"]))
    np.testing.assert_allclose(res.values, g["theta_call_negative"], rtol=RTOL)
    # theta has its minimum at the injected period
    res = PDM(p_min=1.0, p_max=60.0, n_periods=200)(sig)
    ratio = res.period[np.argmin(res.values)] / 13.7     # the injected period or a multiple of it
    assert abs(ratio - round(ratio)) < 0.03


def test_pdm_shapes_and_edges():
    t, y = synth(3000, 4)
    sigma = np.var(y, ddof=1)
    for nb, nc, n_per in ((5, 2, 1), (7, 7, 300), (10, 5, 257), (19, 10, 70), (1, 1, 5), (3, 1, 64)):
        periods = np.linspace(0.9, 77.0, n_per)
        got = _cabi.pdm_scan(t, y, periods, nb, nc, sigma)
        want = so.pdm_scan(t, y, periods, nb, nc)
        np.testing.assert_allclose(got, want, rtol=RTOL, err_msg=f"{nb}x{nc}")
    # samples on and next to bin edges, phases that round to exactly 1.0, negative times
    te = np.array([-1e-20, -0.75, 0.0, 0.2, 0.4, 0.6000000000000001, 0.8, 1.0, 1.2, 2.0 - 1e-16,
                   3.0, 5.5, -7.25, 10.0, 0.1, 0.30000000000000004])
    ye = np.cos(np.arange(te.size)) + 0.01 * np.arange(te.size)
    pe = np.array([1.0, 2.0, 0.5, 0.2, 4.0, 1e-3, 3.3333333333333335, 1e6])
    got = _cabi.pdm_scan(te, ye, pe, 5, 2, np.var(ye, ddof=1))
    with np.errstate(all="ignore"):
        want = so.pdm_scan(te, ye, pe, 5, 2)
    np.testing.assert_allclose(got, want, rtol=RTOL, equal_nan=True)
    assert _cabi.pdm_scan(t, y, np.empty(0), 5, 2, sigma).size == 0
    with pytest.raises(ValueError):
        _cabi.pdm_scan(t, y[:-1], [1.0], 5, 2, sigma)
    with pytest.raises(ValueError):
        _cabi.pdm_scan(t, y, [1.0], 50, 50, sigma)
    raw = PDM(n_periods=40)(y)                            # phase.py:160-161: raw array-likes
    assert raw.size == 40


def test_pdm_mid_size_vs_c_oracle():
    t, y = synth(20000, 9)
    periods = np.linspace(1.0, 100.0, 1500)
    got = _cabi.pdm_scan(t, y, periods, 5, 2, np.var(y, ddof=1))
    want = co.pdm_scan(t, y, periods, 5, 2)
    np.testing.assert_allclose(got, want, rtol=RTOL)


# ---- StringLength ------------------------------------------------------------------------------
def test_stringlength_seam_golden(golden_dir):
    g = load(golden_dir, "g8_stringlength")
    ell = _cabi.stringlength_scan(g["t"], g["m"], g["periods"])
    np.testing.assert_allclose(ell, g["ell"], rtol=RTOL)
    assert np.argmin(ell) == np.argmin(g["ell"]) and np.argmax(ell) == np.argmax(g["ell"])
    # evenly sampled: many bit-identical phases, the stable (time) order of ties matters
    ell = _cabi.stringlength_scan(g["t_even"], g["m_even"], g["periods_even"])
    np.testing.assert_allclose(ell, g["ell_even"], rtol=RTOL)
    assert np.argmin(ell) == np.argmin(g["ell_even"])


def test_stringlength_call_matches_restated_reference(golden_dir):
    g = load(golden_dir, "g8_stringlength")
    sl = StringLength(n_periods=200, cores=1)
    res = sl(TSeries(g["t"], g["y"]))
    freq, ell = so.stringlength(g["t"], g["y"], n_periods=200)
    assert np.array_equal(res.frequency, freq)
    np.testing.assert_allclose(res.values, ell, rtol=RTOL)
    assert np.array_equal(sl.m.values, g["m"]) and sl.periodogram is res
    np.testing.assert_allclose(res.values[::-1], g["ell"], rtol=RTOL)   # ascending frequency
    np.testing.assert_allclose(sl._stringlength(g["periods"][5]), g["ell"][5], rtol=RTOL)
    assert abs(res.period[np.argmin(res.values)] / 13.7 - round(res.period[np.argmin(res.values)] / 13.7)) < 0.05


def test_stringlength_clusters_take_the_scratch_path():
    # 20000 evenly spaced samples folded at commensurate periods: every phase falls into a
    # handful of values (one bucket holds > CAP samples) -> sorted in global scratch
    t = np.arange(20000.0)
    y = np.sin(2 * np.pi * t / 12.5) + 0.05 * np.cos(0.37 * t)
    m = so.stringlength_scale(y)
    periods = np.array([1.0, 2.0, 2.5, 4.0, 12.5, 3.0000000000000004, 7.3, 20000.0, 1e-3])
    got = _cabi.stringlength_scan(t, m, periods)
    want = co.stringlength_scan(t, m, periods)
    np.testing.assert_allclose(got, want, rtol=RTOL)
    np.testing.assert_allclose(got[:3], so.stringlength_scan(t, m, periods[:3]), rtol=RTOL)


def test_stringlength_two_workgroups_per_cu_equal_the_one_workgroup_kernel(tmp_path):
    """N <= 26 048 runs two 512-thread workgroups per CU (sl_duo_kernel); PDC_SL_DUO=0 keeps the 1024-thread
    kernel.  Same ranges, same arithmetic per range; only the order in which the range sums are added
    differs.  Shapes around the switch (26 048 / 26 049), clusters (deferred ranges, the marked-period
    fallback), tied and NaN time stamps; each kernel bitwise reproducible."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("duo", {"PDC_SL_DUO": "1"}), ("one", {"PDC_SL_DUO": "0"})):
        path = str(tmp_path / f"{tag}.npz")
        run = subprocess.run([sys.executable, os.path.join(root, "tools", "sl_ab_check.py"), path],
                             env=dict(os.environ, **env), cwd=root, capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-2000:]
        res[tag] = np.load(path)
    assert sorted(res["duo"].files) == sorted(res["one"].files)
    for key in res["duo"].files:
        a, b = res["duo"][key], res["one"][key]
        assert np.array_equal(np.isnan(a), np.isnan(b)), key
        np.testing.assert_allclose(a, b, rtol=1e-12, err_msg=key)
        if key.endswith("_again"):
            assert np.array_equal(a, res["duo"][key[:-6]], equal_nan=True), key   # run-to-run bitwise
    # and against the oracle where it is cheap
    t = np.arange(20_000.0)
    m = so.stringlength_scale(np.sin(2 * np.pi * t / 12.5) + 0.05 * np.cos(0.37 * t))
    pe = np.array([1.0, 2.0, 2.5, 4.0, 12.5, 3.0000000000000004, 7.3, 20000.0, 1e-3, 0.3, 1 / 3, 100.0, 128.0])
    np.testing.assert_allclose(res["duo"]["clustered"], co.stringlength_scan(t, m, pe), rtol=RTOL)


def test_stringlength_large_n_runs_in_phase_slices():
    """Default dispatch (the several-slice kernels up to 262 143 samples, the streamed kernels beyond), and once more in
    a child process with PDC_SL_STREAM_MIN=100000, which sends the 200 000-sample curve through the streamed kernels
    too."""
    import subprocess
    import sys
    _large_n_body()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "import tests.test_phase_gpu as T; T._large_n_body(); print('ok')"],
                         env=dict(os.environ, PDC_SL_STREAM_MIN="100000"), cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-1500:] + out.stderr[-1500:]


def _large_n_body():
    # more samples than one LDS slice holds (52112 with 16-bit indices): up to 16 slices of ~24k the fast
    # kernel reads every sample's bucket id back per slice and keeps that slice's; beyond, the general kernel groups
    # the samples by coarse bucket once per period in global scratch and sorts slice after slice
    t6, y6 = synth(60_000, 78)                              # two slices of 16-bit indices (N < 65536)
    m6 = so.stringlength_scale(y6)
    p6 = np.array([0.9, 13.7, 4000.0])
    np.testing.assert_allclose(_cabi.stringlength_scan(t6, m6, p6),
                               co.stringlength_scan(t6, m6, p6), rtol=RTOL)
    t, y = synth(70_000, 77)
    m = so.stringlength_scale(y)
    periods = np.array([0.9, 13.7, 333.3, 9000.0])
    np.testing.assert_allclose(_cabi.stringlength_scan(t, m, periods),
                               co.stringlength_scan(t, m, periods), rtol=RTOL)
    te = np.arange(70_000.0)                                # clusters: one bucket > 4096 samples
    me = so.stringlength_scale(np.sin(2 * np.pi * te / 12.5) + 0.05 * np.cos(0.37 * te))
    pe = np.array([1.0, 2.5, 7.3])
    np.testing.assert_allclose(_cabi.stringlength_scan(te, me, pe),
                               co.stringlength_scan(te, me, pe), rtol=RTOL)
    # 2e5: nine slices of 32-bit indices (the several-slice kernel's upper range); 380k / 400k: the streamed kernels
    for n_big, seed in ((200_000, 81), (380_000, 79), (400_000, 80)):
        tb, yb = synth(n_big, seed)
        mb = so.stringlength_scale(yb)
        pb = np.array([0.7, 13.7, 2.0, 51_234.5])
        got = _cabi.stringlength_scan(tb, mb, pb)
        np.testing.assert_allclose(got, co.stringlength_scan(tb, mb, pb), rtol=RTOL)
        assert np.array_equal(got, _cabi.stringlength_scan(tb, mb, pb))


def test_stringlength_edges():
    t, y = synth(700, 2)
    m = so.stringlength_scale(y)
    assert _cabi.stringlength_scan(t, m, np.empty(0)).size == 0
    one = _cabi.stringlength_scan(t[:1], m[:1], [3.0])
    assert one[0] == 0.0
    two = _cabi.stringlength_scan(t[:2], m[:2], [3.0])
    np.testing.assert_allclose(two, so.stringlength_scan(t[:2], m[:2], [3.0]), rtol=RTOL)
    tn = t - 350.0                                        # negative times
    np.testing.assert_allclose(_cabi.stringlength_scan(tn, m, [0.7, 13.7, 401.0]),
                               so.stringlength_scan(tn, m, [0.7, 13.7, 401.0]), rtol=RTOL)
    with pytest.raises(ValueError):
        _cabi.stringlength_scan(t, m[:-1], [1.0])
    for n in (4095, 4096, 4097, 9000):                    # around one LDS range
        tt, yy = synth(n, n)
        mm = so.stringlength_scale(yy)
        pp = np.array([0.77, 13.7, 55.5])
        np.testing.assert_allclose(_cabi.stringlength_scan(tt, mm, pp),
                                   co.stringlength_scan(tt, mm, pp), rtol=RTOL)


def test_streamed_stringlength_kernels_match_the_oracle():
    """The streamed path (no gather: partition by phase bin -> LDS sort per bin -> links) serves curves whose
    (t, m) table outgrows L2; PDC_SL_STREAM_MIN routes small curves through it so that it meets the oracle at
    sizes the oracle finishes in seconds: uneven and even sampling (bins that overflow at commensurate periods
    fall back to the general kernel), a last tile that is not full, one bin, bitwise repeatability."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "sl_stream_check.py"),
                          "70000x48", "4097x33", "20000x40e", "150001x12", "9000x20e"],
                         env=dict(os.environ, PDC_SL_STREAM_MIN="4096"), cwd=root, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok"), out.stdout[-1500:] + out.stderr[-1500:]


def test_streamed_stringlength_slices_mode_equals_the_lists_bit_for_bit(tmp_path):
    """With t non-decreasing the samples of (cycle of the period, phase bin) are one slice of t[] / m[]: the sort
    kernel fetches a bin's records from those slices (a table of first samples, sl_bound_kernel) instead of from the
    partition kernel's lists.  Both feed the same sort, so PDC_SL_SLICES=0 must give the same bits - over grids that
    mix the two modes (short periods keep the lists), duplicate time stamps, gaps of many periods (cycles with no
    sample), a negative start, even sampling; samples in random order make the slices mode step aside."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    specs = ["70000x48", "150001x12", "90000x40d", "4097x33d", "20000x40e", "60000x24u"]
    got = {}
    for mode in ("1", "0"):
        path = str(tmp_path / f"slices{mode}.npz")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "sl_stream_check.py"), *specs],
                             env=dict(os.environ, PDC_SL_STREAM_MIN="4096", PDC_SL_SLICES=mode, SL_CHECK_SAVE=path),
                             cwd=root, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok"), out.stdout[-1500:] + out.stderr[-1500:]
        got[mode] = np.load(path)
    for spec in specs:
        same_sort = ~got["1"][spec + ":one_cycle"]      # (a period that outlasts the samples is summed without a sort)
        assert np.array_equal(got["1"][spec][same_sort], got["0"][spec][same_sort]), spec
        np.testing.assert_allclose(got["1"][spec], got["0"][spec], rtol=1e-13)


def test_streamed_stringlength_slices_mode_at_its_own_sizes(tmp_path):
    """The same equality at the sizes the streamed path serves by default, where a workgroup of the sort kernel walks
    more than 64 bins (its table of items is refilled) and the boundary kernel runs 1536 workgroups - a barrier missing
    there showed up as a different period wrong in every other run at exactly these shapes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    specs = ["300000x4096", "1000000x384"]
    runs = []
    for mode in ("1", "0", "1"):
        path = str(tmp_path / f"ab{len(runs)}.npz")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "sl_slices_ab.py"), path, *specs],
                             env=dict(os.environ, PDC_SL_SLICES=mode), cwd=root, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        runs.append(np.load(path))
    for spec in specs:
        same_sort = ~runs[0][spec + ":one_cycle"]       # (a period that outlasts the samples is summed without a sort)
        assert np.array_equal(runs[0][spec][same_sort], runs[1][spec][same_sort]), spec
        np.testing.assert_allclose(runs[0][spec], runs[1][spec], rtol=1e-13)
        assert np.array_equal(runs[0][spec], runs[2][spec]), spec


def _sl_oracle_full(specs, **env):
    """tools/sl_oracle_full.py in a child process (the library reads its PDC_SL_* switches once per process): EVERY
    period of the reference's grid against the C oracle (OpenMP over periods: seconds on the GPU box's host)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "sl_oracle_full.py"), *specs],
                         env=dict(os.environ, **env), cwd=root, capture_output=True, text=True, timeout=1700)
    print(out.stdout)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok"), out.stdout[-2500:] + out.stderr[-2500:]


@pytest.mark.parametrize("slices", ["1", "0"])
def test_streamed_stringlength_every_period_against_the_oracle_at_its_own_sizes(slices):
    """The round-4 streamed family where it actually runs - 300 000 x 4096 and 1 000 000 x 384 (a full batch of 384
    periods), ALL periods against the oracle, not a sample of three: slices mode (default) and lists mode
    (PDC_SL_SLICES=0).  phase.py:45-51 + core.py:473-477,543-544."""
    _sl_oracle_full(["300000x4096", "1000000x384"], PDC_SL_SLICES=slices)


@pytest.mark.parametrize("groups", ["1", "2", "4"])
def test_streamed_stringlength_full_batches_per_group_count(groups):
    """The shape of the race fixed in 597cf50 (one period in thousands wrong at batches of 320+ periods): a batch of
    384 periods for every group count of the histogram / boundary / partition kernels, every period against the
    oracle; duplicates + gaps + a negative start, a Julian-date offset, and a grid of short periods (lists mode for
    time-ordered samples), all at N >= 262 144."""
    _sl_oracle_full(["262144x384", "300000x384d", "280000x352o", "270000x384s", "262144x320ds"], PDC_SL_STREAM_GROUPS=groups)


@pytest.mark.parametrize("dev", ["", "1"])
def test_streamed_stringlength_samples_in_any_order(dev):
    """Samples handed over in a random order at N >= 262 144 (the C ABI allows it; a TSeries never is): ordered by time on
    the device first (csrc/timesort.inc: stable radix sort - duplicates of a time stamp keep the caller's order), then
    the kernels a TSeries gets; every period against the oracle's result for the time-ordered series.  Duplicates + gaps
    + a negative start, a Julian-date offset, times that straddle zero with -0.0 / +0.0 stamps, N = 1e6.  Through the
    host entry (which sees that the samples are out of order) and through the _dev entry (which cannot: the decision is
    the device's)."""
    _sl_oracle_full(["300000x384u", "262144x320du", "280000x352ou", "1000000x96uz"], SL_FULL_DEV=dev)


def test_streamed_stringlength_dev_entry_with_samples_in_order():
    """The _dev entry enqueues the time sort's launches whatever the order: for samples in order each of them returns at
    once and the copies the kernels read are the samples as they stand."""
    _sl_oracle_full(["300000x384", "262144x320d"], SL_FULL_DEV="1")


def test_streamed_stringlength_samples_in_any_order_without_the_time_sort():
    """PDC_SL_TIMESORT=0: the lists mode (what such samples got up to round 4) is still right."""
    _sl_oracle_full(["300000x96u", "262144x64du"], PDC_SL_TIMESORT="0")


def test_untame_time_stamps_out_of_order_keep_the_lists_mode():
    """Out of order AND a time stamp the exact fold cannot take the fast way (|t| beyond 1e150), or a NaN: no time sort on
    the device (sl_tame_kernel's first flag), the lists mode as before - values as the oracle's on the arrays as given."""
    rng = np.random.default_rng(19)
    n = 262_144
    t = rng.uniform(0.0, 5000.0, n)
    m = so.stringlength_scale(np.sin(2 * np.pi * t / 13.7) + 0.2 * rng.standard_normal(n))
    periods = np.array([3.3, 13.7, 47.0, 900.0])
    t[12345] = 1e200
    np.testing.assert_allclose(_cabi.stringlength_scan(t, m, periods), co.stringlength_scan(t, m, periods), rtol=RTOL)
    t[777] = np.nan
    with np.errstate(all="ignore"):
        want = so.stringlength_scan(t, m, periods)
    np.testing.assert_allclose(_cabi.stringlength_scan(t, m, periods), want, rtol=RTOL, equal_nan=True)


def test_time_sort_at_the_largest_streamed_size():
    """N = 5e6 (the streamed kernels serve up to 5.5 M samples; 2442 tiles in the time sort): shuffled samples give the
    bits of the same samples in order - no time stamp is repeated here, so the sorted arrays are the same arrays."""
    rng = np.random.default_rng(8)
    n = 5_000_000
    t = np.sort(rng.uniform(-1e6, 4e6, n))
    assert np.all(np.diff(t) > 0)
    m = so.stringlength_scale(np.sin(2 * np.pi * t / 1234.5) + 0.2 * rng.standard_normal(n))
    df = 0.1 / (t[-1] - t[0])
    periods = 1 / np.linspace(24 * df, df, 24)
    want = _cabi.stringlength_scan(t, m, periods)
    order = rng.permutation(n)
    assert np.array_equal(_cabi.stringlength_scan(t[order], m[order], periods), want)
    np.testing.assert_allclose(want[[0, 23]], co.stringlength_scan(t, m, periods[[0, 23]]), rtol=RTOL)


def test_several_slice_stringlength_every_period_against_the_oracle():
    """The several-slice instances of sl_fast_kernel (52 112 < N < 262 144; 16-bit indices + bit planes): all periods."""
    _sl_oracle_full(["74326x2048", "131000x1024d", "200000x1024o", "261000x512"])


def test_stringlength_periods_that_outlast_the_samples():
    """p > baseline: the samples span less than one cycle, their phase order is their time order (or, across a cycle
    boundary, the later samples first) and no kernel sorts anything (onecycle::* / sl_direct_kernel).  Julian-date
    offsets put the cycle boundary inside the samples for many of these periods; duplicates keep their index order;
    p just below the baseline takes the ordinary kernels.  All families: one slice, several slices, streamed."""
    rng = np.random.default_rng(31)
    for n in (3000, 30_000, 90_000, 300_000):
        t = np.sort(rng.uniform(0.0, 400.0, n)) + 2454953.5
        t[7:n:11] = t[6:n - 1:11]                            # duplicate time stamps
        y = np.sin(2 * np.pi * t / 13.7) + 0.2 * rng.standard_normal(n)
        m = so.stringlength_scale(y)
        base = t[-1] - t[0]
        periods = np.concatenate([base * np.array([0.97, 0.999, 1.0, 1.001, 1.5, 2.0, 3.3, 9.99, 10.0, 57.0]),
                                  base * rng.uniform(1.0, 12.0, 14), [2454953.5, 2454953.5 / 2, 1e7, 1e9]])
        got = _cabi.stringlength_scan(t, m, periods)
        np.testing.assert_allclose(got, co.stringlength_scan(t, m, periods), rtol=RTOL)
        assert np.array_equal(got, _cabi.stringlength_scan(t, m, periods))
        # the same curve in another order: nothing may be assumed about it.  (Ties: periods[2] is the baseline itself - the
        # first and the last sample share a phase without sharing a time stamp.  Below 262 144 samples such a tie keeps
        # the order given; from there on the samples are ordered by time on the device first and the tie is taken in time
        # order, as the reference's TSeries(t, m) would - core.py:473-477 sorts by time, stably.)
        order = rng.permutation(n)
        to, mo = t[order], m[order]
        if n >= 262_144:
            back = np.argsort(to, kind="stable")
            want = co.stringlength_scan(to[back], mo[back], periods[:6])
        else:
            want = co.stringlength_scan(to, mo, periods[:6])
        np.testing.assert_allclose(_cabi.stringlength_scan(to, mo, periods[:6]), want, rtol=RTOL)


def test_streamed_stringlength_at_its_own_sizes():
    """N = 4e5 (just above the several-slice kernel's range) and N = 1e6 take the streamed kernels by default."""
    rng = np.random.default_rng(12)
    for n, n_per in ((400_000, 40), (1_000_000, 24)):
        t = np.sort(rng.uniform(0, float(n), n))
        y = np.sin(2 * np.pi * t / 13.7) + 0.2 * rng.standard_normal(n)
        m = so.stringlength_scale(y)
        df = 0.1 / (t[-1] - t[0])
        periods = 1 / np.linspace(n_per * df * 100, df, n_per)
        got = _cabi.stringlength_scan(t, m, periods)
        pick = np.array([0, n_per // 2, n_per - 1])
        np.testing.assert_allclose(got[pick], co.stringlength_scan(t, m, periods[pick]), rtol=RTOL)
        assert np.array_equal(got, _cabi.stringlength_scan(t, m, periods))


def test_phase_scans_full_size_c5():
    """BASELINE configs[4]: N=5e4 samples x 1e5 trial periods, both scans, EVERY period against the C oracle
    (round 6: the 600-period sample is gone - 5e9 pairs per scan through `oracle/scan_oracle.c`, OpenMP over the
    periods), the argmin index identical, plus invariances of the statistics."""
    co.tune_threads()        # the host is shared: the thread count the C checker runs fastest with, measured once
    n, n_per = 50_000, 100_000
    t, y = synth(n, 20241012)
    periods = np.linspace(1.0, 100.0, n_per)
    sigma = np.var(y, ddof=1)
    theta = _cabi.pdm_scan(t, y, periods, 5, 2, sigma)
    assert np.all(np.isfinite(theta)) and theta.min() > 0 and theta.max() < 1.5
    assert abs(periods[np.argmin(theta)] / 13.7 - round(periods[np.argmin(theta)] / 13.7)) < 0.01
    want = co.pdm_scan(t, y, periods, 5, 2)
    assert want.shape == theta.shape
    np.testing.assert_allclose(theta, want, rtol=RTOL)
    assert int(np.argmin(theta)) == int(np.argmin(want))             # "peak-period index bit-exact", whole grid
    # theta is invariant under x -> a*x + b (sigma scales with it)
    again = _cabi.pdm_scan(t, 2.5 * y - 3.0, periods[:4096], 5, 2, 2.5 ** 2 * sigma)
    np.testing.assert_allclose(again, theta[:4096], rtol=1e-8)

    m = so.stringlength_scale(y)
    df = 0.1 / (t[-1] - t[0])
    sl_periods = 1 / np.linspace(n_per * df, df, n_per)
    ell = _cabi.stringlength_scan(t, m, sl_periods)
    assert np.all(np.isfinite(ell)) and ell.min() > 0
    want = co.stringlength_scan(t, m, sl_periods)
    np.testing.assert_allclose(ell, want, rtol=RTOL)
    assert int(np.argmin(ell)) == int(np.argmin(want))
    # the closed polygon is at least twice the phase span plus twice the value span
    assert ell.min() >= 2 * (m.max() - m.min())
    # the PDM period grid through StringLength too (uniform in period, 1 ... 100 d: few cycles per period at the
    # long end, the one-cycle pre-pass and the fast kernel's other branches): every period again
    ell_p = _cabi.stringlength_scan(t, m, periods)
    want_p = co.stringlength_scan(t, m, periods)
    np.testing.assert_allclose(ell_p, want_p, rtol=RTOL)
    assert int(np.argmin(ell_p)) == int(np.argmin(want_p))


def test_period_grid_sharded_over_device_slots():
    """pdc_pdm_scan_multi / pdc_stringlength_scan_multi cut the period grid into one contiguous slab
    per listed device (listing device 0 three times runs three slabs on three streams): results are
    bit-identical to the single launch, including a grid shorter than the device list."""
    rng = np.random.default_rng(8)
    t = np.sort(rng.uniform(0, 60.0, 1500))
    x = np.sin(2 * np.pi * t / 4.4) + 0.2 * rng.standard_normal(t.size)
    m = (x - x.max()) / (2 * (x.max() - x.min())) + 0.25
    for n_periods in (1000, 7, 2):
        periods = np.linspace(0.7, 30.0, n_periods)
        one = _cabi.pdm_scan(t, x, periods, 5, 2, np.var(x, ddof=1))
        many = _cabi.pdm_scan(t, x, periods, 5, 2, np.var(x, ddof=1), devices=(0, 0, 0))
        # (a shorter period grid may split the samples over more workgroups: same bins and counts,
        # sums added in another order)
        np.testing.assert_allclose(many, one, rtol=1e-12)
        one = _cabi.stringlength_scan(t, m, periods)
        many = _cabi.stringlength_scan(t, m, periods, devices=(0, 0, 0))
        assert np.array_equal(one, many)
    pdm = PDM(p_min=0.7, p_max=30.0, n_periods=300, devices=(0, 0))
    np.testing.assert_allclose(pdm(TSeries(t, x)).values,
                               PDM(p_min=0.7, p_max=30.0, n_periods=300)(TSeries(t, x)).values, rtol=1e-12)
    sl = StringLength(n_periods=300, devices=(0, 0))
    assert np.array_equal(sl(TSeries(t, x)).values, StringLength(n_periods=300)(TSeries(t, x)).values)


def test_pdm_few_periods_many_samples_split_the_samples(tmp_path):
    """The reference's default grid has 1000 trial periods: with lanes = periods the chip would idle,
    so the samples are split over workgroups as well (statistics once, partial histograms, one
    finishing launch).  Same thetas as the C oracle and as the unsplit kernel (child process with
    PDC_PDM_SPLIT=0), for odd sizes, NaNs in the data and a grid of one period."""
    import subprocess
    import sys
    rng = np.random.default_rng(31)
    for n, n_periods, nb, nc in [(50_001, 1000, 5, 2), (9_999, 7, 10, 3), (4_100, 1, 5, 2), (20_000, 130, 4, 1)]:
        t = np.sort(rng.uniform(0, 300.0, n)) - 20.0
        x = np.sin(2 * np.pi * t / 6.3) + 0.3 * rng.standard_normal(n)
        periods = np.linspace(0.9, 40.0, n_periods)
        got = _cabi.pdm_scan(t, x, periods, nb, nc, np.var(x, ddof=1))
        want = co.pdm_scan(t, x, periods, nb, nc)
        np.testing.assert_allclose(got, want, rtol=1e-9)
    x[5] = np.nan
    got = _cabi.pdm_scan(t, x, periods, 4, 1, 1.0)
    assert np.all(np.isnan(got))          # a NaN sample poisons every bin sum it lands in and the mean
    code = ("import numpy as np; from periodicity_amd import _cabi; rng = np.random.default_rng(31); n = 50001;"
            "t = np.sort(rng.uniform(0, 300.0, n)) - 20.0; x = np.sin(2 * np.pi * t / 6.3) + 0.3 * rng.standard_normal(n);"
            "p = np.linspace(0.9, 40.0, 1000); a = _cabi.pdm_scan(t, x, p, 5, 2, np.var(x, ddof=1)); np.save(r'%s', a)" % str(tmp_path / 'pdm_unsplit.npy'))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PDC_PDM_SPLIT="0"), cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rng = np.random.default_rng(31)
    n = 50_001
    t = np.sort(rng.uniform(0, 300.0, n)) - 20.0
    x = np.sin(2 * np.pi * t / 6.3) + 0.3 * rng.standard_normal(n)
    split = _cabi.pdm_scan(t, x, np.linspace(0.9, 40.0, 1000), 5, 2, np.var(x, ddof=1))
    np.testing.assert_allclose(split, np.load(str(tmp_path / "pdm_unsplit.npy")), rtol=1e-12)


def test_pdm_split_mode_is_stable_over_repeated_calls_and_reallocations():
    """Regression: the split mode's partial histograms used to live in stream-ordered pool memory
    (hipMallocAsync); a block handed out again by the pool gave the next call stale partials - the
    second of two back-to-back calls after a large reallocation came back wrong by O(1).  The scratch
    is now cached per (device, stream)."""
    rng = np.random.default_rng(1)
    n, n_per = 200_000, 2048
    t = np.sort(rng.uniform(0, float(n), n))
    y = np.sin(2 * np.pi * t / 13.7) + 0.1 * rng.standard_normal(n)
    m = so.stringlength_scale(y)
    periods = np.linspace(1.0, 100.0, n_per)
    sigma = np.var(y, ddof=1)
    first = _cabi.pdm_scan(t, y, periods, 5, 2, sigma)
    _cabi.stringlength_scan(t, m, periods[:64])                 # grows the cached workspace (hipFree + hipMalloc)
    for _ in range(3):
        assert np.array_equal(_cabi.pdm_scan(t, y, periods, 5, 2, sigma), first)
    pick = np.array([0, n_per // 3, n_per - 1])
    np.testing.assert_allclose(first[pick], co.pdm_scan(t, y, periods[pick], 5, 2), rtol=RTOL)
    aov = _cabi.aov_scan(t, y, periods, 10)
    assert np.array_equal(_cabi.aov_scan(t, y, periods, 10), aov)
