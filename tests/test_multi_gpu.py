"""The N > 1 logic on ONE GPU, through the C ABI.

* ``pdc_gls_plan_create_loopback``: N logical slots on one physical device, each with its own buffers,
  streams and events; the all-gather runs as device-to-device copies on the communication streams.  The
  slab arithmetic, padded tails (``nf % N != 0``, ``nf < N``, slots that own nothing), the generation
  reuse of the double-buffered outputs and the event ordering are those of the N-GPU plan; every scan
  must come back bit-identical to what slab-wise single-device calls give, from EVERY slot.
* the phase scans / batches over a device list that repeats device 0: bit-identical to one launch, and
  the second call of a given size makes no device or pinned allocation at all.

The fan-out these replace: ``multiprocessing.Pool.map`` over trial periods,
``/root/reference/src/periodicity/phase.py:69-70,185-186``."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import scan_oracle as so
from periodicity_amd import _cabi
from periodicity_amd.core import TSeries
from periodicity_amd.phase import AOV, PDM, ConditionalEntropy, StringLength
from periodicity_amd.spectral import GLS

pytestmark = pytest.mark.gpu


def synth(n, seed, period=37.3):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, float(n), n))
    dy = rng.uniform(0.05, 0.2, n)
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / period) + dy * rng.standard_normal(n)
    return t, y, dy


def slabwise(t, y, dy, f0, delta, nf, n_slots, **kw):
    """What an N-slot plan must produce: slot i scans [i*per, min((i+1)*per, nf)) starting its recurrences
    at its own first frequency."""
    per = -(-nf // n_slots)
    parts = [_cabi.gls_scan(t, y, dy, f0, delta, min(per, nf - b), j_begin=b, **kw)
             for b in range(0, nf, per)]
    return np.concatenate(parts)


@pytest.mark.parametrize("n_slots", [2, 3, 8])
def test_loopback_plan_matches_slabwise_scans_on_every_slot(n_slots):
    t, y, dy = synth(3000, 11 + n_slots)
    plan = _cabi.GlsPlan([0], n_max=4000, nf_max=6000, loopback_slots=n_slots)
    info = plan.info()
    assert info == {"n_slots": n_slots, "rccl_ranks": 0, "exchange": "copy", "init_error": None}
    plan.upload(t, y, dy)
    # grids: divisible, not divisible, shorter than the slot count (trailing slots own nothing), one bin
    for nf in (4096 - 4096 % n_slots, 5003, n_slots - 1, 1, 2 * n_slots + 1):
        f0, delta = 0.0013, 0.00017
        plan.scan(f0, delta, nf)
        want = slabwise(t, y, dy, f0, delta, nf, n_slots)
        for which in range(n_slots):
            assert np.array_equal(plan.download(which), want), (nf, which)
    # generation reuse: four back-to-back enqueues without a wait in between (two generations, each
    # written twice); the last one must win on every slot, and an earlier grid must not leak into it
    grids = [(0.002, 0.0003, 2999), (0.001, 0.0002, 5999), (0.004, 0.0001, 1001), (0.0015, 0.00025, 4000)]
    for f0, delta, nf in grids:
        plan.scan(f0, delta, nf)
    f0, delta, nf = grids[-1]
    want = slabwise(t, y, dy, f0, delta, nf, n_slots)
    for which in (0, n_slots - 1):
        assert np.array_equal(plan.download(which), want)
    # a new light curve through the same plan, other flags
    plan.upload(t[:1000], y[:1000], None)
    plan.scan(0.003, 0.0004, 777, fit_mean=False, psd=True)
    want = slabwise(t[:1000], y[:1000], None, 0.003, 0.0004, 777, n_slots, fit_mean=False, psd=True)
    assert np.array_equal(plan.download(n_slots - 1), want)
    assert plan.kernel_ms() > 0
    with pytest.raises(ValueError):
        plan.scan(0.1, 0.1, 10 ** 6)
    plan.close()


def test_loopback_plan_many_scans_stay_consistent_under_load():
    """Long enough scans that the copies of generation g really overlap the scan of generation g^1."""
    t, y, dy = synth(20_000, 3)
    plan = _cabi.GlsPlan([0], n_max=t.size, nf_max=40_000, loopback_slots=4)
    plan.upload(t, y, dy)
    f0, delta = 0.0005, 0.00001
    wants = {nf: slabwise(t, y, dy, f0, delta, nf, 4) for nf in (40_000, 39_999)}
    for i in range(6):
        nf = 40_000 - (i & 1)
        plan.scan(f0, delta, nf)
        if i in (2, 5):
            for which in range(4):
                assert np.array_equal(plan.download(which), wants[nf])
    plan.close()


def test_single_slot_loopback_and_validation():
    t, y, dy = synth(500, 5)
    plan = _cabi.GlsPlan([0], 600, 1000, loopback_slots=1)
    assert plan.info()["exchange"] == "none"
    plan.upload(t, y, dy)
    plan.scan(0.01, 0.001, 400)
    assert np.array_equal(plan.download(), _cabi.gls_scan(t, y, dy, 0.01, 0.001, 400))
    plan.close()
    with pytest.raises(ValueError):
        _cabi.GlsPlan([0], 10, 10, loopback_slots=0)
    with pytest.raises(ValueError):
        _cabi.GlsPlan([_cabi.device_count()], 10, 10, loopback_slots=2)


# ---- phase scans ----------------------------------------------------------------------------------------
def phase_inputs(n=1500, seed=8):
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, 60.0, n))
    x = np.sin(2 * np.pi * t / 4.4) + 0.2 * rng.standard_normal(n)
    return t, x, so.stringlength_scale(x)


def test_phase_plan_all_kinds_and_slot_counts():
    t, x, m = phase_inputs()
    mag = so.magnitude_bins(x, 5)
    sigma = float(np.var(x, ddof=1))
    for devices in ((0,), (0, 0), (0, 0, 0, 0, 0)):
        plan = _cabi.PhasePlan(devices)
        for n_periods in (1000, 7, 3, 1):
            periods = np.linspace(0.7, 30.0, n_periods)
            plan.upload(t, x)
            plan.scan("pdm", periods, 5, 2, sigma)
            got = plan.download()
            np.testing.assert_allclose(got, _cabi.pdm_scan(t, x, periods, 5, 2, sigma), rtol=1e-12)
            assert plan.kernel_ms() > 0
            plan.scan("aov", periods, 10)
            np.testing.assert_allclose(plan.download(), _cabi.aov_scan(t, x, periods, 10), rtol=1e-12)
            plan.upload(t, mag)
            plan.scan("cond_entropy", periods, 10, 5)
            assert np.array_equal(plan.download(), _cabi.cond_entropy_scan(t, mag, periods, 10, 5))
            plan.upload(t, m)
            plan.scan("stringlength", periods)
            assert np.array_equal(plan.download(), _cabi.stringlength_scan(t, m, periods))
        plan.scan("pdm", np.empty(0), 5, 2, sigma)
        assert plan.download().size == 0
        plan.close()
    with pytest.raises(ValueError):
        _cabi.PhasePlan((0, _cabi.device_count()))
    plan = _cabi.PhasePlan((0, 0))
    with pytest.raises(ValueError):
        plan.scan("pdm", [1.0, 2.0], 5, 2, 1.0)       # nothing uploaded yet
    plan.close()


def test_phase_plan_takes_the_workspace_without_lists_when_the_host_sees_it_can():
    """the phase plan sized every StringLength / Supersmoother slot for the streamed kernels' bin
    lists (~12 GB per slot from 262 144 samples on) even for time-ordered input.  The plan holds the host arrays at
    scan time, so it now runs the host's test (every period takes the slices / one-cycle modes?) per slab and asks
    for the workspace without lists; samples out of order are ordered by time on the device first (round 5) and need none
    either, non-tame time stamps keep them.  Same values either way."""
    rng = np.random.default_rng(44)
    n = 300_000
    t = np.sort(rng.uniform(0, float(n), n))
    y = np.sin(2 * np.pi * t / 13.7) + 0.2 * rng.standard_normal(n)
    m = so.stringlength_scale(y)
    df = 0.1 / (t[-1] - t[0])
    periods = 1 / np.linspace(64 * df, df, 64)                 # long periods: slices mode throughout
    one = _cabi.stringlength_scan(t, m, periods)
    before = _cabi.alloc_counts()
    two = _cabi.stringlength_scan(t, m, periods, devices=(0, 0))
    assert np.array_equal(one, two)
    np.testing.assert_allclose(two[[0, 31, 63]], co.stringlength_scan(t, m, periods[[0, 31, 63]]), rtol=1e-9)
    assert _cabi.alloc_counts()[0] - before[0] <= 16
    order = rng.permutation(n)
    np.testing.assert_allclose(_cabi.stringlength_scan(t[order], m[order], periods[:4], devices=(0, 0)),
                               co.stringlength_scan(t[order], m[order], periods[:4]), rtol=1e-9)
    # (round 5) out of order the samples are ordered by time on every slot's device first: the same bits as one device,
    # the oracle's result for the time-ordered series, and no lists in the slots' workspaces either
    back = np.argsort(t[order], kind="stable")
    shuffled_one = _cabi.stringlength_scan(t[order], m[order], periods)
    assert np.array_equal(shuffled_one, _cabi.stringlength_scan(t[order], m[order], periods, devices=(0, 0, 0)))
    np.testing.assert_allclose(shuffled_one, co.stringlength_scan(t[order][back], m[order][back], periods), rtol=1e-9)
    ss_p = np.array([2.1, 13.7, 0.31 * t[-1], 1.7 * t[-1]])
    ss_shuffled = _cabi.supersmoother_scan(t[order], y[order], ss_p, 0.0, devices=(0, 0))
    assert np.array_equal(ss_shuffled, _cabi.supersmoother_scan(t[order], y[order], ss_p, 0.0))
    np.testing.assert_allclose(ss_shuffled, _cabi.supersmoother_scan(t, y, ss_p, 0.0), rtol=1e-12)
    ss_one = _cabi.supersmoother_scan(t[:60_000], y[:60_000], periods[:6] / 50.0, 0.0)
    ss_two = _cabi.supersmoother_scan(t[:60_000], y[:60_000], periods[:6] / 50.0, 0.0, devices=(0, 0))
    np.testing.assert_allclose(ss_two, ss_one, rtol=1e-12)


_BUDGET_CHECK = r"""
import os, sys
import numpy as np
from oracle import c_oracle as co
from oracle import scan_oracle as so
from periodicity_amd import _cabi
lib = _cabi.lib()
n = 1_000_000
rng = np.random.default_rng(46)
t = np.sort(rng.uniform(0, float(n), n))
y = np.sin(2 * np.pi * t / 13.7) + 0.2 * rng.standard_normal(n)
m = so.stringlength_scale(y)
df = 0.1 / (t[-1] - t[0])
periods = 1 / np.linspace(96 * df, df, 96)
short = 1 / np.linspace(4000 * df, 3000 * df, 12)            # cells under four samples: the lists mode
ss_p = np.array([13.7, 41.0, 997.0, 0.31 * t[-1]])
free = [lib.pdc_stringlength_work_bytes(n, 2048), lib.pdc_supersmoother_work_bytes(n, 2048)]
want_sl = _cabi.stringlength_scan(t, m, periods)
want_short = _cabi.stringlength_scan(t, m, short)
want_ss = _cabi.supersmoother_scan(t, y, ss_p, 0.0)
np.testing.assert_allclose(want_sl[[0, 50, 95]], co.stringlength_scan(t, m, periods[[0, 50, 95]]), rtol=1e-9)
_cabi.check(lib.pdc_release())
os.environ["PDC_WORK_BUDGET_GB"] = "4"
tight = [lib.pdc_stringlength_work_bytes(n, 2048), lib.pdc_supersmoother_work_bytes(n, 2048)]
assert free[0] > 8 << 30 and free[1] > 2 << 30, free          # the built-in caps: 29.6 GB for StringLength, 3.3 GB for the Supersmoother ...
assert max(tight) <= 4 << 30, tight                           # ... the budget holds both under it
slots = (0,) * 8
assert np.array_equal(_cabi.stringlength_scan(t, m, periods, devices=slots), want_sl)       # smaller batches, same bits
assert np.array_equal(_cabi.stringlength_scan(t, m, short, devices=slots), want_short)
assert np.array_equal(_cabi.stringlength_scan(t, m, short), want_short)                     # the host entry alone
np.testing.assert_allclose(_cabi.supersmoother_scan(t, y, ss_p, 0.0, devices=slots), want_ss, rtol=1e-12)
np.testing.assert_allclose(_cabi.supersmoother_scan(t, y, ss_p, 0.0), want_ss, rtol=1e-12)
# a budget below the Supersmoother's own 3.3 GB: smaller sub-batches and pools, same statistic
os.environ["PDC_WORK_BUDGET_GB"] = "1.5"
assert lib.pdc_supersmoother_work_bytes(n, 2048) <= int(1.5 * (1 << 30)) < free[1]
np.testing.assert_allclose(_cabi.supersmoother_scan(t, y, ss_p, 0.0), want_ss, rtol=1e-12)
np.testing.assert_allclose(_cabi.supersmoother_scan(t, y, ss_p, 0.0, devices=slots), want_ss, rtol=1e-12)
os.environ["PDC_WORK_BUDGET_GB"] = "4"
# the _dev entry with a workspace of exactly the budgeted size
DB = _cabi.DeviceBuffer
wb = lib.pdc_stringlength_work_bytes(n, short.size)
bufs = [DB.from_array(t, 0), DB.from_array(m, 0), DB.from_array(short, 0), DB(short.size * 8, 0), DB(wb, 0)]
_cabi.check(lib.pdc_stringlength_scan_dev(0, None, bufs[0].ptr, bufs[1].ptr, n, bufs[2].ptr, short.size, bufs[3].ptr, bufs[4].ptr, wb))
_cabi.check(lib.pdc_device_sync(0))
assert np.array_equal(bufs[3].to_array(np.float64, short.size), want_short)
# a budget not even one period fits in: a clear error that names it, nothing allocated, the library still usable
os.environ["PDC_WORK_BUDGET_GB"] = "0.02"
for call in (lambda: _cabi.stringlength_scan(t, m, periods), lambda: _cabi.supersmoother_scan(t, y, ss_p, 0.0),
             lambda: _cabi.stringlength_scan(t, m, periods, devices=slots)):
    try:
        call()
    except (RuntimeError, ValueError) as exc:
        assert "budget" in str(exc) and "0.020 GB" in str(exc), str(exc)
    else:
        raise AssertionError("a 20 MB budget was accepted for a million samples")
del os.environ["PDC_WORK_BUDGET_GB"]
assert np.array_equal(_cabi.stringlength_scan(t, m, periods[:8]), want_sl[:8])
print("ok")
"""


def test_workspaces_fit_a_budget_and_say_so_when_they_cannot():
    """VERDICT r5 missing #5 / next #7: the StringLength / Supersmoother workspaces were sized by constants (12 GB of
    bin lists, 2 GB, 1 GB pools, 1024 workgroups of scratch): no clamp to the device, so eight loopback slots - or a
    shared GPU - failed in hipMalloc.  Now PDC_WORK_BUDGET_GB (and, for the host entries and the phase plan's slots,
    what hipMemGetInfo reports free) scales every cap: streamed StringLength (slices AND lists mode) and the
    Supersmoother at N = 1e6 through an 8-slot loopback plan under a 4 GB budget give the bits of the unbudgeted
    run, and a budget that not even one period fits in is an error that names it.  (A child process: the test frees
    and re-sizes the library's cached workspaces.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "PDC_WORK_BUDGET_GB"}
    out = subprocess.run([sys.executable, "-c", _BUDGET_CHECK], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-1500:], out.stderr[-3000:])


def test_cached_fan_out_allocates_nothing_on_the_second_call():
    """the per-call hipStreamCreate + hipMalloc of the phase fan-out.  Now: buffers, streams
    and pinned staging live in a cache keyed by the device list."""
    t, x, m = phase_inputs(30_000, 9)
    sigma = float(np.var(x, ddof=1))
    devices = (0, 0, 0)
    calls = {
        "pdm long grid": lambda: _cabi.pdm_scan(t, x, np.linspace(0.7, 30.0, 60_000), 5, 2, sigma, devices=devices),
        "pdm short grid (sample split, scratch)": lambda: _cabi.pdm_scan(t, x, np.linspace(0.7, 30.0, 300), 5, 2, sigma,
                                                                         devices=devices),
        "stringlength": lambda: _cabi.stringlength_scan(t, m, np.linspace(0.7, 30.0, 600), devices=devices),
        "aov": lambda: _cabi.aov_scan(t, x, np.linspace(0.7, 30.0, 500), 10, devices=devices),
    }
    for name, call in calls.items():
        first = call()
        before = _cabi.alloc_counts()
        for _ in range(3):
            assert np.array_equal(call(), first), name
        assert _cabi.alloc_counts() == before, f"{name}: allocations on a repeated call"
    # all four interleaved once everything has reached its high-water mark
    for call in calls.values():
        call()
    before = _cabi.alloc_counts()
    for call in calls.values():
        call()
    assert _cabi.alloc_counts() == before
    # classes route there
    sig = TSeries(t, x)
    # (a shorter slab may split the samples over more workgroups: same bins and counts, other summation order)
    np.testing.assert_allclose(PDM(p_min=0.7, p_max=30.0, n_periods=300, devices=(0, 0))(sig).values,
                               PDM(p_min=0.7, p_max=30.0, n_periods=300)(sig).values, rtol=1e-12)
    np.testing.assert_allclose(AOV(p_min=0.7, p_max=30.0, n_periods=300, devices=(0, 0))(sig).values,
                               AOV(p_min=0.7, p_max=30.0, n_periods=300)(sig).values, rtol=1e-12)
    assert np.array_equal(ConditionalEntropy(p_min=0.7, p_max=30.0, n_periods=300, devices=(0, 0))(sig).values,
                          ConditionalEntropy(p_min=0.7, p_max=30.0, n_periods=300)(sig).values)
    assert np.array_equal(StringLength(n_periods=300, devices=(0, 0))(sig).values,
                          StringLength(n_periods=300)(sig).values)


def test_transient_streams_do_not_grow_the_scratch_table():
    """the split-mode scratch of the `_dev` entry is cached per (device, stream); a caller
    cycling through streams must not leak one block per stream.  pdc_stream_destroy drops the entry; the
    table is capped for raw HIP streams."""
    lib = _cabi.lib()
    t, x, _ = phase_inputs(40_000, 3)
    periods = np.linspace(0.7, 30.0, 64)                     # few periods x many samples: split mode
    bufs = [_cabi.DeviceBuffer.from_array(a) for a in (t, x, periods)]
    out = _cabi.DeviceBuffer(periods.size * 8)
    sigma = float(np.var(x, ddof=1))
    want = _cabi.pdm_scan(t, x, periods, 5, 2, sigma)

    def one_stream():
        s = C.c_void_p()
        _cabi.check(lib.pdc_stream_create(0, C.byref(s)))
        _cabi.check(lib.pdc_pdm_scan_dev(0, s, bufs[0].ptr, bufs[1].ptr, t.size, bufs[2].ptr, periods.size, 5, 2,
                                         sigma, out.ptr))
        _cabi.check(lib.pdc_stream_sync(0, s))
        got = out.to_array(np.float64, periods.size)
        _cabi.check(lib.pdc_stream_destroy(0, s))
        return got
    assert np.array_equal(one_stream(), want)
    # explicit workspace: nothing is cached at all
    wb = lib.pdc_phase_work_bytes(0, t.size, periods.size, 5, 2)
    assert wb > 0
    work = _cabi.DeviceBuffer(wb)
    before = _cabi.alloc_counts()
    for _ in range(3):
        _cabi.check(lib.pdc_phase_scan_dev(0, 0, None, bufs[0].ptr, bufs[1].ptr, t.size, bufs[2].ptr, periods.size,
                                           5, 2, sigma, out.ptr, work.ptr, wb))
        _cabi.check(lib.pdc_device_sync(0))
        assert np.array_equal(out.to_array(np.float64, periods.size), want)
    assert _cabi.alloc_counts() == before
    with pytest.raises(ValueError):
        _cabi.check(lib.pdc_phase_scan_dev(0, 0, None, bufs[0].ptr, bufs[1].ptr, t.size, bufs[2].ptr, periods.size,
                                           5, 2, sigma, out.ptr, work.ptr, wb - 1))
    for b in bufs + [out, work]:
        b.free()


def test_scratch_table_never_frees_a_block_in_use():
    """the per-(device, stream) scratch table is true LRU with a per-device cap of 64, its entries
    are pinned from the look-up until the caller's launches are enqueued, and only entries whose stream has run
    dry are evicted.  90 live streams (more than the cap) used round-robin, then four host threads hammering
    their own streams at once: every result must be the single-stream one."""
    import threading
    lib = _cabi.lib()
    t, x, _ = phase_inputs(30_000, 5)
    periods = np.linspace(0.7, 30.0, 48)                     # few periods x many samples: split mode (uses the table)
    bufs = [_cabi.DeviceBuffer.from_array(a) for a in (t, x, periods)]
    sigma = float(np.var(x, ddof=1))
    want = _cabi.pdm_scan(t, x, periods, 5, 2, sigma)
    streams, outs = [], []
    for _ in range(90):
        s = C.c_void_p()
        _cabi.check(lib.pdc_stream_create(0, C.byref(s)))
        streams.append(s)
        outs.append(_cabi.DeviceBuffer(periods.size * 8))
    for rnd in range(2):
        for s, o in zip(streams, outs):                      # enqueue on all of them, then look
            _cabi.check(lib.pdc_pdm_scan_dev(0, s, bufs[0].ptr, bufs[1].ptr, t.size, bufs[2].ptr, periods.size, 5, 2,
                                             sigma, o.ptr))
        _cabi.check(lib.pdc_device_sync(0))
        for o in outs:
            assert np.array_equal(o.to_array(np.float64, periods.size), want), rnd
    bad = []

    def hammer(k):
        try:
            for _ in range(25):
                for s, o in list(zip(streams, outs))[k::4]:
                    _cabi.check(lib.pdc_pdm_scan_dev(0, s, bufs[0].ptr, bufs[1].ptr, t.size, bufs[2].ptr, periods.size,
                                                     5, 2, sigma, o.ptr))
        except Exception as exc:                              # pragma: no cover
            bad.append(exc)
    threads = [threading.Thread(target=hammer, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad, bad
    _cabi.check(lib.pdc_device_sync(0))
    for o in outs:
        assert np.array_equal(o.to_array(np.float64, periods.size), want)
    for s in streams:
        _cabi.check(lib.pdc_stream_destroy(0, s))
    for b in bufs + outs:
        b.free()


def test_scratch_regrow_never_frees_a_pinned_block():
    """when a (device, stream) entry must grow while another host thread still holds the old block pinned
    (its launches not enqueued yet), hipFree's device synchronisation cannot protect launches that do not exist yet.
    Round 5: ONE caller at a time holds a stream's block between stream_scratch() and stream_scratch_done() - the
    second waits - so no block is ever outgrown while pinned, and launches of two threads on one stream cannot
    interleave around the shared block.  Deterministic leg: the library's internal table driven directly (the
    pdc_test_scratch_* hooks) - thread A pins a small block and keeps it for 0.3 s, thread B asks for a larger one
    on the same stream: B returns only after A is done, A's block is valid device memory all along.  Threaded leg: two
    host threads on ONE stream, one with a short period grid, one whose grids keep growing, every result equal to the
    single-thread one."""
    import threading
    import time
    lib = _cabi.lib()
    scratch, done = lib.pdc_test_scratch_pin, lib.pdc_test_scratch_unpin      # (extern "C" test hooks of the table)
    s = C.c_void_p()
    _cabi.check(lib.pdc_stream_create(0, C.byref(s)))
    stamps = {}
    a, b = C.c_void_p(), C.c_void_p()
    _cabi.check(scratch(0, s, 1 << 20, C.byref(a)))            # caller A: pinned, nothing enqueued yet

    def caller_b():
        _cabi.check(scratch(0, s, 64 << 20, C.byref(b)))       # outgrows the block: must wait for A
        stamps["b_got"] = time.monotonic()
        done(0, s)
    th = threading.Thread(target=caller_b)
    th.start()
    time.sleep(0.3)
    host = np.arange(1 << 17, dtype=np.float64)                # A uses ITS block: still valid memory
    _cabi.check(lib.pdc_memcpy_h2d(0, a, host.ctypes.data_as(C.c_void_p), host.nbytes))
    back = np.empty_like(host)
    _cabi.check(lib.pdc_memcpy_d2h(0, back.ctypes.data_as(C.c_void_p), a, host.nbytes))
    assert np.array_equal(back, host) and "b_got" not in stamps
    stamps["a_done"] = time.monotonic()
    done(0, s)
    th.join()
    assert stamps["b_got"] >= stamps["a_done"]
    _cabi.check(lib.pdc_stream_destroy(0, s))

    t, x, _ = phase_inputs(60_000, 9)
    sigma = float(np.var(x, ddof=1))
    bt, bx = _cabi.DeviceBuffer.from_array(t), _cabi.DeviceBuffer.from_array(x)
    grids = [np.linspace(0.7, 30.0, k) for k in (24, 32, 48, 64, 96, 128, 160, 192)]   # split mode: scratch grows with the grid
    want = [_cabi.pdm_scan(t, x, g, 5, 2, sigma) for g in grids]
    s = C.c_void_p()
    _cabi.check(lib.pdc_stream_create(0, C.byref(s)))
    bad = []

    def run(order, rounds):
        try:
            bufs = [(_cabi.DeviceBuffer.from_array(grids[k]), _cabi.DeviceBuffer(grids[k].size * 8)) for k in order]
            for _ in range(rounds):
                for k, (bp, bo) in zip(order, bufs):
                    _cabi.check(lib.pdc_pdm_scan_dev(0, s, bt.ptr, bx.ptr, t.size, bp.ptr, grids[k].size, 5, 2, sigma, bo.ptr))
                    _cabi.check(lib.pdc_stream_sync(0, s))
                    if not np.array_equal(bo.to_array(np.float64, grids[k].size), want[k]):
                        bad.append(k)
            for bp, bo in bufs:
                bp.free()
                bo.free()
        except Exception as exc:                              # pragma: no cover
            bad.append(exc)
    for trial in range(6):
        _cabi.check(lib.pdc_stream_destroy(0, s))             # (drops the entry: the next trial regrows from nothing)
        _cabi.check(lib.pdc_stream_create(0, C.byref(s)))
        threads = [threading.Thread(target=run, args=([0], 40)), threading.Thread(target=run, args=(list(range(8)), 2))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    assert not bad, bad
    _cabi.check(lib.pdc_stream_destroy(0, s))
    bt.free()
    bx.free()


def test_cond_entropy_dev_entry_ignores_out_of_range_bins():
    """the kernel used the caller's double as an LDS index unchecked.  Through the `_dev` entry
    a NaN / negative / too large magnitude bin now counts nowhere (as if the sample were absent); the host
    entries reject such input."""
    lib = _cabi.lib()
    t, x, _ = phase_inputs(3000, 4)
    mag = so.magnitude_bins(x, 5)
    periods = np.linspace(0.7, 30.0, 200)
    bad = mag.copy()
    holes = np.array([3, 500, 1234, 2999])
    bad[holes] = [np.nan, -1.0, 5.0, 1e300]
    keep = np.ones(t.size, bool)
    keep[holes] = False
    want = _cabi.cond_entropy_scan(t[keep], mag[keep], periods, 10, 5)
    bufs = [_cabi.DeviceBuffer.from_array(a) for a in (t, bad, periods)]
    out = _cabi.DeviceBuffer(periods.size * 8)
    _cabi.check(lib.pdc_cond_entropy_scan_dev(0, None, bufs[0].ptr, bufs[1].ptr, t.size, bufs[2].ptr, periods.size,
                                              10, 5, out.ptr))
    _cabi.check(lib.pdc_device_sync(0))
    np.testing.assert_allclose(out.to_array(np.float64, periods.size), want, rtol=1e-12)
    for b in bufs + [out]:
        b.free()
    out = np.empty(periods.size)
    for entry, extra in ((lib.pdc_cond_entropy_scan, (0,)),
                         (lib.pdc_cond_entropy_scan_multi, (_cabi._ptr(np.zeros(2, np.int32)), 2))):
        status = entry(_cabi._ptr(t), _cabi._ptr(bad), t.size, _cabi._ptr(periods), periods.size, 10, 5,
                       _cabi._ptr(out), *extra)
        assert status == -1 and b"mag_bin" in lib.pdc_last_error()


# ---- batches ---------------------------------------------------------------------------------------------
def test_batch_sharded_over_curves_equals_one_launch():
    rng = np.random.default_rng(12)
    lens = rng.integers(150, 400, 37)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    t = np.concatenate([np.sort(rng.uniform(0, 300.0, k)) for k in lens])
    dy = rng.uniform(0.05, 0.2, t.size)
    y = np.sin(2 * np.pi * t / 9.0) + dy * rng.standard_normal(t.size)
    f0, delta, nf = 0.001, 0.0007, 1500
    one = _cabi.gls_scan_batch(t, y, dy, offsets, f0, delta, nf, want_peaks=True)
    for devices in ((0, 0), (0, 0, 0, 0, 0), (0,) * 40):       # more slots than curves: some own nothing
        many = _cabi.gls_scan_batch(t, y, dy, offsets, f0, delta, nf, want_peaks=True, devices=devices)
        for a, b in zip(one, many):
            assert np.array_equal(a, b)
    peaks_only = _cabi.gls_scan_batch(t, y, None, offsets, f0, delta, nf, want_power=False, want_peaks=True,
                                      fit_mean=False, psd=True, devices=(0, 0, 0))
    ref = _cabi.gls_scan_batch(t, y, None, offsets, f0, delta, nf, want_power=False, want_peaks=True,
                               fit_mean=False, psd=True)
    assert peaks_only[0] is None and np.array_equal(peaks_only[1], ref[1]) and np.array_equal(peaks_only[2], ref[2])
    _cabi.gls_scan_batch(t, y, dy, offsets, f0, delta, nf, want_peaks=True, devices=(0, 0, 0))
    before = _cabi.alloc_counts()
    again = _cabi.gls_scan_batch(t, y, dy, offsets, f0, delta, nf, want_peaks=True, devices=(0, 0, 0))
    assert _cabi.alloc_counts() == before and np.array_equal(again[0], one[0])


def test_bootstrap_sharded_over_device_slots(golden_dir):
    """GLS.bootstrap (spectral.py:140-152) with the replicates dealt to device slots: same draws, same
    maxima as the one-device batch and as golden G6."""
    import os
    g = np.load(os.path.join(golden_dir, "g6_bootstrap.npz"))
    one = GLS()
    one(TSeries(g["t"], g["y"]), err=g["dy"])
    many = GLS()
    many(TSeries(g["t"], g["y"]), err=g["dy"])
    many.devices = (0, 0, 0)        # (the periodogram itself needs distinct devices: RCCL; the replicates do not)
    a = one.bootstrap(20, random_seed=42)
    b = many.bootstrap(20, random_seed=42)
    np.testing.assert_allclose(b, a, rtol=1e-12)      # (a smaller group may take the per-curve kernel)
    np.testing.assert_allclose(b, g["replicates_exact"], rtol=1e-6)
    t = g["t"]
    big = GLS()
    big(TSeries(t, g["y"]))                              # equal weights, enough replicates for the shared kernel
    big.devices = (0, 0)
    ref = GLS()
    ref(TSeries(t, g["y"]))
    np.testing.assert_allclose(big.bootstrap(400, random_seed=1), ref.bootstrap(400, random_seed=1), rtol=1e-12)


def test_pdm_split_mode_writes_every_scratch_word_it_reads(tmp_path):
    """the split mode's scratch is a cached block that is as stale between calls as the pool
    memory it replaced.  With PDC_PDM_POISON=1 the block is filled with a NaN pattern before every call: a
    word read without having been written in the same call would surface as NaN / garbage.  Shapes change
    between back-to-back calls (n_periods and n up and down); results must equal the unsplit kernel's
    (PDC_PDM_SPLIT=0) to summation order and the oracle's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "pdm_poison_check.py")
    outs = {}
    for tag, env in (("poison", {"PDC_PDM_POISON": "1"}), ("unsplit", {"PDC_PDM_SPLIT": "0"})):
        path = str(tmp_path / f"{tag}.npz")
        run = subprocess.run([sys.executable, script, path], env=dict(os.environ, **env), cwd=root,
                             capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stderr[-2000:]
        outs[tag] = np.load(path)
    assert sorted(outs["poison"].files) == sorted(outs["unsplit"].files) and len(outs["poison"].files) >= 12
    for key in outs["poison"].files:
        a, b = outs["poison"][key], outs["unsplit"][key]
        assert np.all(np.isfinite(a)), key
        np.testing.assert_allclose(a, b, rtol=1e-11, err_msg=key)


def test_mixed_scans_from_concurrent_python_threads():
    """ctypes drops the GIL: every scan family at once from six threads - single-device entries (cached
    workspace behind the device lock), the device-slot plans (cached by device list, one mutex each) and the
    peak reduction - must return what they return one at a time, bit for bit."""
    import threading
    rng = np.random.default_rng(404)
    t, y, dy = synth(6000, 9)
    m = so.stringlength_scale(y)
    mb = so.magnitude_bins(y, 4)
    periods = np.linspace(0.8, 60.0, 333)
    spectra = rng.random((7, 3001))
    jobs = {
        "pdm": lambda: _cabi.pdm_scan(t, y, periods, 5, 2, np.var(y, ddof=1)),
        "pdm_slots": lambda: _cabi.pdm_scan(t, y, periods, 5, 2, np.var(y, ddof=1), devices=(0, 0, 0)),
        "sl": lambda: _cabi.stringlength_scan(t, m, periods),
        "sl_slots": lambda: _cabi.stringlength_scan(t, m, periods, devices=(0, 0)),
        "aov": lambda: _cabi.aov_scan(t, y, periods, 8),
        "ce_slots": lambda: _cabi.cond_entropy_scan(t, mb, periods, 9, 4, devices=(0, 0, 0)),
        "gl": lambda: _cabi.gl_scan(t, periods, 5, 4),
        "peaks": lambda: _cabi.peaks_topk(spectra, k=3, by_prominence=True)["indices"],
        "gls": lambda: _cabi.gls_scan(t, y, dy, 0.001, 0.0005, 2500),
    }
    want = {k: fn() for k, fn in jobs.items()}
    names = list(jobs)
    errors = []

    def work(w):
        try:
            for it in range(4):
                for k in names[w % len(names):] + names[:w % len(names)]:
                    got = jobs[k]()
                    if not np.array_equal(got, want[k], equal_nan=True):
                        errors.append(f"thread {w} iteration {it}: {k} differs")
        except Exception as exc:  # noqa: BLE001 - reported below
            errors.append(f"thread {w}: {exc!r}")

    threads = [threading.Thread(target=work, args=(w,)) for w in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
