"""ctypes binding of ``libperiodicity_hip.so`` (``include/periodicity_hip.h``).

This is the only place the Python host code touches native code.  There is **no CPU fallback**:
if the shared library is missing, or no gfx950 device is visible, every scan raises — loudly.
"""
import ctypes as C
import os
import threading

import numpy as np

_LIB_NAME = "libperiodicity_hip.so"
_lib = None
_lock = threading.Lock()

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)

STATUS_EXC = {-1: ValueError, -2: RuntimeError, -3: RuntimeError, -4: RuntimeError,
              -5: MemoryError}

# name -> (restype, argtypes); kept in the order of include/periodicity_hip.h
_VP, _I, _L, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
PROTOTYPES = {
    "pdc_last_error": (C.c_char_p, []),
    "pdc_version": (_I, []),
    "pdc_device_count": (_I, [C.POINTER(_I)]),
    "pdc_device_info": (_I, [_I, C.c_char_p, _I, C.POINTER(_I), c_int64_p, C.POINTER(_I)]),
    "pdc_release": (_I, []),
    "pdc_alloc_counts": (_I, [c_int64_p, c_int64_p]),
    "pdc_malloc": (_I, [_I, _L, C.POINTER(_VP)]),
    "pdc_free": (_I, [_I, _VP]),
    "pdc_memcpy_h2d": (_I, [_I, _VP, _VP, _L]),
    "pdc_memcpy_d2h": (_I, [_I, _VP, _VP, _L]),
    "pdc_memset": (_I, [_I, _VP, _I, _L]),
    "pdc_stream_create": (_I, [_I, C.POINTER(_VP)]),
    "pdc_stream_destroy": (_I, [_I, _VP]),
    "pdc_stream_sync": (_I, [_I, _VP]),
    "pdc_device_sync": (_I, [_I]),
    "pdc_test_scratch_pin": (_I, [_I, _VP, _L, C.POINTER(_VP)]),
    "pdc_test_scratch_unpin": (_I, [_I, _VP]),
    "pdc_event_create": (_I, [_I, C.POINTER(_VP)]),
    "pdc_event_destroy": (_I, [_I, _VP]),
    "pdc_event_record": (_I, [_I, _VP, _VP]),
    "pdc_event_elapsed_ms": (_I, [_I, _VP, _VP, C.POINTER(C.c_float)]),
    "pdc_clock_probe": (_I, [_I, _VP, _I, C.POINTER(C.c_float), C.POINTER(_D), C.POINTER(_D)]),
    "pdc_bglst_scan": (_I, [_VP, _VP, _VP, _L, _D, _D, _L, _L, _VP, _VP, _I]),
    "pdc_bglst_scan_dev": (_I, [_I, _VP, _VP, _VP, _VP, _L, _D, _D, _L, _L, _VP, _VP, _VP, _L]),
    "pdc_gls_scan": (_I, [_VP, _VP, _VP, _L, _D, _D, _L, _L, _I, _I, _VP, _I]),
    "pdc_gls_scan_batch": (_I, [_VP, _VP, _VP, _VP, _L, _I, _D, _D, _L, _L, _I, _I,
                                _VP, _VP, _VP, _I]),
    "pdc_gls_scan_batch_multi": (_I, [_VP, _VP, _VP, _VP, _L, _I, _D, _D, _L, _I, _I, _VP, _VP, _VP, _VP, _I]),
    "pdc_gls_scan_multi": (_I, [_VP, _VP, _VP, _L, _D, _D, _L, _I, _I, _VP, _VP, _I]),
    "pdc_gls_bootstrap": (_I, [_VP, _VP, _VP, _L, _VP, _L, _D, _D, _L, _I, _I, _I, _VP, _VP, _VP, _I]),
    "pdc_gls_bootstrap_work_bytes": (_L, [_L, _L, _L]),
    "pdc_gls_bootstrap_dev": (_I, [_I, _VP, _VP, _VP, _VP, _L, _VP, _L, _D, _D, _L, _I, _I, _VP, _VP, _VP, _L]),
    "pdc_gls_plan_create": (_I, [_VP, _I, _L, _L, C.POINTER(_VP)]),
    "pdc_gls_plan_create_loopback": (_I, [_I, _I, _L, _L, C.POINTER(_VP)]),
    "pdc_gls_plan_info": (_I, [_VP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "pdc_gls_plan_init_error": (_I, [_VP, C.c_char_p, _I]),
    "pdc_gls_plan_upload": (_I, [_VP, _VP, _VP, _VP, _L]),
    "pdc_gls_plan_scan": (_I, [_VP, _D, _D, _L, _I, _I]),
    "pdc_gls_plan_wait": (_I, [_VP]),
    "pdc_gls_plan_download": (_I, [_VP, _VP, _L, _I]),
    "pdc_gls_plan_kernel_ms": (_I, [_VP, C.POINTER(C.c_float)]),
    "pdc_gls_plan_slot_ms": (_I, [_VP, C.POINTER(C.c_float), _I]),
    "pdc_gls_plan_destroy": (_I, [_VP]),
    "pdc_trig_sums": (_I, [_VP, _VP, _L, _D, _D, _L, _VP, _VP, _I]),
    "pdc_gls_work_bytes": (_L, [_L, _L, _L]),
    "pdc_gls_scan_dev": (_I, [_I, _VP, _VP, _VP, _VP, _VP, _L, _L, _I, _D, _D, _L, _L, _I, _I,
                              _VP, _VP, _VP, _VP, _L]),
    "pdc_gls_fft_work_bytes": (_L, [_L, _L]),
    "pdc_gls_scan_fft": (_I, [_VP, _VP, _VP, _L, _D, _D, _L, _I, _I, _VP, _I]),
    "pdc_gls_scan_fft_dev": (_I, [_I, _VP, _VP, _VP, _VP, _L, _D, _D, _L, _I, _I, _VP, _VP, _L]),
    "pdc_trig_sums_fft": (_I, [_VP, _VP, _L, _D, _L, _D, _VP, _VP, _I]),
    "pdc_gls_scan_fft_batch": (_I, [_VP, _VP, _VP, _VP, _L, _I, _D, _D, _L, _I, _I, _VP, _VP, _VP, _I]),
    "pdc_highest_peak": (_I, [_VP, _L, _L, _VP, _VP, _I]),
    "pdc_highest_peak_dev": (_I, [_I, _VP, _VP, _L, _L, _VP, _VP]),
    "pdc_gls_batch_highest_peak": (_I, [_VP, _VP, _VP, _VP, _L, _I, _D, _D, _L, _I, _I, _VP, _VP, _I]),
    "pdc_peaks_topk": (_I, [_VP, _L, _L, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I]),
    "pdc_peaks_topk_dev": (_I, [_I, _VP, _VP, _L, _L, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "pdc_gls_batch_peaks": (_I, [_VP, _VP, _VP, _VP, _L, _I, _D, _D, _L, _I, _I, _I, _I,
                                 _VP, _VP, _VP, _VP, _VP, _VP, _I]),
    "pdc_pdm_scan": (_I, [_VP, _VP, _L, _VP, _L, _I, _I, _D, _VP, _I]),
    "pdc_pdm_scan_dev": (_I, [_I, _VP, _VP, _VP, _L, _VP, _L, _I, _I, _D, _VP]),
    "pdc_aov_scan": (_I, [_VP, _VP, _L, _VP, _L, _I, _VP, _I]),
    "pdc_aov_scan_dev": (_I, [_I, _VP, _VP, _VP, _L, _VP, _L, _I, _VP]),
    "pdc_cond_entropy_scan": (_I, [_VP, _VP, _L, _VP, _L, _I, _I, _VP, _I]),
    "pdc_cond_entropy_scan_dev": (_I, [_I, _VP, _VP, _VP, _L, _VP, _L, _I, _I, _VP]),
    "pdc_stringlength_scan": (_I, [_VP, _VP, _L, _VP, _L, _VP, _I]),
    "pdc_stringlength_work_bytes": (_L, [_L, _L]),
    "pdc_stringlength_scan_dev": (_I, [_I, _VP, _VP, _VP, _L, _VP, _L, _VP, _VP, _L]),
    "pdc_gl_scan": (_I, [_VP, _L, _VP, _L, _I, _I, _VP, _I]),
    "pdc_gl_scan_dev": (_I, [_I, _VP, _VP, _L, _VP, _L, _I, _I, _VP]),
    "pdc_phase_work_bytes": (_L, [_I, _L, _L, _I, _I]),
    "pdc_phase_scan_dev": (_I, [_I, _I, _VP, _VP, _VP, _L, _VP, _L, _I, _I, _D, _VP, _VP, _L]),
    "pdc_pdm_scan_multi": (_I, [_VP, _VP, _L, _VP, _L, _I, _I, _D, _VP, _VP, _I]),
    "pdc_aov_scan_multi": (_I, [_VP, _VP, _L, _VP, _L, _I, _VP, _VP, _I]),
    "pdc_cond_entropy_scan_multi": (_I, [_VP, _VP, _L, _VP, _L, _I, _I, _VP, _VP, _I]),
    "pdc_gl_scan_multi": (_I, [_VP, _L, _VP, _L, _I, _I, _VP, _VP, _I]),
    "pdc_phase_plan_create": (_I, [_VP, _I, _L, _L, C.POINTER(_VP)]),
    "pdc_phase_plan_upload": (_I, [_VP, _VP, _VP, _L]),
    "pdc_phase_plan_scan": (_I, [_VP, _I, _VP, _L, _I, _I, _D]),
    "pdc_phase_plan_wait": (_I, [_VP]),
    "pdc_phase_plan_download": (_I, [_VP, _VP, _L]),
    "pdc_phase_plan_kernel_ms": (_I, [_VP, C.POINTER(C.c_float)]),
    "pdc_phase_plan_destroy": (_I, [_VP]),
    "pdc_stringlength_scan_multi": (_I, [_VP, _VP, _L, _VP, _L, _VP, _VP, _I]),
    "pdc_supersmoother_scan": (_I, [_VP, _VP, _L, _VP, _L, _D, _VP, _I]),
    "pdc_supersmoother_scan_multi": (_I, [_VP, _VP, _L, _VP, _L, _D, _VP, _VP, _I]),
    "pdc_supersmoother_work_bytes": (_L, [_L, _L]),
    "pdc_supersmoother_scan_dev": (_I, [_I, _VP, _VP, _VP, _L, _VP, _L, _D, _VP, _VP, _L]),
}


def library_path():
    """The in-tree HIP library; ``PDC_LIBRARY`` names another build of it (kernel A/B experiments)."""
    override = os.environ.get("PDC_LIBRARY")
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def lib():
    """The loaded shared library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                path = library_path()
                if not os.path.isfile(path):
                    raise RuntimeError(
                        f"{path} is missing: the HIP extension has not been built "
                        "(run `make -C periodicity_amd/csrc` or `python -c 'import "
                        "__graft_entry__ as g; g.build()'`). There is no CPU fallback.")
                handle = C.CDLL(path)
                for name, (res, args) in PROTOTYPES.items():
                    try:
                        fn = getattr(handle, name)
                    except AttributeError:
                        if os.environ.get("PDC_LIBRARY"):   # an older build under A/B comparison may lack newer entries
                            continue
                        raise
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def check(status):
    if status != 0:
        msg = lib().pdc_last_error().decode("utf-8", "replace")
        raise STATUS_EXC.get(status, RuntimeError)(f"libperiodicity_hip: {msg} (status {status})")


def device_count():
    n = C.c_int(0)
    status = lib().pdc_device_count(C.byref(n))
    return n.value if status == 0 else 0


def pick_device(device=None, devices=None):
    """The ONE precedence rule for the ``device=`` / ``devices=`` pair every scan accepts: a ``devices``
    list, when given, wins (its first entry for a single-device call); else ``device``; else None (the
    process default, ``default_device()``)."""
    if devices is not None and len(devices) > 0:
        return int(devices[0])
    return device


def device_info(device=0):
    name = C.create_string_buffer(256)
    cu, mem, clk = C.c_int(0), C.c_int64(0), C.c_int(0)
    check(lib().pdc_device_info(device, name, 256, C.byref(cu), C.byref(mem), C.byref(clk)))
    return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value,
            "clock_khz": clk.value}


def alloc_counts():
    """(device, pinned-host) allocations the library has made so far in this process."""
    dev, pin = C.c_int64(0), C.c_int64(0)
    check(lib().pdc_alloc_counts(C.byref(dev), C.byref(pin)))
    return dev.value, pin.value


def default_device():
    return int(os.environ.get("PERIODICITY_AMD_DEVICE", "0"))


def _f64(a, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 1:
        raise ValueError(f"{name} must be one-dimensional")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def grid_params(frequency):
    """(f0, delta, nf) of a grid built by ``np.arange`` — numpy fills ``start + i*delta`` with
    ``delta = a[1] - a[0]``, which is what the kernels recompute bit-for-bit."""
    frequency = np.asarray(frequency, dtype=np.float64)
    nf = frequency.size
    if nf == 0:
        return 0.0, 0.0, 0
    f0 = float(frequency[0])
    delta = float(frequency[1] - frequency[0]) if nf > 1 else 0.0
    return f0, delta, nf


# ---- host-buffer entry points ---------------------------------------------------------------------
def gls_scan(t, y, dy, f0, delta, nf, fit_mean=True, psd=False, j_begin=0, device=None):
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    if y.size != t.size or (dy is not None and dy.size != t.size):
        raise ValueError("Input arrays have incompatible lengths.")
    out = np.empty(nf, dtype=np.float64)
    dev = default_device() if device is None else device
    check(lib().pdc_gls_scan(_ptr(t), _ptr(y), _ptr(dy), t.size, f0, delta, j_begin, nf,
                             int(bool(fit_mean)), int(bool(psd)), _ptr(out), dev))
    return out


def bglst_scan(t, y, dy, f0, delta, nf, scalars, j_begin=0, device=None):
    """Log marginal likelihood per trial frequency of the harmonic + linear-trend model (``pdc_bglst_scan``);
    ``scalars``: the twelve frequency-independent inputs listed in include/periodicity_hip.h."""
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    if y.size != t.size or (dy is not None and dy.size != t.size):
        raise ValueError("Input arrays have incompatible lengths.")
    scalars = _f64(scalars, "scalars")
    if scalars.size != 12:
        raise ValueError("bglst_scan takes twelve scalars")
    out = np.empty(nf, dtype=np.float64)
    dev = default_device() if device is None else device
    check(lib().pdc_bglst_scan(_ptr(t), _ptr(y), _ptr(dy), t.size, f0, delta, j_begin, nf, _ptr(scalars), _ptr(out), dev))
    return out


def gls_scan_batch(t, y, dy, offsets, f0, delta, nf, fit_mean=True, psd=False, shared_t=False,
                   want_power=True, want_peaks=False, j_begin=0, device=None, devices=None):
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    nb = offsets.size - 1
    if nb < 1 or offsets[-1] != y.size or (dy is not None and dy.size != y.size):
        raise ValueError("Input arrays have incompatible lengths.")
    if (shared_t and t.size != offsets[1]) or (not shared_t and t.size != y.size):
        raise ValueError("Input arrays have incompatible lengths.")
    power = np.empty((nb, nf), dtype=np.float64) if want_power else None
    amax = np.empty(nb, dtype=np.float64) if want_peaks else None
    argmax = np.empty(nb, dtype=np.int64) if want_peaks else None
    if devices is not None and len(devices) > 1:
        # curves dealt to the device slots in contiguous groups, no exchange (pdc_gls_scan_batch_multi)
        if j_begin:
            raise ValueError("a batch sharded over devices scans the whole grid (j_begin must be 0)")
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        check(lib().pdc_gls_scan_batch_multi(_ptr(t), _ptr(y), _ptr(dy), _ptr(offsets), nb, int(shared_t),
                                             f0, delta, nf, int(bool(fit_mean)), int(bool(psd)),
                                             _ptr(power), _ptr(amax), _ptr(argmax), _ptr(devs), devs.size))
        return power, amax, argmax
    device = pick_device(device, devices)
    dev = default_device() if device is None else device
    check(lib().pdc_gls_scan_batch(_ptr(t), _ptr(y), _ptr(dy), _ptr(offsets), nb, int(shared_t),
                                   f0, delta, j_begin, nf, int(bool(fit_mean)), int(bool(psd)),
                                   _ptr(power), _ptr(amax), _ptr(argmax), dev))
    return power, amax, argmax


def gls_bootstrap(t, y, dy, picks, f0, delta, nf, fit_mean=True, psd=False, method="direct", device=None,
                  devices=None):
    """Maxima (and their bins) of the periodograms of the bootstrap replicates ``(y[picks[b]], dy[picks[b]])``
    on the unchanged time axis (``pdc_gls_bootstrap``): only the curve and the int32 indices are uploaded,
    the resampled arrays are never built.  ``method="fft"``: ``f0`` / ``delta`` are ``fmin`` / ``df``."""
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    picks = np.ascontiguousarray(picks, dtype=np.int32)
    if y.size != t.size or (dy is not None and dy.size != t.size):
        raise ValueError("Input arrays have incompatible lengths.")
    if picks.ndim != 2 or (picks.shape[0] and picks.shape[1] != t.size):
        raise ValueError("picks must be [n_bootstraps][n_samples]")
    nb = picks.shape[0]
    amax = np.empty(nb, dtype=np.float64)
    argmax = np.empty(nb, dtype=np.int64)
    if devices is not None and len(devices) > 0:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
    else:
        devs = np.array([default_device() if device is None else device], dtype=np.int32)
    check(lib().pdc_gls_bootstrap(_ptr(t), _ptr(y), _ptr(dy), t.size, _ptr(picks), nb, float(f0), float(delta),
                                  int(nf), int(bool(fit_mean)), int(bool(psd)), 1 if method == "fft" else 0,
                                  _ptr(amax), _ptr(argmax), _ptr(devs), devs.size))
    return amax, argmax


def gls_scan_multi(t, y, dy, f0, delta, nf, fit_mean=True, psd=False, devices=(0,)):
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    if y.size != t.size or (dy is not None and dy.size != t.size):
        raise ValueError("Input arrays have incompatible lengths.")
    devs = np.ascontiguousarray(devices, dtype=np.int32)
    out = np.empty(nf, dtype=np.float64)
    check(lib().pdc_gls_scan_multi(_ptr(t), _ptr(y), _ptr(dy), t.size, f0, delta, nf,
                                   int(bool(fit_mean)), int(bool(psd)), _ptr(out), _ptr(devs),
                                   devs.size))
    return out


class GlsPlan:
    """A persistent multi-GPU GLS plan (``pdc_gls_plan_*``): buffers, streams and RCCL communicators
    are created once; ``scan`` only enqueues (double-buffered), ``download`` waits and copies."""

    def __init__(self, devices, n_max, nf_max, loopback_slots=None):
        """``loopback_slots=k``: k logical slots on the ONE device listed, the all-gather replaced by the
        equivalent device-to-device copies (``pdc_gls_plan_create_loopback``) - how the N > 1 logic runs
        on a single GPU."""
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        self._plan = C.c_void_p()
        if loopback_slots is not None:
            if devs.size != 1:
                raise ValueError("a loopback plan lives on one device")
            self.devices = [int(devs[0])] * int(loopback_slots)
            check(lib().pdc_gls_plan_create_loopback(int(devs[0]), int(loopback_slots), int(n_max),
                                                     int(nf_max), C.byref(self._plan)))
        else:
            self.devices = [int(d) for d in devs]
            check(lib().pdc_gls_plan_create(_ptr(devs), devs.size, int(n_max), int(nf_max),
                                            C.byref(self._plan)))
        self.nf = 0

    def info(self):
        """``{"n_slots", "rccl_ranks", "exchange"}`` - the communicator size as RCCL reports it and which
        exchange the plan performs ("none", "rccl", "copy")."""
        n, r, x = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().pdc_gls_plan_info(self._plan, C.byref(n), C.byref(r), C.byref(x)))
        buf = C.create_string_buffer(512)
        check(lib().pdc_gls_plan_init_error(self._plan, buf, 512))
        return {"n_slots": n.value, "rccl_ranks": r.value, "exchange": ("none", "rccl", "copy")[x.value],
                "init_error": buf.value.decode() or None}

    def upload(self, t, y, dy=None):
        t, y = _f64(t, "t"), _f64(y, "y")
        dy = None if dy is None else _f64(dy, "dy")
        if y.size != t.size or (dy is not None and dy.size != t.size):
            raise ValueError("Input arrays have incompatible lengths.")
        check(lib().pdc_gls_plan_upload(self._plan, _ptr(t), _ptr(y), _ptr(dy), t.size))

    def scan(self, f0, delta, nf, fit_mean=True, psd=False):
        check(lib().pdc_gls_plan_scan(self._plan, float(f0), float(delta), int(nf),
                                      int(bool(fit_mean)), int(bool(psd))))
        self.nf = int(nf)

    def wait(self):
        check(lib().pdc_gls_plan_wait(self._plan))

    def download(self, which=0):
        out = np.empty(self.nf, dtype=np.float64)
        check(lib().pdc_gls_plan_download(self._plan, _ptr(out), self.nf, int(which)))
        return out

    def kernel_ms(self):
        ms = C.c_float()
        check(lib().pdc_gls_plan_kernel_ms(self._plan, C.byref(ms)))
        return ms.value

    def slot_ms(self):
        """HIP-event time of the latest slab scan on every slot."""
        ms = (C.c_float * len(self.devices))()
        check(lib().pdc_gls_plan_slot_ms(self._plan, ms, len(self.devices)))
        return [float(v) for v in ms]

    def close(self):
        if self._plan:
            check(lib().pdc_gls_plan_destroy(self._plan))
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def trig_sums(t, w, f0, delta, nf, device=None):
    t, w = _f64(t, "t"), _f64(w, "w")
    if w.size != t.size:
        raise ValueError("Input arrays have incompatible lengths.")
    S, Cc = np.empty(nf), np.empty(nf)
    dev = default_device() if device is None else device
    check(lib().pdc_trig_sums(_ptr(t), _ptr(w), t.size, f0, delta, nf, _ptr(S), _ptr(Cc), dev))
    return S, Cc


def gls_scan_fft(t, y, dy, fmin, df, nf, fit_mean=True, psd=False, device=None):
    """The reference's own FFT/extirpolation algorithm on the device (Tier F)."""
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    if y.size != t.size or (dy is not None and dy.size != t.size):
        raise ValueError("Input arrays have incompatible lengths.")
    out = np.empty(nf, dtype=np.float64)
    dev = default_device() if device is None else device
    check(lib().pdc_gls_scan_fft(_ptr(t), _ptr(y), _ptr(dy), t.size, float(fmin), float(df), nf,
                                 int(bool(fit_mean)), int(bool(psd)), _ptr(out), dev))
    return out


def gls_scan_fft_batch(t, y, dy, offsets, fmin, df, nf, fit_mean=True, psd=False, shared_t=False,
                       want_power=True, want_peaks=False, device=None):
    """Batch of curves through the FFT path (Tier F); same layout/outputs as ``gls_scan_batch``."""
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    nb = offsets.size - 1
    if nb < 1 or offsets[-1] != y.size or (dy is not None and dy.size != y.size):
        raise ValueError("Input arrays have incompatible lengths.")
    if (shared_t and t.size != offsets[1]) or (not shared_t and t.size != y.size):
        raise ValueError("Input arrays have incompatible lengths.")
    power = np.empty((nb, nf), dtype=np.float64) if want_power else None
    amax = np.empty(nb, dtype=np.float64) if want_peaks else None
    argmax = np.empty(nb, dtype=np.int64) if want_peaks else None
    dev = default_device() if device is None else device
    check(lib().pdc_gls_scan_fft_batch(_ptr(t), _ptr(y), _ptr(dy), _ptr(offsets), nb, int(shared_t),
                                       float(fmin), float(df), nf, int(bool(fit_mean)),
                                       int(bool(psd)), _ptr(power), _ptr(amax), _ptr(argmax), dev))
    return power, amax, argmax


def trig_sums_fft(t, h, df, nf, fmin, device=None):
    """``_trig_sum(t, h, df, nf, fmin)`` (spectral.py:11-40) on the device."""
    t, h = _f64(t, "t"), _f64(h, "h")
    if h.size != t.size:
        raise ValueError("Input arrays have incompatible lengths.")
    S, Cc = np.empty(nf), np.empty(nf)
    dev = default_device() if device is None else device
    check(lib().pdc_trig_sums_fft(_ptr(t), _ptr(h), t.size, float(df), nf, float(fmin), _ptr(S),
                                  _ptr(Cc), dev))
    return S, Cc


def highest_peak(power, device=None):
    """(index, value) of the highest ``find_peaks`` local maximum of each row of ``power``."""
    power = np.ascontiguousarray(power, dtype=np.float64)
    rows = power.reshape(1, -1) if power.ndim == 1 else power
    idx = np.empty(rows.shape[0], dtype=np.int64)
    val = np.empty(rows.shape[0], dtype=np.float64)
    dev = default_device() if device is None else device
    check(lib().pdc_highest_peak(_ptr(rows), rows.shape[0], rows.shape[1], _ptr(idx), _ptr(val), dev))
    return (int(idx[0]), float(val[0])) if power.ndim == 1 else (idx, val)


def gls_batch_highest_peak(t, y, dy, offsets, f0, delta, nf, fit_mean=True, psd=False,
                           shared_t=False, device=None):
    """Batched periodograms reduced on the device to each curve's highest peak (index, power)."""
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    nb = offsets.size - 1
    if nb < 1 or offsets[-1] != y.size or (dy is not None and dy.size != y.size):
        raise ValueError("Input arrays have incompatible lengths.")
    if (shared_t and t.size != offsets[1]) or (not shared_t and t.size != y.size):
        raise ValueError("Input arrays have incompatible lengths.")
    idx = np.empty(nb, dtype=np.int64)
    val = np.empty(nb, dtype=np.float64)
    dev = default_device() if device is None else device
    check(lib().pdc_gls_batch_highest_peak(_ptr(t), _ptr(y), _ptr(dy), _ptr(offsets), nb,
                                           int(shared_t), f0, delta, nf, int(bool(fit_mean)),
                                           int(bool(psd)), _ptr(idx), _ptr(val), dev))
    return idx, val


def _topk_outputs(nb, k):
    return {"count": np.empty(nb, dtype=np.int64), "indices": np.empty((nb, k), dtype=np.int64),
            "heights": np.empty((nb, k)), "prominences": np.empty((nb, k)),
            "half_lo": np.empty((nb, k), dtype=np.int64), "half_hi": np.empty((nb, k), dtype=np.int64)}


def peaks_topk(power, k=1, by_prominence=False, device=None):
    """The ``k`` (<= 1024; beyond 64 in launches of 64 ranks) highest, or most prominent, ``find_peaks`` maxima of each row of ``power`` with
    prominences and half-maximum crossings (``pdc_peaks_topk``); a dict of arrays shaped ``[rows, k]``
    (``count``: ``[rows]``), ranked descending, padded with -1 / NaN."""
    power = np.ascontiguousarray(power, dtype=np.float64)
    rows = power.reshape(1, -1) if power.ndim == 1 else power
    out = _topk_outputs(rows.shape[0], int(k))
    dev = default_device() if device is None else device
    check(lib().pdc_peaks_topk(_ptr(rows), rows.shape[0], rows.shape[1], int(k), int(bool(by_prominence)),
                               _ptr(out["count"]), _ptr(out["indices"]), _ptr(out["heights"]),
                               _ptr(out["prominences"]), _ptr(out["half_lo"]), _ptr(out["half_hi"]), dev))
    return out


def gls_batch_peaks(t, y, dy, offsets, f0, delta, nf, k=1, by_prominence=False, fit_mean=True,
                    psd=False, shared_t=False, device=None):
    """Batched periodograms reduced on the device to their ``k`` best peaks (see :func:`peaks_topk`)."""
    t, y = _f64(t, "t"), _f64(y, "y")
    dy = None if dy is None else _f64(dy, "dy")
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    nb = offsets.size - 1
    if nb < 1 or offsets[-1] != y.size or (dy is not None and dy.size != y.size):
        raise ValueError("Input arrays have incompatible lengths.")
    if (shared_t and t.size != offsets[1]) or (not shared_t and t.size != y.size):
        raise ValueError("Input arrays have incompatible lengths.")
    out = _topk_outputs(nb, int(k))
    dev = default_device() if device is None else device
    check(lib().pdc_gls_batch_peaks(_ptr(t), _ptr(y), _ptr(dy), _ptr(offsets), nb, int(shared_t), f0,
                                    delta, nf, int(bool(fit_mean)), int(bool(psd)), int(k),
                                    int(bool(by_prominence)), _ptr(out["count"]), _ptr(out["indices"]),
                                    _ptr(out["heights"]), _ptr(out["prominences"]), _ptr(out["half_lo"]),
                                    _ptr(out["half_hi"]), dev))
    return out


def pdm_scan(t, x, periods, nb, nc, sigma, device=None, devices=None):
    """theta at every trial period; ``devices`` (a sequence of GPU ordinals) cuts the period grid
    into one contiguous slab per entry (``pdc_pdm_scan_multi``)."""
    t, x, periods = _f64(t, "t"), _f64(x, "x"), _f64(periods, "periods")
    if x.size != t.size:
        raise ValueError("Input arrays have incompatible lengths.")
    out = np.empty(periods.size, dtype=np.float64)
    if devices is not None and len(devices) > 1:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        check(lib().pdc_pdm_scan_multi(_ptr(t), _ptr(x), t.size, _ptr(periods), periods.size, int(nb),
                                       int(nc), float(sigma), _ptr(out), _ptr(devs), devs.size))
        return out
    device = pick_device(device, devices)
    dev = default_device() if device is None else device
    check(lib().pdc_pdm_scan(_ptr(t), _ptr(x), t.size, _ptr(periods), periods.size, int(nb),
                             int(nc), float(sigma), _ptr(out), dev))
    return out


def aov_scan(t, x, periods, n_bins, device=None, devices=None):
    """Analysis-of-Variance statistic at every trial period (``pdc_aov_scan``; ``devices`` as in
    :func:`pdm_scan`)."""
    t, x, periods = _f64(t, "t"), _f64(x, "x"), _f64(periods, "periods")
    if x.size != t.size:
        raise ValueError("Input arrays have incompatible lengths.")
    out = np.empty(periods.size, dtype=np.float64)
    if devices is not None and len(devices) > 1:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        check(lib().pdc_aov_scan_multi(_ptr(t), _ptr(x), t.size, _ptr(periods), periods.size, int(n_bins),
                                       _ptr(out), _ptr(devs), devs.size))
        return out
    device = pick_device(device, devices)
    dev = default_device() if device is None else device
    check(lib().pdc_aov_scan(_ptr(t), _ptr(x), t.size, _ptr(periods), periods.size, int(n_bins),
                             _ptr(out), dev))
    return out


def cond_entropy_scan(t, mag_bin, periods, n_phase, n_mag, device=None, devices=None):
    """Conditional entropy at every trial period (``pdc_cond_entropy_scan``); ``mag_bin`` holds the
    magnitude bin (0 .. n_mag-1) of every sample; ``devices`` as in :func:`pdm_scan`."""
    t, mag_bin, periods = _f64(t, "t"), _f64(mag_bin, "mag_bin"), _f64(periods, "periods")
    if mag_bin.size != t.size:
        raise ValueError("Input arrays have incompatible lengths.")
    if mag_bin.size and not (np.all(mag_bin >= 0) and np.all(mag_bin < n_mag)):
        raise ValueError("magnitude bins must lie in 0 .. n_mag-1")
    out = np.empty(periods.size, dtype=np.float64)
    if devices is not None and len(devices) > 1:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        check(lib().pdc_cond_entropy_scan_multi(_ptr(t), _ptr(mag_bin), t.size, _ptr(periods), periods.size,
                                                int(n_phase), int(n_mag), _ptr(out), _ptr(devs), devs.size))
        return out
    device = pick_device(device, devices)
    dev = default_device() if device is None else device
    check(lib().pdc_cond_entropy_scan(_ptr(t), _ptr(mag_bin), t.size, _ptr(periods), periods.size,
                                      int(n_phase), int(n_mag), _ptr(out), dev))
    return out


def gl_scan(t, periods, m, n_offsets, device=None, devices=None):
    """Gregory-Loredo ``ln S_m`` at every trial period for the arrival times ``t`` (``pdc_gl_scan``): ``m``
    phase bins, the bin-offset integral as the mean over ``n_offsets`` shifts; ``devices`` as in
    :func:`pdm_scan`."""
    t, periods = _f64(t, "t"), _f64(periods, "periods")
    out = np.empty(periods.size, dtype=np.float64)
    if devices is not None and len(devices) > 1:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        check(lib().pdc_gl_scan_multi(_ptr(t), t.size, _ptr(periods), periods.size, int(m), int(n_offsets),
                                      _ptr(out), _ptr(devs), devs.size))
        return out
    device = pick_device(device, devices)
    dev = default_device() if device is None else device
    check(lib().pdc_gl_scan(_ptr(t), t.size, _ptr(periods), periods.size, int(m), int(n_offsets), _ptr(out), dev))
    return out


def stringlength_scan(t, m, periods, device=None, devices=None):
    """String length at every trial period; ``devices`` as in :func:`pdm_scan`
    (``pdc_stringlength_scan_multi``)."""
    t, m, periods = _f64(t, "t"), _f64(m, "m"), _f64(periods, "periods")
    if m.size != t.size:
        raise ValueError("Input arrays have incompatible lengths.")
    out = np.empty(periods.size, dtype=np.float64)
    if devices is not None and len(devices) > 1:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        check(lib().pdc_stringlength_scan_multi(_ptr(t), _ptr(m), t.size, _ptr(periods), periods.size,
                                                _ptr(out), _ptr(devs), devs.size))
        return out
    device = pick_device(device, devices)
    dev = default_device() if device is None else device
    check(lib().pdc_stringlength_scan(_ptr(t), _ptr(m), t.size, _ptr(periods), periods.size,
                                      _ptr(out), dev))
    return out


PHASE_KINDS = {"pdm": 0, "aov": 1, "cond_entropy": 2, "stringlength": 3, "gregory_loredo": 4, "supersmoother": 5}


def supersmoother_scan(t, y, periods, alpha=0.0, device=None, devices=None):
    """Mean absolute residual of the folded curve about its supersmoother fit at every trial period
    (``pdc_supersmoother_scan``; Friedman 1984 + Reimann 1994, a TODO upstream)."""
    t, y = _f64(t, "t"), _f64(y, "y")
    periods = _f64(periods, "periods")
    if y.size != t.size:
        raise ValueError("Input arrays have incompatible lengths.")
    out = np.empty(periods.size, dtype=np.float64)
    if devices is not None and len(devices) > 1:
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        check(lib().pdc_supersmoother_scan_multi(_ptr(t), _ptr(y), t.size, _ptr(periods), periods.size, float(alpha),
                                                 _ptr(out), _ptr(devs), devs.size))
        return out
    device = pick_device(device, devices)
    dev = default_device() if device is None else device
    check(lib().pdc_supersmoother_scan(_ptr(t), _ptr(y), t.size, _ptr(periods), periods.size, float(alpha), _ptr(out), dev))
    return out


class PhasePlan:
    """A persistent fan-out of the phase scans over device slots (``pdc_phase_plan_*``): streams, buffers
    and page-locked staging are created once; ``upload`` replicates the samples, ``scan`` only enqueues one
    slab of the period grid per slot, ``download`` waits and returns the whole grid."""

    def __init__(self, devices, n_max=0, n_periods_max=0):
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        self.devices = [int(d) for d in devs]
        self._plan = C.c_void_p()
        check(lib().pdc_phase_plan_create(_ptr(devs), devs.size, int(n_max), int(n_periods_max),
                                          C.byref(self._plan)))
        self.n_periods = 0

    def upload(self, t, v):
        t, v = _f64(t, "t"), _f64(v, "v")
        if v.size != t.size:
            raise ValueError("Input arrays have incompatible lengths.")
        check(lib().pdc_phase_plan_upload(self._plan, _ptr(t), _ptr(v), t.size))

    def scan(self, kind, periods, nb=1, nc=1, sigma=1.0):
        periods = _f64(periods, "periods")
        check(lib().pdc_phase_plan_scan(self._plan, PHASE_KINDS[kind], _ptr(periods), periods.size, int(nb),
                                        int(nc), float(sigma)))
        self.n_periods = periods.size

    def wait(self):
        check(lib().pdc_phase_plan_wait(self._plan))

    def download(self):
        out = np.empty(self.n_periods, dtype=np.float64)
        check(lib().pdc_phase_plan_download(self._plan, _ptr(out), self.n_periods))
        return out

    def kernel_ms(self):
        ms = C.c_float()
        check(lib().pdc_phase_plan_kernel_ms(self._plan, C.byref(ms)))
        return ms.value

    def close(self):
        if self._plan:
            check(lib().pdc_phase_plan_destroy(self._plan))
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- device-resident helpers (bench.py, tests) ----------------------------------------------------
class DeviceBuffer:
    """A raw HBM allocation owned by the caller."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(lib().pdc_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_array(cls, a, device=0):
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes, device)
        check(lib().pdc_memcpy_h2d(device, buf.ptr, _ptr(a), a.nbytes))
        return buf

    def to_array(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        check(lib().pdc_memcpy_d2h(self.device, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(lib().pdc_free(self.device, self.ptr))
            self.ptr = None
