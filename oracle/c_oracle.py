"""TEST INFRASTRUCTURE — ctypes access to oracle/_build/liboracle.so (``scan_oracle.c``).

Same role and rules as ``scan_oracle.py``: a checker for tests / smoke / the CPU-baseline leg of
bench.py, never the product path.  Multi-threaded with OpenMP (``threads`` = cores used).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    """``ORACLE_LIBRARY`` names another build of the same source (the sanitized one, `make -C oracle SAN=1`)."""
    global _lib
    if _lib is None:
        path = os.environ.get("ORACLE_LIBRARY") or _PATH
        if not os.path.isfile(path):
            build()
        _lib = C.CDLL(path)
    return _lib


def set_threads(n):
    os.environ["OMP_NUM_THREADS"] = str(int(n))
    try:
        omp = C.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(n))
    except OSError:
        pass


_tuned = None


def tune_threads(candidates=None):
    """Pick the OpenMP thread count that gives the double-precision direct sums the highest throughput HERE and set
    it (returns it).  On the shared GPU boxes of the pool `os.cpu_count()` says 256 while a lease gets a fraction of
    the machine: 32 threads ran the checker at 10 Gpair/s, 256 at 4 (profiles/r06_oracle_rate.txt).  Timed once per
    process on a fixed 2e8-pair sample per candidate."""
    global _tuned
    if _tuned is not None:
        return _tuned
    import time
    cpus = os.cpu_count() or 1
    cands = candidates or sorted({max(1, cpus // 8), max(1, cpus // 4), max(1, cpus // 2), cpus})
    rng = np.random.default_rng(0)
    n, nf = 20_000, 10_000
    t = np.sort(rng.uniform(0, n, n))
    h = rng.uniform(0.5, 1.5, n)
    f = 0.5 / n / 5 + np.arange(nf) / n / 5
    best = None
    for th in cands:
        set_threads(th)
        gls_sums_f64(t, h, h, f[:256])                  # (threads up)
        t0 = time.perf_counter()
        gls_sums_f64(t, h, h, f)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, th)
    _tuned = best[1]
    set_threads(_tuned)
    return _tuned


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def trig_sums_exact(t, h, frequency):
    t, h, f = _f(t), _f(h), _f(frequency)
    S, Cc = np.empty(f.size), np.empty(f.size)
    lib().oracle_trig_sums_exact(_p(t), _p(h), C.c_int64(t.size), _p(f), C.c_int64(f.size),
                                 _p(S), _p(Cc))
    return S, Cc


def pdm_scan(t, x, periods, nb=5, nc=2, sigma=None):
    t, x, p = _f(t), _f(x), _f(periods)
    if sigma is None:
        sigma = np.var(x, ddof=1)
    out = np.empty(p.size)
    lib().oracle_pdm_scan(_p(t), _p(x), C.c_int64(t.size), _p(p), C.c_int64(p.size),
                          C.c_int(nb), C.c_int(nc), C.c_double(sigma), _p(out))
    return out


def stringlength_scan(t, m, periods):
    t, m, p = _f(t), _f(m), _f(periods)
    out = np.empty(p.size)
    lib().oracle_stringlength_scan(_p(t), _p(m), C.c_int64(t.size), _p(p), C.c_int64(p.size),
                                   _p(out))
    return out


def gls_power_exact(t, values, err, frequency, fit_mean=True, psd=False):
    """Reference prologue/epilogue (numpy restatement) around the C long-double sums."""
    from . import scan_oracle as so
    w, y, err = so.gls_weights(values, err, fit_mean)
    f = _f(frequency)
    Sh, Ch = trig_sums_exact(t, w * y, f)
    # doubling a double is exact, so 2*f is the grid the 2-omega sums are defined on
    S2, C2 = trig_sums_exact(t, w, 2 * f)
    S, Cc = trig_sums_exact(t, w, f) if fit_mean else (None, None)
    return so.gls_epilogue(Sh, Ch, S2, C2, S, Cc, np.dot(w, y ** 2), fit_mean, psd, err)


def gls_sums_f64(t, hy, h, frequency):
    """Six direct sums per frequency in double precision (``oracle_gls_sums_f64``): ``(Sh, Ch)`` of ``hy``,
    ``(S, C)`` of ``h`` at ``frequency`` and ``(S2, C2)`` of ``h`` at twice it.  The fast checker for whole
    BASELINE-size grids; pinned to the long-double sums in tests/test_oracle_golden.py."""
    t, hy, h, f = _f(t), _f(hy), _f(h), _f(frequency)
    out = [np.empty(f.size) for _ in range(6)]
    lib().oracle_gls_sums_f64(_p(t), _p(hy), _p(h), C.c_int64(t.size), _p(f), C.c_int64(f.size),
                              *[_p(o) for o in out])
    return out


def gls_power_f64_batch(t, values, err, frequency, fit_mean=True, psd=False):
    """Rows of ``gls_power_f64`` for curves of equal length (2-D ``t``, ``values``, ``err``): one OpenMP region
    over (curve, frequency tile), the reference's prologue / epilogue per row."""
    from . import scan_oracle as so
    t = np.ascontiguousarray(t, dtype=np.float64)
    B, n = t.shape
    w = np.empty((B, n))
    y = np.empty((B, n))
    for b in range(B):
        w[b], y[b], _ = so.gls_weights(values[b], err[b], fit_mean)
    hy = w * y
    f = _f(frequency)
    out = [np.empty((B, f.size)) for _ in range(6)]
    offsets = np.arange(B + 1, dtype=np.int64) * n
    lib().oracle_gls_sums_f64_batch(_p(t), _p(hy), _p(w), _p(offsets), C.c_int64(B), _p(f), C.c_int64(f.size),
                                    *[_p(o) for o in out])
    Sh, Ch, S, Cc, S2, C2 = out
    power = np.empty((B, f.size))
    for b in range(B):
        power[b] = so.gls_epilogue(Sh[b], Ch[b], S2[b], C2[b], S[b] if fit_mean else None, Cc[b] if fit_mean else None,
                                   np.dot(w[b], y[b] ** 2), fit_mean, psd, err[b])
    return power


def gls_power_f64(t, values, err, frequency, fit_mean=True, psd=False):
    """Reference prologue / epilogue (numpy restatement, spectral.py:99-132) around the double-precision direct
    sums: what ``gls_power_exact`` computes, ~100x faster, for exhaustive checks of 1e6-bin grids."""
    from . import scan_oracle as so
    w, y, err = so.gls_weights(values, err, fit_mean)
    Sh, Ch, S, Cc, S2, C2 = gls_sums_f64(t, w * y, w, frequency)
    if not fit_mean:
        S = Cc = None
    return so.gls_epilogue(Sh, Ch, S2, C2, S, Cc, np.dot(w, y ** 2), fit_mean, psd, err)


def aov_scan(t, x, periods, n_bins=10):
    t, x, p = _f(t), _f(x), _f(periods)
    out = np.empty(p.size)
    lib().oracle_aov_scan(_p(t), _p(x), C.c_int64(t.size), _p(p), C.c_int64(p.size), C.c_int(n_bins), _p(out))
    return out


def cond_entropy_scan(t, mag_bin, periods, n_phase=10, n_mag=5):
    t, g, p = _f(t), _f(mag_bin), _f(periods)
    out = np.empty(p.size)
    lib().oracle_cond_entropy_scan(_p(t), _p(g), C.c_int64(t.size), _p(p), C.c_int64(p.size), C.c_int(n_phase),
                                   C.c_int(n_mag), _p(out))
    return out


def gl_scan(t, periods, m, n_offsets=8):
    t, p = _f(t), _f(periods)
    out = np.empty(p.size)
    lib().oracle_gl_scan(_p(t), C.c_int64(t.size), _p(p), C.c_int64(p.size), C.c_int(m), C.c_int(n_offsets), _p(out))
    return out


def bglst_loglik_f64(t, y, err, frequency, sigma_A, sigma_alpha, sigma_beta, t_ref):
    """``scan_oracle.bglst_loglik`` for whole BASELINE-size grids: the per-frequency sums from the double-precision
    direct-sum checker (two calls of ``gls_sums_f64``: weights ``w y`` then ``w tau``; ``sum w cos^2`` and
    ``sum w cos sin`` from its doubled-frequency sums), then the same 4 x 4 marginalisation, vectorised over the
    frequencies in float64.  Trigonometric origin: the first sample, as the numpy oracle's default (the likelihood is
    invariant under it).  PARITY UNPINNED BY THE REFERENCE (spectral.py:207-208 is an empty class)."""
    t = _f(t)
    y = _f(y)
    err = np.ones_like(y) if err is None else _f(err)
    f = _f(frequency)
    span = t[-1] - t[0] if t[-1] != t[0] else 1.0
    tau = (t - t_ref) / span
    w = err ** -2.0
    W = w.sum()
    w = w / W
    tt = t - t[0]
    Yc_s = gls_sums_f64(tt, w * y, w, f)          # Sh, Ch of w y;  S, C of w;  S2, C2 of w at 2 f
    Ys, Yc, S, Cc, S2, C2 = Yc_s
    Ts, Tc = gls_sums_f64(tt, w * tau, w, f)[:2]
    wsum = w.sum()
    cc, ss, cs = 0.5 * (wsum + C2), 0.5 * (wsum - C2), 0.5 * S2
    prec = np.array([sigma_A ** -2.0, sigma_A ** -2.0, sigma_alpha ** -2.0, sigma_beta ** -2.0])
    m = np.empty((f.size, 4, 4))
    m[:, 0, 0], m[:, 1, 1] = W * cc + prec[0], W * ss + prec[1]
    m[:, 2, 2], m[:, 3, 3] = W * np.dot(w, tau * tau) + prec[2], W * wsum + prec[3]
    m[:, 1, 0] = m[:, 0, 1] = W * cs
    m[:, 2, 0] = m[:, 0, 2] = W * Tc
    m[:, 2, 1] = m[:, 1, 2] = W * Ts
    m[:, 3, 0] = m[:, 0, 3] = W * Cc
    m[:, 3, 1] = m[:, 1, 3] = W * S
    m[:, 3, 2] = m[:, 2, 3] = W * np.dot(w, tau)
    b = np.stack([W * Yc, W * Ys, np.full(f.size, W * np.dot(w, tau * y)), np.full(f.size, W * np.dot(w, y))], axis=1)
    chol = np.linalg.cholesky(m)
    z = np.linalg.solve(chol, b[:, :, None])[:, :, 0]
    logdet = 2.0 * np.log(np.diagonal(chol, axis1=1, axis2=2)).sum(axis=1)
    const = np.sum(np.log(2 * np.pi * err ** 2)) + 4 * np.log(sigma_A) + 2 * np.log(sigma_alpha) + 2 * np.log(sigma_beta)
    return -0.5 * (W * np.dot(w, y * y) - np.sum(z * z, axis=1) + logdet + const)
