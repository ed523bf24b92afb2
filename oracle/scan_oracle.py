"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the reference's trial-frequency scans.

This is the *oracle*: the checker the HIP path is compared with.  It is never the product
path; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.

Every function names the reference lines it restates (paths relative to
``/root/reference/src/periodicity/``).  Pinning: ``tests/test_oracle_vs_reference.py`` runs
each function against the reference's own source (loaded by ``oracle/refstub.py``) in the
build container, and ``tests/test_oracle_golden.py`` checks it against the vectors under
``tests/golden/`` (generated from the reference by ``tests/golden/make_golden.py``),
including the two known-answer tests of ``/root/reference/tests/test_spectral.py``.

Two flavours of the Lomb-Scargle trig sums exist because the reference itself is an
approximation:

* ``trig_sum_fft``   — the Press-Rybicki extirpolation + FFT the reference executes
                       (``spectral.py:11-40``); reproduces ``power_ref`` ("Tier R").
* ``trig_sum_exact`` — the sums that function's docstring *defines* (``spectral.py:13-15``)
                       evaluated directly in 80-bit ``np.longdouble`` ("Tier E", the
                       <=1e-6 gate for the direct-sum HIP kernel).
"""
import numpy as np

# 2*pi to the full 64-bit mantissa (np.pi is only a double)
TWO_PI_L = np.longdouble(2) * np.arctan2(np.longdouble(0), np.longdouble(-1))


# ------------------------------------------------------------------------------------------
# Grid rules
# ------------------------------------------------------------------------------------------
def gls_grid(time, n=5, fmin=None, fmax=None):
    """Frequency grid of ``GLS.__call__`` (``spectral.py:88-98``).

    Returns ``(frequency, df, fmin)``; ``frequency`` is produced by ``np.arange`` itself so
    its length and every value carry numpy's own rounding.
    """
    time = np.asarray(time)
    baseline = time[-1] - time[0]
    df = 1.0 / baseline / n
    if fmin is None:
        fmin = 0.5 * df
    if fmax is None:
        fmax = 0.5 / np.median(np.diff(time))
    return np.arange(fmin, fmax + df, df), df, fmin


def stringlength_periods(baseline, dphi=0.1, n_periods=1000):
    """Trial periods of ``StringLength.__call__`` (``phase.py:67-68``): uniform in frequency."""
    df = dphi / baseline
    return 1 / np.linspace(n_periods * df, df, n_periods)


def stringlength_scale(values):
    """Scaling to [-0.25, +0.25] stated at ``phase.py:65-66`` (NaN-aware max/min,
    ``core.py:202-240``)."""
    vmax, vmin = np.nanmax(values), np.nanmin(values)
    return (values - vmax) / (2 * (vmax - vmin)) + 0.25


def pdm_periods(time, p_min=None, p_max=None, n_periods=1000, oversample=1):
    """Trial periods of ``PDM.__call__`` (``phase.py:167-180``): uniform in period."""
    time = np.asarray(time)
    t0 = time[-1] - time[0]
    if p_min is None:
        p_min = 2 * np.median(np.diff(time))
    if p_max is None:
        p_max = oversample * t0
    if n_periods is None:
        n_periods = int((1 / p_min - 1 / p_max) * oversample * t0 + 1)
    return np.linspace(p_min, p_max, n_periods), p_min, p_max


# ------------------------------------------------------------------------------------------
# Lomb-Scargle trig sums
# ------------------------------------------------------------------------------------------
def trig_sum_fft(t, h, df, nf, fmin, oversampling=5):
    """``S_j, C_j = sum_i h_i {sin,cos}(2 pi (fmin + j df) t_i)`` by extirpolation + inverse FFT.

    Restates ``_trig_sum`` (``spectral.py:11-40``): grid length ``2**ceil(log2(5 nf))`` (:18),
    weights pre-rotated to ``fmin`` about ``tmin`` (:19-20), sample positions in grid units
    (:21), exact hits deposited whole (:23-24), all others spread over four neighbours with
    cubic Lagrange weights (:26-33), ``ifft`` truncated to ``nf`` (:34), the ``tmin`` shift
    undone (:35-37) and scaled by the grid length (:38-39).
    """
    t = np.asarray(t, dtype=float)
    nfft = 1 << int(nf * oversampling - 1).bit_length()
    tmin = t.min()
    hc = h * np.exp(2j * np.pi * fmin * (t - tmin))
    pos = ((t - tmin) * nfft * df) % nfft
    grid = np.zeros(nfft, dtype=hc.dtype)
    whole = pos % 1 == 0
    np.add.at(grid, pos[whole].astype(int), hc[whole])
    pos, hc = pos[~whole], hc[~whole]
    lo = np.clip((pos - 2).astype(int), 0, nfft - 4)
    x = pos - lo                                   # in (0, 4): offset from the first node
    full = hc * (x * (x - 1) * (x - 2) * (x - 3))   # prod over all four nodes
    # Lagrange basis l_m(x) = full / ((x - m) * prod_{k != m}(m - k)); the constants are
    # -6, 2, -2, 6 for m = 0..3 (the reference walks them from m = 3 down: 6, -2, 2, -6).
    for m, const in ((3, 6.0), (2, -2.0), (1, 2.0), (0, -6.0)):
        np.add.at(grid, lo + m, full / (const * (pos - (lo + m))))
    spec = np.fft.ifft(grid)[:nf]
    if tmin != 0:
        # in place on the truncated view, as upstream: numpy's complex multiply rounds the last
        # bit differently for a fresh output array (SIMD body vs. peeled elements)
        spec *= np.exp(2j * np.pi * tmin * (fmin + df * np.arange(nf)))
    return nfft * spec.imag, nfft * spec.real


def trig_sum_exact(t, h, frequency, chunk=1 << 14):
    """The sums ``spectral.py:13-15`` defines, evaluated directly in 80-bit long double at the
    given grid frequencies.  O(N nf): for fixtures and small parity cases only."""
    tl = np.asarray(t, dtype=np.longdouble)
    hl = np.asarray(h, dtype=np.longdouble)
    fl = np.asarray(frequency, dtype=np.longdouble)
    S = np.empty(fl.size, dtype=np.longdouble)
    C = np.empty(fl.size, dtype=np.longdouble)
    step = max(1, chunk * 64 // max(1, tl.size))
    for a in range(0, fl.size, step):
        ph = TWO_PI_L * np.outer(fl[a:a + step], tl)
        S[a:a + step] = np.sin(ph) @ hl
        C[a:a + step] = np.cos(ph) @ hl
    return S.astype(float), C.astype(float)


def gls_weights(values, err=None, fit_mean=True):
    """Prologue of ``GLS.__call__`` (``spectral.py:99-108``): ``w``, centred ``y``, ``err``."""
    values = np.asarray(values)
    if err is None:
        err = np.ones_like(values)
    w = err ** -2.0
    w /= w.sum()
    y = values - np.dot(w, values) if fit_mean else values
    return w, y, err


def gls_epilogue(Sh, Ch, S2, C2, S, C, YY, fit_mean=True, psd=False, err=None):
    """Elementwise epilogue of ``GLS.__call__`` (``spectral.py:111-132``)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        if fit_mean:
            tan2 = (S2 - 2 * S * C) / (C2 - (C * C - S * S))
        else:
            tan2 = S2 / C2
        norm = np.sqrt(1 + tan2 * tan2)
        S2w = tan2 / norm
        C2w = 1 / norm
        Cw = np.sqrt(0.5) * np.sqrt(1 + C2w)
        Sw = np.sqrt(0.5) * np.sign(S2w) * np.sqrt(1 - C2w)
        YC = Ch * Cw + Sh * Sw
        YS = Sh * Cw - Ch * Sw
        CC = 0.5 * (1 + C2 * C2w + S2 * S2w)
        SS = 0.5 * (1 - C2 * C2w - S2 * S2w)
        if fit_mean:
            CC -= (C * Cw + S * Sw) ** 2
            SS -= (S * Cw - C * Sw) ** 2
        power = YC * YC / CC + YS * YS / SS
        if psd:
            power *= 0.5 * (err ** -2.0).sum()
        else:
            power /= YY
    return power


def gls_power(t, values, err, frequency, df, fmin, fit_mean=True, psd=False, sums="fft"):
    """Periodogram of ``GLS.__call__`` (``spectral.py:99-132``) on a given grid.

    ``sums="fft"`` follows the reference to the letter (three ``_trig_sum`` calls at
    ``spectral.py:109-112``); ``sums="exact"`` swaps in the long-double direct sums.
    """
    t = np.asarray(t)
    nf = frequency.size
    w, y, err = gls_weights(values, err, fit_mean)
    if sums == "fft":
        Sh, Ch = trig_sum_fft(t, w * y, df, nf, fmin)
        S2, C2 = trig_sum_fft(t, w, 2 * df, nf, 2 * fmin)
        S, C = trig_sum_fft(t, w, df, nf, fmin) if fit_mean else (None, None)
    elif sums == "exact":
        Sh, Ch = trig_sum_exact(t, w * y, frequency)
        S2, C2 = trig_sum_exact(t, w, 2 * np.asarray(frequency, dtype=np.longdouble))
        S, C = trig_sum_exact(t, w, frequency) if fit_mean else (None, None)
    else:
        raise ValueError(sums)
    YY = np.dot(w, y ** 2)
    return gls_epilogue(Sh, Ch, S2, C2, S, C, YY, fit_mean, psd, err)


def gls(time, values, err=None, fit_mean=True, n=5, fmin=None, fmax=None, psd=False,
        sums="fft"):
    """``GLS(fmin, fmax, n, psd)(TSeries(time, values), err, fit_mean)`` → (frequency, power)."""
    frequency, df, f0 = gls_grid(time, n, fmin, fmax)
    return frequency, gls_power(time, values, err, frequency, df, f0, fit_mean, psd, sums)


def gls_bootstrap_maxima(time, values, err, n_bootstraps, random_seed=None, sums="fft", **grid):
    """Replicate maxima of ``GLS.bootstrap`` (``spectral.py:140-152``): resample ``(y, err)``
    jointly with replacement on the unchanged time axis; keep the NaN-aware maximum."""
    rng = np.random.default_rng(random_seed)
    values = np.asarray(values)
    err = np.ones_like(values) if err is None else np.asarray(err)
    frequency, df, f0 = gls_grid(time, grid.get("n", 5), grid.get("fmin"), grid.get("fmax"))
    out = np.empty(n_bootstraps)
    for i in range(n_bootstraps):
        pick = rng.integers(0, values.size, values.size)
        out[i] = np.nanmax(gls_power(time, values[pick], err[pick], frequency, df, f0,
                                     True, grid.get("psd", False), sums))
    return out


# ------------------------------------------------------------------------------------------
# Phase-folding scans
# ------------------------------------------------------------------------------------------
def pdm_theta(t, x, period, nb=5, nc=2, sigma=None):
    """Stellingwerf's theta for one trial period (``PDM._pdm``, ``phase.py:128-149``)."""
    if sigma is None:
        sigma = np.var(x, ddof=1)
    m0 = nb * nc
    phi = (t / period) % 1
    order = np.argsort(phi)                        # phase.py:132-134 (numerically inert)
    phi, xs = phi[order], x[order]
    num, n_sum, good = 0.0, 0, 0
    var_terms, sizes = [], []
    for k in range(m0):
        member = (phi >= k / m0) & (phi < (k + nc) / m0)
        member |= phi < (k - (m0 - nc)) / m0        # wrap-around cover
        xk = xs[member]
        if xk.size > 1:
            var_terms.append(np.var(xk, ddof=1))
            sizes.append(xk.size)
            good += 1
    sj, nj = np.array(var_terms), np.array(sizes)
    return (np.sum((nj - 1) * sj) / (np.sum(nj) - good)) / sigma


def pdm_scan(t, x, periods, nb=5, nc=2):
    """theta for every trial period, in the order of ``periods`` (``phase.py:185-187``)."""
    sigma = np.var(x, ddof=1)
    return np.array([pdm_theta(t, x, p, nb, nc, sigma) for p in periods])


def pdm_subharmonic(thetas, periods, n_samples, p_min, p_max):
    """Sub-harmonic averaging of ``PDM.__call__`` (``phase.py:166,181,188-193``)."""
    thetas = np.array(thetas, dtype=float)
    theta_crit = 1.0 - 11.0 / n_samples ** 0.8
    dp = periods[1] - periods[0]
    (ok,) = np.where((thetas < theta_crit) & (periods <= p_max / 2))
    sub = np.round(2 * ok + p_min / dp).astype(int)
    thetas[ok] = (thetas[ok] + thetas[sub]) / 2
    return thetas


def pdm(time, values, nb=5, nc=2, p_min=None, p_max=None, n_periods=1000, oversample=1,
        do_subharmonic=False):
    """``PDM(...)(TSeries(time, values))`` → (frequency ascending, theta) (``phase.py:151-195``)."""
    time, values = np.asarray(time), np.asarray(values)
    periods, p_lo, p_hi = pdm_periods(time, p_min, p_max, n_periods, oversample)
    thetas = pdm_scan(time, values, periods, nb, nc)
    if do_subharmonic:
        thetas = pdm_subharmonic(thetas, periods, values.size, p_lo, p_hi)
    freq = 1 / periods
    order = np.argsort(freq, kind="stable")        # FSeries sorts ascending, core.py:877-881
    return freq[order], thetas[order]


# The two scans below are listed as TODO in the reference (``phase.py:11-15``) and have no upstream
# implementation: PARITY UNPINNED by the reference.  They restate the published formulas; the phase
# bins are the PDM fine bins of ``phase.py:137`` ([k/r, (k+1)/r) against the doubles k/r, phi from the
# same ``(t / period) % 1``), phi == 1.0 joins the last bin, NaN phases belong to no bin.
def _phase_bins(t, period, r):
    phi = (t / period) % 1
    edges = np.arange(r + 1) / r
    k = np.searchsorted(edges, phi, side="right") - 1
    k = np.where(np.isnan(phi), -1, np.minimum(k, r - 1))
    return k


def aov_theta(t, x, period, n_bins=10):
    """Analysis-of-Variance statistic, Schwarzenberg-Czerny 1989 (MNRAS 241, 153), eq. 1-3:
    ``s1^2 / s2^2`` with ``s1^2 = sum n_i (xbar_i - xbar)^2 / (r - 1)`` and
    ``s2^2 = sum_i sum_j (x_ij - xbar_i)^2 / (n - r)`` over ``r`` phase bins."""
    k = _phase_bins(t, period, n_bins)
    ok = k >= 0
    n = int(ok.sum())
    if n <= n_bins or n_bins < 2:
        return np.nan
    xbar = x[ok].mean()
    between = within = 0.0
    for i in range(n_bins):
        xi = x[k == i]
        if xi.size:
            between += xi.size * (xi.mean() - xbar) ** 2
            within += np.sum((xi - xi.mean()) ** 2)
    return (between / (n_bins - 1)) / (within / (n - n_bins))


def aov_scan(t, x, periods, n_bins=10):
    return np.array([aov_theta(t, x, p, n_bins) for p in periods])


def magnitude_bins(values, n_mag=5):
    """Graham et al. 2013: magnitudes normalised to the unit interval, ``n_mag`` equal bins (the
    maximum joins the last one); the bin index of every sample as float64 (the C ABI's input)."""
    v = np.asarray(values, dtype=float)
    lo, hi = np.nanmin(v), np.nanmax(v)
    unit = (v - lo) / (hi - lo)
    return np.minimum(np.floor(unit * n_mag), n_mag - 1).astype(float)


def cond_entropy(t, mag_bin, period, n_phase=10, n_mag=5):
    """Conditional entropy, Graham et al. 2013 (MNRAS 434, 2629), eq. 1:
    ``H_c = sum_ij p(m_j, phi_i) ln(p(phi_i) / p(m_j, phi_i))`` over the occupied cells."""
    k = _phase_bins(t, period, n_phase)
    ok = k >= 0
    n = int(ok.sum())
    if n == 0:
        return np.nan
    cells = np.zeros((n_phase, n_mag))
    np.add.at(cells, (k[ok], mag_bin[ok].astype(int)), 1)
    rows = cells.sum(axis=1, keepdims=True) * np.ones_like(cells)
    live = cells > 0
    return float(np.sum(cells[live] / n * np.log(rows[live] / cells[live])))


def cond_entropy_scan(t, mag_bin, periods, n_phase=10, n_mag=5):
    return np.array([cond_entropy(t, mag_bin, p, n_phase, n_mag) for p in periods])


def gl_log_s(t, period, m, n_offsets=8):
    """Gregory & Loredo (1992, ApJ 398, 146), eq. 5.13-5.14 marginalised over the bin offset: the log of
    ``<m^N / W_m(w, phi)>_phi`` with ``W_m = N! / (n_1! ... n_m!)`` the multiplicity of the counts of the
    arrival times ``t`` in ``m`` phase bins, the offset average over ``n_offsets`` shifts of the bin
    boundaries by ``1 / (m n_offsets)`` of a cycle.  Phases and fine-bin edges as ``PDM._pdm`` computes and
    compares them (``phase.py:131,137``).  The reference lists the method as TODO (``phase.py:13``): parity
    is to this restatement of the paper."""
    from scipy.special import gammaln
    fine = m * n_offsets
    k = _phase_bins(t, period, fine)
    counts = np.bincount(k[k >= 0], minlength=fine).astype(np.int64)
    n = int(counts.sum())
    if n == 0:
        return np.nan
    log_terms = np.empty(n_offsets)
    for off in range(n_offsets):
        per_bin = np.roll(counts, -off).reshape(m, n_offsets).sum(axis=1)
        log_terms[off] = n * np.log(m) + gammaln(per_bin + 1.0).sum() - gammaln(n + 1.0)
    top = log_terms.max()
    return top + np.log(np.exp(log_terms - top).sum() / n_offsets)


def gl_scan(t, periods, m, n_offsets=8):
    return np.array([gl_log_s(t, p, m, n_offsets) for p in periods])


def stringlength_one(t, m, period):
    """Dworetsky string length for one trial period: ``StringLength._stringlength``
    (``phase.py:45-51``) through ``TSeries.fold`` (``core.py:543-544``) and the stable
    sort-by-phase of the ``TSeries`` constructor (``core.py:473-477``).  The polygon is closed
    with ``np.roll`` and the closing segment is *not* phase-wrapped."""
    phi = ((t - 0) / period) % 1
    order = np.argsort(phi, kind="stable")
    phi, mm = phi[order], m[order]
    return np.hypot(np.roll(mm, -1) - mm, np.roll(phi, -1) - phi).sum()


def stringlength_scan(t, m, periods):
    return np.array([stringlength_one(t, m, p) for p in periods])


def stringlength(time, values, dphi=0.1, n_periods=1000):
    """Intended behaviour of ``StringLength(dphi, n_periods)(signal)`` (``phase.py:53-72``;
    broken at HEAD — SURVEY.md fact 4) → (frequency ascending, ell)."""
    time, values = np.asarray(time), np.asarray(values)
    m = stringlength_scale(values)
    periods = stringlength_periods(time[-1] - time[0], dphi, n_periods)
    ell = stringlength_scan(time, m, periods)
    freq = 1 / periods
    order = np.argsort(freq, kind="stable")
    return freq[order], ell[order]


# ---- Supersmoother period search (the TODO at spectral.py:8: "check out Supersmoother (Reimann 1994)") -----------
# The reference has no implementation: what follows restates the PUBLISHED algorithm - Friedman's variable span
# smoother (J. H. Friedman 1984, "A variable span smoother", SLAC PUB-3477 / LCS TR 5; the Fortran `supsmu` /
# `smooth` distributed with it and shipped as R's stats::supsmu, periodic = TRUE) applied to the phase-folded
# curve, with Reimann's statistic (J. D. Reimann 1994, PhD thesis, UC Berkeley): the absolute residuals about
# the smooth.  PARITY UNPINNED BY THE REFERENCE.
SS_SPANS = (0.05, 0.2, 0.5)          # tweeter, midrange, woofer
SS_BIG, SS_SML, SS_EPS = 1.0e20, 1.0e-7, 1.0e-3


def ss_half_width(n, span):
    """Points on either side of the centre of a running-lines window (`smooth`: ibw)."""
    ibw = int(0.5 * span * n + 0.5)
    return max(ibw, 2)


def ss_smooth_incremental(x, y, span, vsmlsq, cv):
    """Literal restatement of Friedman's `smooth` for periodic x in [0, 1) (jper = 2), unit weights: the running
    window of 2 ibw + 1 points is slid with the updating formulas, one point out, one point in; pure Python loops,
    small n only - it exists to pin the vectorised form below."""
    n = len(x)
    ibw = ss_half_width(n, span)
    it = min(2 * ibw + 1, n)
    xm = ym = var = cvar = fbw = 0.0
    for i in range(1, it + 1):
        j = i - ibw - 1
        xti = x[j - 1] if j >= 1 else x[n + j - 1] - 1.0
        yj = y[j - 1] if j >= 1 else y[n + j - 1]
        fbo = fbw
        fbw = fbw + 1.0
        xm = (fbo * xm + xti) / fbw
        ym = (fbo * ym + yj) / fbw
        tmp = fbw * (xti - xm) / fbo if fbo > 0.0 else 0.0
        var += tmp * (xti - xm)
        cvar += tmp * (yj - ym)
    smo = np.empty(n)
    acvr = np.zeros(n)
    for j in range(1, n + 1):
        out, inn = j - ibw - 1, j + ibw
        if out < 1:
            out = n + out
            xto, xti = x[out - 1] - 1.0, x[inn - 1]
        elif inn > n:
            inn = inn - n
            xti, xto = x[inn - 1] + 1.0, x[out - 1]
        else:
            xto, xti = x[out - 1], x[inn - 1]
        fbo = fbw
        fbw = fbw - 1.0
        tmp = fbo * (xto - xm) / fbw if fbw > 0.0 else 0.0
        var -= tmp * (xto - xm)
        cvar -= tmp * (y[out - 1] - ym)
        if fbw > 0.0:
            xm = (fbo * xm - xto) / fbw
            ym = (fbo * ym - y[out - 1]) / fbw
        fbo = fbw
        fbw = fbw + 1.0
        xm = (fbo * xm + xti) / fbw
        ym = (fbo * ym + y[inn - 1]) / fbw
        tmp = fbw * (xti - xm) / fbo if fbo > 0.0 else 0.0
        var += tmp * (xti - xm)
        cvar += tmp * (y[inn - 1] - ym)
        a = cvar / var if var > vsmlsq else 0.0
        smo[j - 1] = a * (x[j - 1] - xm) + ym
        if cv:
            h = 1.0 / fbw
            if var > vsmlsq:
                h += (x[j - 1] - xm) ** 2 / var
            a1 = 1.0 - h
            if a1 > 0.0:
                acvr[j - 1] = abs(y[j - 1] - smo[j - 1]) / a1
            elif j > 1:
                acvr[j - 1] = acvr[j - 2]
    return ss_average_ties(x, smo), acvr


def ss_average_ties(x, smo):
    """`smooth`, label 90-110: fitted values of equal abscissae are replaced by their mean."""
    smo = np.array(smo, dtype=float)
    n = len(x)
    j = 0
    while j < n:
        j0 = j
        while j + 1 < n and x[j + 1] <= x[j]:
            j += 1
        if j > j0:
            smo[j0:j + 1] = smo[j0:j + 1].sum() / (j - j0 + 1)
        j += 1
    return smo


def ss_smooth(x, y, span, vsmlsq, cv):
    """The same smoother from window sums: window of sample j = the 2 ibw + 1 points j - ibw .. j + ibw (periodic:
    points taken from the other end count with abscissae - 1 / + 1), mean line through them evaluated at x[j], and -
    `cv` - the absolute leave-one-out residual |y - smo| / (1 - 1/fbw - (x - xm)^2 / var).

    Long-double prefix sums over the n points AS THEY STAND, abscissae relative to the median (a running-lines fit
    does not change under a shift of x); a wrapped window adds its far part's own sums U and point count c through
    sum (x -+ 1) = U_x -+ c, sum (x -+ 1)^2 = U_xx -+ 2 U_x + c, sum (x -+ 1) z = U_xz -+ U_z.  (Round 4 prefixed
    over concat(x - 1, x, x + 1): every prefix value then carried ~n terms of order 1, and for phases crowded into a
    sliver of the cycle - a period far beyond the baseline - even 80-bit sums lost var = Sxx - fbw xm^2: the ORACLE
    was wrong in the 8th digit at a spread of 1e-5, found when the device kernels stopped agreeing with it there.)"""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    n = x.size
    ibw = ss_half_width(n, span)
    if 2 * ibw + 1 > n:
        raise ValueError("supersmoother: too few samples for the span")
    L = np.longdouble
    xc = (x - x[n // 2]).astype(L)
    yl = y.astype(L)
    j = np.arange(n)
    lo, hi = j - ibw, j + ibw + 1
    low, high = lo < 0, hi > n
    wl = np.where(low, n + lo, 0)
    wh = np.where(low, n, np.where(high, hi - n, 0))
    sh = np.where(low, L(-1), np.where(high, L(1), L(0)))
    cw = (wh - wl).astype(L)
    lo_c, hi_c = np.maximum(lo, 0), np.minimum(hi, n)
    def sums(v):
        c = np.concatenate([[L(0)], np.cumsum(v)])
        return c[hi_c] - c[lo_c], c[wh] - c[wl]
    (mx, ux), (mxx, uxx), (my, uy), (mxy, uxy) = sums(xc), sums(xc * xc), sums(yl), sums(xc * yl)
    fbw = L(2 * ibw + 1)
    sx, sxx, sy, sxy = mx + (ux + sh * cw), mxx + (uxx + 2 * sh * ux + cw), my + uy, mxy + (uxy + sh * uy)
    xm, ym = sx / fbw, sy / fbw
    var = sxx - fbw * xm * xm
    cvar = sxy - fbw * xm * ym
    a = np.where(var > vsmlsq, cvar / np.where(var > vsmlsq, var, 1), 0)
    smo = a * (xc - xm) + ym
    acvr = np.zeros(n)
    if cv:
        h = 1 / fbw + np.where(var > vsmlsq, (xc - xm) ** 2 / np.where(var > vsmlsq, var, 1), 0)
        a1 = 1 - h
        ok = a1 > 0
        acvr = np.where(ok, np.abs(yl - smo) / np.where(ok, a1, 1), 0).astype(float)
        for jj in np.nonzero(~ok)[0]:                    # (`smooth`, label 70: carried over from the point before)
            if jj > 0:
                acvr[jj] = acvr[jj - 1]
    return ss_average_ties(x, smo.astype(float)), acvr


def supersmoother(x, y, alpha=0.0, smooth=ss_smooth):
    """Friedman's `supsmu` for periodic x in [0, 1), sorted ascending, unit weights, automatic span: three
    running-lines smooths (tweeter / midrange / woofer) with cross-validated residuals; the residuals smoothed
    with the midrange span; per point the span with the smallest smoothed residual (pulled towards the woofer by
    the bass control `alpha` in (0, 10]); those spans smoothed (midrange); the two neighbouring smooths
    interpolated at the smoothed span; the result smoothed with the tweeter span."""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    n = x.size
    i, j = n // 4, 3 * (n // 4)                     # (1-based in the Fortran: x(i), x(j))
    scale = x[j - 1] - x[max(i, 1) - 1]
    while scale <= 0.0:                             # (`supsmu`, label 30: widen until the abscissae differ)
        if j < n:
            j += 1
        if i > 1:
            i -= 1
        if j >= n and i <= 1:
            break
        scale = x[j - 1] - x[i - 1]
    vsmlsq = (SS_EPS * scale) ** 2
    sm = np.empty((3, n))
    res = np.empty((3, n))
    for k, span in enumerate(SS_SPANS):
        sm[k], acvr = smooth(x, y, span, vsmlsq, True)
        res[k], _ = smooth(x, acvr, SS_SPANS[1], vsmlsq, False)
    best = np.argmin(res, axis=0)                   # (first minimum: `if (sc(j,2i) < resmin)`)
    resmin = res[best, np.arange(n)]
    span_j = np.asarray(SS_SPANS)[best]
    if 0.0 < alpha <= 10.0:
        # (R's stats::supsmu guards with `resmin .lt. sc(j,6) .and. resmin .gt. 0`; Friedman's original has no
        # positivity test: a smoothed residual <= 0 - an exactly fitted run - leaves the span alone here, as in R)
        pull = (resmin < res[2]) & (resmin > 0)
        ratio = np.maximum(SS_SML, resmin / np.where(pull, res[2], 1.0))
        span_j = np.where(pull, span_j + (SS_SPANS[2] - span_j) * ratio ** (10.0 - alpha), span_j)
    span_s, _ = smooth(x, span_j, SS_SPANS[1], vsmlsq, False)
    span_s = np.clip(span_s, SS_SPANS[0], SS_SPANS[2])
    f = span_s - SS_SPANS[1]
    up = f >= 0.0
    fu = f / (SS_SPANS[2] - SS_SPANS[1])
    fd = -f / (SS_SPANS[1] - SS_SPANS[0])
    mixed = np.where(up, (1.0 - fu) * sm[1] + fu * sm[2], (1.0 - fd) * sm[1] + fd * sm[0])
    out, _ = smooth(x, mixed, SS_SPANS[0], vsmlsq, False)
    return out


def supersmoother_stat(t, y, period, alpha=0.0):
    """Mean absolute residual of the folded curve about its supersmoother fit (Reimann 1994): the fold and the
    stable sort by phase are the reference's (core.py:543-544, 473-477), as for StringLength."""
    phi = (np.asarray(t, dtype=float) - 0.0) / period % 1
    order = np.argsort(phi, kind="stable")
    xs, ys = phi[order], np.asarray(y, dtype=float)[order]
    return float(np.mean(np.abs(ys - supersmoother(xs, ys, alpha))))


def supersmoother_scan(t, y, periods, alpha=0.0):
    return np.array([supersmoother_stat(t, y, p, alpha) for p in np.asarray(periods, dtype=float)])


# ---- BGLST (spectral.py:7,207-208: the name is exported, the class body is `pass`) -------------------------------------
# PARITY UNPINNED BY THE REFERENCE.  Restates the published statistic - Olspert, Pelt, Kapyla & Lehtinen 2018, A&A 615,
# A111, "Bayesian generalised Lomb-Scargle periodogram with trend": data y_i = A cos(2 pi f t_i) + B sin(2 pi f t_i) +
# alpha tau_i + beta + eps_i, eps_i ~ N(0, err_i^2), independent zero-mean Gaussian priors on the four linear
# parameters, which are integrated out - by the two textbook routes that must agree: the dense one (y ~ N(0, Phi Sigma
# Phi^T + N), scipy's multivariate normal: the third-party pin, small n only) and the 4 x 4 one (Woodbury + the matrix
# determinant lemma) in 80-bit arithmetic for whole grids.  tau = (t - t_ref) / (t[-1] - t[0]).
def bglst_design(t, frequency, t_ref, trig_origin=None):
    """Columns (cos, sin, tau, 1).  The trigonometric pair is evaluated on ``t - trig_origin`` (default: the first
    sample): with one prior width for A and B the marginal likelihood does not change under a rotation of that pair,
    i.e. under a shift of its time origin (tests/test_bglst.py checks exactly this), and Julian-date stamps times a
    high frequency would otherwise cost the 80-bit phase its last digits."""
    t = np.asarray(t, dtype=np.longdouble)
    span = t[-1] - t[0] if t[-1] != t[0] else np.longdouble(1)
    origin = t[0] if trig_origin is None else np.longdouble(trig_origin)
    arg = TWO_PI_L * np.longdouble(frequency) * (t - origin)
    return np.stack([np.cos(arg), np.sin(arg), (t - np.longdouble(t_ref)) / span, np.ones_like(t)], axis=1)


def bglst_loglik_dense(t, y, err, frequency, sigma_A, sigma_alpha, sigma_beta, t_ref):
    """log N(y; 0, Phi Sigma Phi^T + diag err^2) for ONE frequency, by scipy (O(n^3): the pin, not the oracle)."""
    from scipy.stats import multivariate_normal
    phi = bglst_design(t, frequency, t_ref).astype(float)
    cov = phi @ np.diag([sigma_A ** 2, sigma_A ** 2, sigma_alpha ** 2, sigma_beta ** 2]) @ phi.T + np.diag(np.asarray(err, float) ** 2)
    return float(multivariate_normal(mean=np.zeros(len(y)), cov=cov, allow_singular=False).logpdf(np.asarray(y, float)))


def bglst_loglik(t, y, err, frequency, sigma_A, sigma_alpha, sigma_beta, t_ref):
    """log p(y | f) for every frequency of ``frequency``: -1/2 [y^T N^-1 y - b^T M^-1 b + log|M| + log|Sigma| +
    sum log(2 pi err^2)], M = Phi^T N^-1 Phi + Sigma^-1, b = Phi^T N^-1 y, in long double with an explicit 4 x 4
    Cholesky (numpy's linalg has no 80-bit routines)."""
    L = np.longdouble
    y = np.asarray(y, dtype=L)
    err = np.ones_like(y) if err is None else np.asarray(err, dtype=L)
    w = err ** -2
    prec = np.array([L(sigma_A) ** -2, L(sigma_A) ** -2, L(sigma_alpha) ** -2, L(sigma_beta) ** -2])
    const = np.sum(np.log(TWO_PI_L * err ** 2)) + 4 * np.log(L(sigma_A)) + 2 * np.log(L(sigma_alpha)) + 2 * np.log(L(sigma_beta))
    yy = np.dot(w, y * y)
    out = np.empty(len(frequency))
    for j, f in enumerate(np.asarray(frequency, dtype=float)):
        phi = bglst_design(t, f, t_ref)
        m = phi.T @ (phi * w[:, None]) + np.diag(prec)
        b = phi.T @ (w * y)
        c = np.zeros((4, 4), dtype=L)                     # Cholesky, lower
        for i in range(4):
            for k in range(i + 1):
                s = m[i, k] - np.dot(c[i, :k], c[k, :k])
                c[i, k] = np.sqrt(s) if i == k else s / c[k, k]
        z = np.zeros(4, dtype=L)
        for i in range(4):
            z[i] = (b[i] - np.dot(c[i, :i], z[:i])) / c[i, i]
        out[j] = float(-0.5 * (yy - np.dot(z, z) + 2 * np.sum(np.log(np.diag(c))) + const))
    return out
