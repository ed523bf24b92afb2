"""Generalized Lomb-Scargle periodogram with the reference's callable API, computed on MI355X.

Drop-in for ``periodicity.spectral`` (``/root/reference/src/periodicity/spectral.py``):
``GLS(fmin, fmax, n, psd)(signal, err, fit_mean) -> FSeries`` with the same positional order,
defaults and attribute side effects (``.frequency .err .signal .periodogram``,
``spectral.py:97,101,133-134``), plus ``bootstrap / fap / fal / window / model / copy``
(``spectral.py:137-204``).  ``LombScargle`` is an alias of ``GLS``.

What differs, deliberately: the three ``_trig_sum`` calls (``spectral.py:109-112``) are an
FFT/extirpolation *approximation* upstream; here they are the exact direct sums that function's
docstring defines (``spectral.py:13-15``), evaluated by the HIP kernel in
``csrc/gls.hip``.  The frequency grid, weights and epilogue follow the reference line by line.
Nothing in this module computes a periodogram on the CPU.
"""
import copy as _copy

import numpy as np

from . import _cabi
from .core import FSeries, TSeries

__all__ = ["GLS", "LombScargle", "BGLST"]


def _as_tseries(signal):
    """``spectral.py:86-87``: anything that is not a time series is wrapped as values on
    ``arange`` times.  Real ``periodicity.core.TSeries`` objects pass through (duck typing)."""
    if isinstance(signal, TSeries) or (hasattr(signal, "time") and hasattr(signal, "values")
                                       and hasattr(signal, "baseline")):
        return signal
    return TSeries(values=signal)


class GLS(object):
    """Generalized Lomb-Scargle periodogram (Zechmeister & Kurster 2009) of a discrete signal.

    Parameters (``spectral.py:53-72``)
    ----------
    fmin, fmax: float, optional
        Grid limits; default half a cycle per baseline and the pseudo-Nyquist ``0.5/median_dt``.
    n: float, optional
        Samples per peak (default 5): grid spacing is ``1 / baseline / n``.
    psd: bool, optional
        Leave the periodogram un-normalised.
    method: {"direct", "fft"}, keyword-only, optional
        ``"direct"`` (default): exact direct summation of the trig sums (``spectral.py:13-15``).
        ``"fft"``: the reference's own Press-Rybicki extirpolation + FFT (``spectral.py:18-39``)
        on the device — reproduces upstream's values including their approximation error.
    device: int, keyword-only, optional
        GPU ordinal (default ``$PERIODICITY_AMD_DEVICE`` or 0).
    devices: sequence of int, keyword-only, optional
        Shard the frequency grid over these GPUs of one node (RCCL all-gather of the power array).
    """

    def __init__(self, fmin=None, fmax=None, n=5, psd=False, *, method="direct", device=None,
                 devices=None):
        if method not in ("direct", "fft"):
            raise ValueError("method must be 'direct' or 'fft'")
        self.fmin = fmin
        self.fmax = fmax
        self.n = n
        self.psd = psd
        self.method = method
        self.device = device
        self.devices = None if devices is None else tuple(devices)

    def _grid_scalars(self, signal):
        """``df``, ``fmin``, ``fmax`` of ``spectral.py:88-96``."""
        df = 1.0 / signal.baseline / self.n
        fmin = 0.5 * df if self.fmin is None else self.fmin
        fmax = 0.5 / signal.median_dt if self.fmax is None else self.fmax
        return df, fmin, fmax

    def _grid(self, signal):
        """Uniform frequency grid of ``spectral.py:88-98`` — built by ``np.arange`` itself so its
        length and values carry numpy's rounding; the kernel reproduces ``start + j*step``."""
        df, fmin, fmax = self._grid_scalars(signal)
        return np.arange(fmin, fmax + df, df)

    def __call__(self, signal, err=None, fit_mean=True):
        """Periodogram of ``signal`` on the default (or configured) uniform grid.

        Parameters (``spectral.py:74-85``)
        ----------
        err: array-like, optional
            Measurement uncertainties for each sample (may be heteroscedastic).
        fit_mean: bool, optional
            Let the mean float with the fit (the "generalized" part).
        """
        signal = _as_tseries(signal)
        self.frequency = self._grid(signal)
        f0, delta, nf = _cabi.grid_params(self.frequency)
        have_err = err is not None
        if not have_err:
            err = np.ones_like(signal.values)
        self.err = err
        dy = np.asarray(err, dtype=float) if have_err else None
        t = np.asarray(signal.time, dtype=float)
        y = np.asarray(signal.values, dtype=float)
        if self.method == "fft":
            # the reference's own extirpolation + FFT evaluation of the trig sums, on the device
            df, fmin, _ = self._grid_scalars(signal)
            dev = _cabi.pick_device(self.device, self.devices)
            power = _cabi.gls_scan_fft(t, y, dy, fmin, df, nf, fit_mean, self.psd, device=dev)
        elif self.devices is not None and len(self.devices) > 1:
            power = _cabi.gls_scan_multi(t, y, dy, f0, delta, nf, fit_mean, self.psd,
                                         self.devices)
        else:
            dev = _cabi.pick_device(self.device, self.devices)
            power = _cabi.gls_scan(t, y, dy, f0, delta, nf, fit_mean, self.psd, device=dev)
        self.signal = signal
        self.periodogram = FSeries(self.frequency, power)
        return self.periodogram

    def copy(self):
        return _copy.deepcopy(self)

    def bootstrap(self, n_bootstraps, random_seed=None):
        """Maxima of ``n_bootstraps`` periodograms of ``(values, err)`` resampled with replacement
        on the unchanged time axis (``spectral.py:140-152``).  The draws come from
        ``default_rng(seed).integers(0, n, n)`` once per replicate, in order, exactly as
        upstream; the replicates then run as ONE batched launch that shares the time axis and
        returns only the NaN-aware maximum of each spectrum (with ``devices=(...)``: one contiguous
        group of replicates per GPU, no exchange).  Only the curve and the 4-byte indices go to the device
        (``pdc_gls_bootstrap``)."""
        rng = np.random.default_rng(random_seed)
        ndata = len(self.signal)
        values = np.asarray(self.signal.values, dtype=float)
        err = np.asarray(self.err, dtype=float)
        t = np.asarray(self.signal.time, dtype=float)
        # the draws, exactly as upstream makes them (one ``integers(0, n, n)`` call per replicate, in order) -
        # kept as 4-byte indices: the resampled (values, err) arrays are never built on the host, the device
        # prologue gathers ``values[picks]``, ``err[picks]`` while it lays out the weight table
        picks = np.empty((n_bootstraps, ndata), dtype=np.int32)
        for i in range(n_bootstraps):
            picks[i] = rng.integers(0, ndata, ndata)
        bs_replicates = np.empty(n_bootstraps)
        # err=None upstream means all-ones errors (``spectral.py:99-100``): resampling leaves them
        # all ones, and the equal-weights kernels share the weight-only sums between replicates
        dy = None if np.all(err == 1.0) else err
        if n_bootstraps and self.method == "fft":
            # all replicates through the reference's own algorithm in one batched set of launches
            df, fmin, _ = self._grid_scalars(self.signal)
            nf = self._grid(self.signal).size
            bs_replicates[:] = _cabi.gls_bootstrap(t, values, dy, picks, fmin, df, nf, True, self.psd, method="fft",
                                                   device=_cabi.pick_device(self.device, self.devices))[0]
        elif n_bootstraps:
            f0, delta, nf = _cabi.grid_params(self._grid(self.signal))
            bs_replicates[:] = _cabi.gls_bootstrap(t, values, dy, picks, f0, delta, nf, True, self.psd,
                                                   device=self.device, devices=self.devices)[0]
        self.bs_replicates = bs_replicates
        return self.bs_replicates

    def fap(self, power):
        """Fraction of bootstrap maxima above ``power`` (``spectral.py:154-160``)."""
        return np.mean(power < self.bs_replicates)

    def fal(self, fap):
        """Power level at false-alarm probability ``fap`` (``spectral.py:162-163``)."""
        return np.quantile(self.bs_replicates, 1 - fap)

    def window(self):
        """Spectral window: periodogram of an all-ones signal, no floating mean
        (``spectral.py:165-167``)."""
        gls = self.copy()
        return gls(0.0 * self.signal + 1.0, fit_mean=False)

    def model(self, tf, f0):
        """Weighted least-squares sinusoid-plus-offset at frequency ``f0`` evaluated at times
        ``tf`` (``spectral.py:169-204``).  O(N) host arithmetic, not part of the scan."""
        t = np.asarray(self.signal.time, dtype=float)
        sigma = np.asarray(self.err, dtype=float)
        w = sigma ** -2.0
        y = np.asarray(self.signal.values, dtype=float)
        y_mean = np.dot(y, w) / w.sum()

        def basis(times):
            arg = 2 * np.pi * f0 * np.asarray(times, dtype=float)
            return np.vstack([np.ones_like(arg), np.sin(arg), np.cos(arg)])

        A = basis(t) / sigma
        theta = np.linalg.solve(A @ A.T, A @ ((y - y_mean) / sigma))
        return TSeries(tf, y_mean + basis(tf).T @ theta)


LombScargle = GLS


class BGLST(GLS):
    """Bayesian generalised Lomb-Scargle periodogram with linear trend.

    The reference exports this name (``spectral.py:7``) for an empty class (``spectral.py:207-208``; its README lists
    the method as "soon"): there is no upstream behaviour to reproduce - **parity unpinned by the reference**.  This
    class computes the published statistic (Olspert, Pelt, Käpylä & Lehtinen 2018, A&A 615, A111) on the grid rule of
    ``GLS`` (``spectral.py:88-98``): for every trial frequency the log marginal likelihood of

        ``y_i = A cos(2 pi f t_i) + B sin(2 pi f t_i) + alpha tau_i + beta + eps_i``,  ``eps_i ~ N(0, err_i**2)``,

    ``tau = (t - t_ref) / baseline``, with independent zero-mean Gaussian priors ``A, B ~ N(0, sigma_A**2)``,
    ``alpha ~ N(0, sigma_alpha**2)`` (trend over the whole baseline), ``beta ~ N(0, sigma_beta**2)`` (level at
    ``t_ref``) integrated out analytically.  Unlike ``GLS`` the values are NOT centred or detrended first - the
    trend is part of the model and competes with long periods on equal terms, which is the point of the method.

    Parameters
    ----------
    fmin, fmax, n: as ``GLS``.
    sigma_A, sigma_alpha, sigma_beta: float, keyword-only, optional
        Prior standard deviations.  Defaults (this build's; the reference has none): ``std(values)`` for the
        amplitudes and for the trend over the baseline, ``sqrt(var(values) + mean(values)**2)`` for the level.
    t_ref: float, keyword-only, optional
        Time at which ``beta`` is the level (default: the middle of the series).
    device: int, keyword-only, optional
    """

    def __init__(self, fmin=None, fmax=None, n=5, *, sigma_A=None, sigma_alpha=None, sigma_beta=None, t_ref=None,
                 device=None):
        super().__init__(fmin, fmax, n, False, device=device)
        self.sigma_A, self.sigma_alpha, self.sigma_beta, self.t_ref = sigma_A, sigma_alpha, sigma_beta, t_ref

    def priors(self, signal):
        """``(sigma_A, sigma_alpha, sigma_beta, t_ref)`` with the defaults filled in for ``signal``."""
        signal = _as_tseries(signal)
        y = np.asarray(signal.values, dtype=float)
        t = np.asarray(signal.time, dtype=float)
        spread = float(np.std(y))
        spread = spread if spread > 0 else 1.0
        return (float(self.sigma_A) if self.sigma_A is not None else spread,
                float(self.sigma_alpha) if self.sigma_alpha is not None else spread,
                float(self.sigma_beta) if self.sigma_beta is not None else float(np.sqrt(np.var(y) + np.mean(y) ** 2)) or 1.0,
                float(self.t_ref) if self.t_ref is not None else 0.5 * (t[0] + t[-1]))

    @staticmethod
    def _scalars(t, y, err, sigma_A, sigma_alpha, sigma_beta, t_ref):
        """The twelve frequency-independent inputs of ``pdc_bglst_scan`` (include/periodicity_hip.h)."""
        span = float(t[-1] - t[0]) or 1.0
        w = err ** -2.0
        W = float(w.sum())
        w = w / W
        tau = (t - t_ref) / span
        return np.array([W, np.dot(w, y * y), np.dot(w, y), np.dot(w, tau * y), np.dot(w, tau * tau), np.dot(w, tau),
                         (t[0] - t_ref) / span, 1.0 / span, sigma_A ** -2.0, sigma_alpha ** -2.0, sigma_beta ** -2.0,
                         float(np.sum(np.log(2 * np.pi * err ** 2))) + 4 * np.log(sigma_A) + 2 * np.log(sigma_alpha)
                         + 2 * np.log(sigma_beta)])

    def __call__(self, signal, err=None):
        """``FSeries(frequency, log marginal likelihood)``; ``period_at_highest_peak`` etc. as for ``GLS``."""
        signal = _as_tseries(signal)
        if len(signal) < 4:
            raise ValueError("BGLST marginalises four parameters: at least four samples")
        self.frequency = self._grid(signal)
        f0, delta, nf = _cabi.grid_params(self.frequency)
        t = np.asarray(signal.time, dtype=float)
        y = np.asarray(signal.values, dtype=float)
        have_err = err is not None
        err = np.ones_like(y) if not have_err else np.asarray(err, dtype=float)
        if err.size != y.size:
            raise ValueError("Input arrays have incompatible lengths.")
        self.err = err
        sA, sa, sb, t_ref = self.priors(signal)
        scalars = self._scalars(t, y, err, sA, sa, sb, t_ref)
        dev = _cabi.pick_device(self.device, None)
        ll = _cabi.bglst_scan(t, y, err if have_err else None, f0, delta, nf, scalars, device=dev)
        self.signal = signal
        self.periodogram = FSeries(self.frequency, ll)
        return self.periodogram

    def posterior_mean(self, frequency):
        """Posterior means ``(A, B, alpha, beta)`` of the model's parameters at one frequency (``alpha`` per unit
        time, ``beta`` at ``t_ref``) for the last signal: a 4 x 4 solve on the host."""
        t = np.asarray(self.signal.time, dtype=float)
        y = np.asarray(self.signal.values, dtype=float)
        sA, sa, sb, t_ref = self.priors(self.signal)
        span = float(t[-1] - t[0]) or 1.0
        phi = np.stack([np.cos(2 * np.pi * frequency * t), np.sin(2 * np.pi * frequency * t), (t - t_ref) / span,
                        np.ones_like(t)], axis=1)
        w = np.asarray(self.err, dtype=float) ** -2.0
        m = phi.T @ (phi * w[:, None]) + np.diag([sA ** -2.0, sA ** -2.0, sa ** -2.0, sb ** -2.0])
        a, b, alpha, beta = np.linalg.solve(m, phi.T @ (w * y))
        return a, b, alpha / span, beta

    # the GLS-only methods make no sense for a likelihood
    def bootstrap(self, *args, **kwargs):
        raise NotImplementedError("BGLST has no bootstrap: the log-likelihood itself carries the significance")

    fap = fal = window = model = bootstrap
