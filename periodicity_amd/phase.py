"""Phase-folding period scans with the reference's callable API, computed on MI355X.

Drop-in for ``periodicity.phase`` (``/root/reference/src/periodicity/phase.py``):

* ``StringLength(dphi, n_periods, cores)(signal) -> FSeries``   (``phase.py:18-72``)
* ``PDM(nb, nc, p_min, p_max, n_periods, oversample, do_subharmonic, cores)(signal) -> FSeries``
  (``phase.py:75-195``)

Positional order, defaults and the attributes each call leaves behind (``.signal .m
.periodogram`` / ``.signal .t .x .sigma .periods .periodogram``) are the reference's.  Upstream maps
one Python task per trial period over a ``multiprocessing.Pool`` (``phase.py:69-70,185-186``);
here a whole period grid is ONE kernel launch (``csrc/stringlength.hip``, ``csrc/pdm.hip``), so
``cores`` is kept only for signature compatibility.  No scan is ever computed on the CPU.
"""
from multiprocessing import cpu_count

import numpy as np

from . import _cabi
from .core import FSeries, TSeries

MAX_CORES = cpu_count()

__all__ = ["StringLength", "PDM", "AOV", "ConditionalEntropy", "GregoryLoredo", "SuperSmoother"]


# ---- host-side grid / scaling rules (O(N) or O(n_periods) numpy, as upstream) -----------------------
def _coerce(signal):
    """Raw array-likes become ``TSeries(values=...)`` (``phase.py:62-63,160-161``); anything that
    already looks like a time series (ours or the reference's xarray-backed one) passes through."""
    looks_like_series = all(hasattr(signal, name) for name in ("time", "values", "baseline"))
    return signal if isinstance(signal, TSeries) or looks_like_series else TSeries(values=signal)


def _quarter_scaled(values):
    """Map the signal onto [-0.25, +0.25] — the intent stated at ``phase.py:65-66`` (NaN-aware
    extrema, ``core.py:202-240``)."""
    top, bottom = np.nanmax(values), np.nanmin(values)
    return (values - top) / (2 * (top - bottom)) + 0.25


def _string_periods(baseline, dphi, count):
    """``count`` trial periods equally spaced in FREQUENCY, from ``baseline / (dphi * count)`` up
    to ``baseline / dphi`` (``phase.py:67-68``)."""
    step = dphi / baseline
    return 1 / np.linspace(count * step, step, count)


def _pdm_periods(signal, p_min, p_max, count, oversample):
    """Trial periods equally spaced in PERIOD (``phase.py:167-180``) and the limits used."""
    span = signal.baseline
    shortest = 2 * signal.median_dt if p_min is None else p_min
    longest = oversample * span if p_max is None else p_max
    if count is None:
        count = int((1 / shortest - 1 / longest) * oversample * span + 1)
    return np.linspace(shortest, longest, count), shortest, longest


def _average_with_double_period(thetas, periods, n_samples, shortest, longest):
    """Sub-harmonic averaging (``phase.py:166,181,188-193``): wherever theta is significant
    (below ``1 - 11 / N**0.8``) and the doubled period is still on the grid, replace theta by the
    mean of itself and the value at twice the period.  All reads use the pre-update values."""
    significant = 1.0 - 11.0 / n_samples ** 0.8
    spacing = periods[1] - periods[0]
    (here,) = np.where((thetas < significant) & (periods <= longest / 2))
    doubled = np.round(2 * here + shortest / spacing).astype(int)
    thetas[here] = (thetas[here] + thetas[doubled]) / 2
    return thetas


class StringLength(object):
    """String Length period search (Dworetsky 1983, MNRAS 203, 917).

    dphi: float
        Frequency step in units of ``1 / baseline`` (0.1 by default).
    n_periods: int
        How many trial periods (1000 by default).
    cores: int, optional
        Ignored on the GPU; clamped to the host's core count exactly like upstream
        (``phase.py:41-43``) so that code reading ``.cores`` keeps working.
    device: int, keyword-only
        GPU ordinal.
    devices: sequence of int, keyword-only
        Several GPUs of this node: the period grid is cut into one contiguous slab per entry —
        the GPU counterpart of upstream's ``Pool(cores)`` fan-out (``phase.py:69-70``).
    """

    def __init__(self, dphi=0.1, n_periods=1000, cores=None, *, device=None, devices=None):
        self.dphi = dphi
        self.n_periods = n_periods
        self.cores = MAX_CORES if cores is None or cores > MAX_CORES else cores
        self.device = device
        self.devices = None if devices is None else tuple(devices)

    def _stringlength(self, period):
        """Length of the closed (phase, magnitude) polygon at one trial period — the seam of
        ``phase.py:45-51`` — evaluated by the same kernel as the full scan."""
        lengths = _cabi.stringlength_scan(self.m.time, self.m.values, [period], device=self.device)
        return float(lengths[0])

    def __call__(self, signal):
        """Scan ``n_periods`` trial periods (``phase.py:53-72``).

        The upstream call cannot run at HEAD (it hands a list to ``FSeries`` and subtracts
        misaligned one-element series — SURVEY.md fact 4); this does what its comments say:
        scale the signal to [-0.25, +0.25], fold at each period, sort by phase, sum the closed
        polygon.  Returned on ascending frequency like every ``FSeries``.
        """
        signal = _coerce(signal)
        self.signal = signal
        times = np.asarray(signal.time, dtype=float)
        self.m = TSeries(times, _quarter_scaled(np.asarray(signal.values, dtype=float)),
                         assume_sorted=True)
        periods = _string_periods(signal.baseline, self.dphi, self.n_periods)
        lengths = _cabi.stringlength_scan(times, self.m.values, periods, device=self.device,
                                          devices=self.devices)
        self.periodogram = FSeries(1 / periods, lengths)
        return self.periodogram


class PDM(object):
    """Phase Dispersion Minimization (Stellingwerf 1978, ApJ 224, 953; Stellingwerf 2011).

    nb, nc: int
        Bins per cover and number of covers (5 and 2 by default): every sample falls in ``nc``
        overlapping bins of width ``1 / nb``.
    p_min, p_max: float, optional
        Shortest / longest trial period; by default twice the median sampling step and
        ``oversample`` times the baseline.
    n_periods: int or None
        Number of trial periods (1000 by default; ``None`` derives it from the frequency range).
    oversample: scalar
        See ``p_max``.
    do_subharmonic: bool
        Average theta at each significant period with theta at twice that period: a real
        variation shows at both, noise does not.
    cores: int, optional
        Stored but unused on the GPU.
    device: int, keyword-only
        GPU ordinal.
    devices: sequence of int, keyword-only
        Several GPUs of this node, one contiguous slab of the period grid each (upstream's
        ``Pool(cores)``, ``phase.py:182-186``).
    """

    def __init__(self, nb=5, nc=2, p_min=None, p_max=None, n_periods=1000, oversample=1,
                 do_subharmonic=False, cores=None, *, device=None, devices=None):
        self.nb, self.nc = nb, nc
        self.p_min, self.p_max = p_min, p_max
        self.n_periods = n_periods
        self.oversample = oversample
        self.do_subharmonic = do_subharmonic
        self.cores = cores
        self.device = device
        self.devices = None if devices is None else tuple(devices)

    def _scan(self, periods):
        return _cabi.pdm_scan(self.t, self.x, periods, self.nb, self.nc, self.sigma,
                              device=self.device, devices=self.devices)

    def _pdm(self, period):
        """Stellingwerf's theta at one trial period — the seam of ``phase.py:128-149``."""
        return float(self._scan([period])[0])

    def __call__(self, signal):
        """theta (Eq. 3 of the 1978 paper) on the trial-period grid (``phase.py:151-195``)."""
        signal = _coerce(signal)
        self.signal = signal
        self.t = np.asarray(signal.time, dtype=float)
        self.x = np.asarray(signal.values, dtype=float)
        self.sigma = np.var(signal.values, ddof=1)
        self.periods, shortest, longest = _pdm_periods(signal, self.p_min, self.p_max,
                                                       self.n_periods, self.oversample)
        thetas = self._scan(self.periods)
        if self.do_subharmonic:
            thetas = _average_with_double_period(thetas, self.periods, signal.size, shortest, longest)
        self.periodogram = FSeries(1 / self.periods, thetas)
        return self.periodogram


class AOV(object):
    """Analysis of Variance period search (Schwarzenberg-Czerny 1989) - one of the scans the
    reference lists as TODO (``phase.py:11``), shaped like :class:`PDM` and computed by the same
    binning kernel (``csrc/pdm.hip``): the statistic is large where the folded curve is coherent.

    Parameters
    ----------
    n_bins: int, optional
        Number of phase bins r (the default is 10).
    p_min, p_max, n_periods, oversample, cores:
        The trial-period grid, exactly as for :class:`PDM` (``phase.py:167-180``).
    device: int, keyword-only
        GPU ordinal.
    devices: sequence of int, keyword-only
        Several GPUs of this node, one contiguous slab of the period grid each.
    """

    def __init__(self, n_bins=10, p_min=None, p_max=None, n_periods=1000, oversample=1, cores=None,
                 *, device=None, devices=None):
        self.n_bins = n_bins
        self.p_min, self.p_max = p_min, p_max
        self.n_periods = n_periods
        self.oversample = oversample
        self.cores = cores
        self.device = device
        self.devices = None if devices is None else tuple(devices)

    def __call__(self, signal):
        signal = _coerce(signal)
        self.signal = signal
        self.t = np.asarray(signal.time, dtype=float)
        self.x = np.asarray(signal.values, dtype=float)
        self.periods, _, _ = _pdm_periods(signal, self.p_min, self.p_max, self.n_periods,
                                          self.oversample)
        theta = _cabi.aov_scan(self.t, self.x, self.periods, self.n_bins, device=self.device,
                               devices=self.devices)
        self.periodogram = FSeries(1 / self.periods, theta)
        return self.periodogram


class SuperSmoother(object):
    """Supersmoother period search - the reference only names it (``spectral.py:8``: "TODO: check out
    Supersmoother (Reimann 1994)"); shaped like :class:`PDM`.  Per trial period the curve is folded and sorted by
    phase, Friedman's variable span smoother (Friedman 1984: three running-lines smooths with spans 0.05 / 0.2 /
    0.5 of the curve, periodic in phase, the span chosen per point by leave-one-out residuals) is fitted to it,
    and the periodogram is the mean absolute residual about that fit (Reimann 1994): minimal at the period.

    Parameters
    ----------
    alpha: float, optional
        Friedman's bass control in [0, 10]: larger values pull the chosen spans towards the widest one
        (smoother fits); 0 (the default) switches it off.
    p_min, p_max, n_periods, oversample, cores:
        The trial-period grid, exactly as for :class:`PDM` (``phase.py:167-180``).
    device / devices: keyword-only
        GPU ordinal / several GPUs of this node, one contiguous slab of the period grid each.
    """

    def __init__(self, alpha=0.0, p_min=None, p_max=None, n_periods=1000, oversample=1, cores=None,
                 *, device=None, devices=None):
        self.alpha = alpha
        self.p_min, self.p_max = p_min, p_max
        self.n_periods = n_periods
        self.oversample = oversample
        self.cores = cores
        self.device = device
        self.devices = None if devices is None else tuple(devices)

    def __call__(self, signal):
        signal = _coerce(signal)
        self.signal = signal
        self.t = np.asarray(signal.time, dtype=float)
        self.x = np.asarray(signal.values, dtype=float)
        self.periods, _, _ = _pdm_periods(signal, self.p_min, self.p_max, self.n_periods,
                                          self.oversample)
        stat = _cabi.supersmoother_scan(self.t, self.x, self.periods, self.alpha, device=self.device,
                                        devices=self.devices)
        self.periodogram = FSeries(1 / self.periods, stat)
        return self.periodogram


class ConditionalEntropy(object):
    """Conditional-entropy period search (Graham et al. 2013) - TODO upstream (``phase.py:15``),
    shaped like :class:`PDM`: the entropy of the magnitudes given the phase, over an
    ``n_phase x n_mag`` partition of the folded, unit-normalised light curve; minimal at the period.

    Parameters
    ----------
    n_phase, n_mag: int, optional
        Phase and magnitude bins (defaults 10 and 5).
    p_min, p_max, n_periods, oversample, cores:
        The trial-period grid, exactly as for :class:`PDM`.
    device: int, keyword-only
        GPU ordinal.
    devices: sequence of int, keyword-only
        Several GPUs of this node, one contiguous slab of the period grid each.
    """

    def __init__(self, n_phase=10, n_mag=5, p_min=None, p_max=None, n_periods=1000, oversample=1,
                 cores=None, *, device=None, devices=None):
        self.n_phase, self.n_mag = n_phase, n_mag
        self.p_min, self.p_max = p_min, p_max
        self.n_periods = n_periods
        self.oversample = oversample
        self.cores = cores
        self.device = device
        self.devices = None if devices is None else tuple(devices)

    def __call__(self, signal):
        signal = _coerce(signal)
        self.signal = signal
        self.t = np.asarray(signal.time, dtype=float)
        values = np.asarray(signal.values, dtype=float)
        low, high = np.nanmin(values), np.nanmax(values)
        unit = (values - low) / (high - low)
        self.mag_bin = np.minimum(np.floor(unit * self.n_mag), self.n_mag - 1).astype(float)
        self.periods, _, _ = _pdm_periods(signal, self.p_min, self.p_max, self.n_periods,
                                          self.oversample)
        entropy = _cabi.cond_entropy_scan(self.t, self.mag_bin, self.periods, self.n_phase, self.n_mag,
                                          device=self.device, devices=self.devices)
        self.periodogram = FSeries(1 / self.periods, entropy)
        return self.periodogram


class GregoryLoredo(object):
    """Gregory-Loredo period search (Gregory & Loredo 1992, ApJ 398, 146) - the third scan the reference
    lists as TODO (``phase.py:13``), shaped like :class:`PDM`.  It works on ARRIVAL TIMES: the time stamps of
    ``signal`` are the events, its values are not used.  Model ``M_m`` is a periodic rate that is constant
    in each of ``m`` phase bins; per trial frequency the data enter through the multiplicity of the bin
    counts, ``W_m = N! / (n_1! ... n_m!)``, marginalised over the unknown offset of the bins (their eq.
    5.13-5.14).  The periodogram returned is the log of the per-frequency odds in favour of a periodic
    signal, ``ln sum_{m=2}^{m_max} O_m1(w) / (m_max - 1)`` with
    ``O_m1(w) = N! (m-1)! / (N+m-1)! * <m^N / W_m(w, phi)>_phi`` (the integrand of eq. 5.28; equal prior
    odds for every ``m``); it peaks at the period.  ``.log_s[m]`` keeps ``ln <m^N / W_m>`` per ``m``.

    Parameters
    ----------
    m_max: int, optional
        Largest number of phase bins (the default is 12, as in the paper); ``m`` runs over 2 .. m_max.
    n_offsets: int, optional
        Shifts of the bin boundaries the offset integral is averaged over (the default is 8);
        ``m_max * n_offsets`` must not exceed 190.
    p_min, p_max, n_periods, oversample, cores:
        The trial-period grid, exactly as for :class:`PDM` (``phase.py:167-180``).
    device / devices: keyword-only
        GPU ordinal / several GPUs of this node, one contiguous slab of the period grid each.
    """

    def __init__(self, m_max=12, n_offsets=8, p_min=None, p_max=None, n_periods=1000, oversample=1, cores=None,
                 *, device=None, devices=None):
        self.m_max, self.n_offsets = m_max, n_offsets
        self.p_min, self.p_max = p_min, p_max
        self.n_periods = n_periods
        self.oversample = oversample
        self.cores = cores
        self.device = device
        self.devices = None if devices is None else tuple(devices)

    def __call__(self, signal):
        from scipy.special import gammaln, logsumexp
        signal = _coerce(signal)
        self.signal = signal
        self.t = np.asarray(signal.time, dtype=float)
        self.periods, _, _ = _pdm_periods(signal, self.p_min, self.p_max, self.n_periods, self.oversample)
        n = int(np.sum(np.isfinite(self.t)))
        self.log_s, terms = {}, []
        for m in range(2, self.m_max + 1):
            self.log_s[m] = _cabi.gl_scan(self.t, self.periods, m, self.n_offsets, device=self.device,
                                          devices=self.devices)
            # ln [N! (m-1)! / (N+m-1)!]: the prior volume of the m bin heights
            terms.append(self.log_s[m] + gammaln(n + 1) + gammaln(m) - gammaln(n + m))
        log_odds = logsumexp(np.array(terms), axis=0) - np.log(self.m_max - 1)
        self.periodogram = FSeries(1 / self.periods, log_odds)
        return self.periodogram
