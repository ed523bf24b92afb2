"""Phase-folding period scans with the reference's callable API, computed on MI355X.

Drop-in for ``periodicity.phase`` (``/root/reference/src/periodicity/phase.py``):
``StringLength(dphi, n_periods, cores)(signal) -> FSeries`` and
``PDM(nb, nc, p_min, p_max, n_periods, oversample, do_subharmonic, cores)(signal) -> FSeries``
with the same positional order, defaults and attribute side effects.  The reference maps one
Python task per trial period over a ``multiprocessing.Pool`` (``phase.py:69-70,185-186``); here
the whole period grid is one kernel launch (``csrc/stringlength.hip``, ``csrc/pdm.hip``), so
``cores`` is accepted for compatibility and ignored.  Nothing here computes a scan on the CPU.
"""
from multiprocessing import cpu_count

import numpy as np

from . import _cabi
from .core import FSeries, TSeries

MAX_CORES = cpu_count()

__all__ = ["StringLength", "PDM"]


def _as_tseries(signal):
    if isinstance(signal, TSeries) or (hasattr(signal, "time") and hasattr(signal, "values")
                                       and hasattr(signal, "baseline")):
        return signal
    return TSeries(values=signal)


class StringLength(object):
    """String Length (Dworetsky 1983).

    Parameters (``phase.py:19-43``)
    ----------
    dphi: float, optional
        Factor multiplying ``1 / baseline`` to get the frequency separation (default 0.1).
    n_periods: int, optional
        Number of trial periods (default 1000).
    cores: int, optional
        Accepted for compatibility with the reference's process pool; unused on the GPU.
    device: int, keyword-only, optional
        GPU ordinal.
    """

    def __init__(self, dphi=0.1, n_periods=1000, cores=None, *, device=None):
        self.dphi = dphi
        self.n_periods = n_periods
        if cores is None or cores > MAX_CORES:
            cores = MAX_CORES
        self.cores = cores
        self.device = device

    def _stringlength(self, period):
        """String length for a single trial period (the seam of ``phase.py:45-51``)."""
        ell = _cabi.stringlength_scan(self.m.time, self.m.values, [period], device=self.device)
        return float(ell[0])

    def __call__(self, signal):
        """String length on ``n_periods`` trial periods uniform in frequency between
        ``baseline / (dphi * n_periods)`` and ``baseline / dphi`` (``phase.py:53-72``).

        The upstream call is broken at HEAD (it hands a list to ``FSeries`` and subtracts
        misaligned one-element series); this implements what its comments state: scale the signal
        to [-0.25, +0.25] (``phase.py:65-66``), fold, sort by phase, sum the closed polygon.
        """
        signal = _as_tseries(signal)
        self.signal = signal
        values = np.asarray(signal.values, dtype=float)
        vmax, vmin = np.nanmax(values), np.nanmin(values)
        self.m = TSeries(signal.time, (values - vmax) / (2 * (vmax - vmin)) + 0.25,
                         assume_sorted=True)
        df = self.dphi / signal.baseline
        periods = 1 / np.linspace(self.n_periods * df, df, self.n_periods)
        ell = _cabi.stringlength_scan(np.asarray(signal.time, dtype=float), self.m.values, periods,
                                      device=self.device)
        self.periodogram = FSeries(1 / periods, ell)
        return self.periodogram


class PDM(object):
    """Phase Dispersion Minimization (Stellingwerf 1978).

    Parameters (``phase.py:75-126``)
    ----------
    nb: int, optional
        Number of phase bins (default 5).
    nc: int, optional
        Number of covers per bin (default 2).
    p_min, p_max: float, optional
        Minimum / maximum trial period (defaults ``2 * median_dt`` and ``oversample * baseline``).
    n_periods: int, optional
        Number of trial periods (default 1000); ``None`` derives it from the frequency range.
    oversample: scalar, optional
        Baseline multiplier used when ``p_max`` is omitted.
    do_subharmonic: bool, optional
        Average theta at each significant period with theta at its double.
    cores: int, optional
        Accepted for compatibility with the reference's process pool; unused on the GPU.
    device: int, keyword-only, optional
        GPU ordinal.
    """

    def __init__(self, nb=5, nc=2, p_min=None, p_max=None, n_periods=1000, oversample=1,
                 do_subharmonic=False, cores=None, *, device=None):
        self.nb = nb
        self.nc = nc
        self.p_min = p_min
        self.p_max = p_max
        self.n_periods = n_periods
        self.oversample = oversample
        self.do_subharmonic = do_subharmonic
        self.cores = cores
        self.device = device

    def _pdm(self, period):
        """theta for a single trial period (the seam of ``phase.py:128-149``)."""
        theta = _cabi.pdm_scan(self.t, self.x, [period], self.nb, self.nc, self.sigma,
                               device=self.device)
        return float(theta[0])

    def __call__(self, signal):
        """theta statistic on ``n_periods`` trial periods uniform in period
        (``phase.py:151-195``); returned on ascending frequency like every ``FSeries``."""
        signal = _as_tseries(signal)
        self.signal = signal
        self.t = np.asarray(signal.time, dtype=float)
        self.x = np.asarray(signal.values, dtype=float)
        self.sigma = np.var(signal.values, ddof=1)
        theta_crit = 1.0 - 11.0 / signal.size ** 0.8
        t0 = signal.baseline
        p_min = 2 * signal.median_dt if self.p_min is None else self.p_min
        p_max = self.oversample * t0 if self.p_max is None else self.p_max
        if self.n_periods is None:
            n_periods = int((1 / p_min - 1 / p_max) * self.oversample * t0 + 1)
        else:
            n_periods = self.n_periods
        self.periods = np.linspace(p_min, p_max, n_periods)
        thetas = _cabi.pdm_scan(self.t, self.x, self.periods, self.nb, self.nc, self.sigma,
                                device=self.device)
        if self.do_subharmonic:
            dp = self.periods[1] - self.periods[0]
            (can_average,) = np.where((thetas < theta_crit) & (self.periods <= p_max / 2))
            sub_indices = np.round(2 * can_average + p_min / dp).astype(int)
            thetas[can_average] = (thetas[can_average] + thetas[sub_indices]) / 2
        self.periodogram = FSeries(1 / self.periods, thetas)
        return self.periodogram
