"""Numpy-only labelled 1-D containers for the scan path: ``TSeries`` in, ``FSeries`` out.

The reference builds these on xarray (``/root/reference/src/periodicity/core.py:53-58``).
Only the slice of that surface which the trial-frequency scan reads or writes is provided
here, on plain numpy arrays:

* ``TSeries`` — ctor defaults and sort-by-time of ``core.py:460-477``; ``time``, ``values``
  (settable, ``core.py:60-66``), ``size/shape/len`` (``:94-103``), ``baseline`` (``:504-506``),
  ``median_dt`` (``:508-510``), ``dt`` (``:512-519``), ``copy`` (``:144-145``), ``fold``
  (``:543-544``), ``timeshift/timescale`` (``:537-541``), NaN-aware reductions
  (``:192-260``), scalar arithmetic through the numpy ufunc protocol (``:158-187``).
* ``FSeries`` — ``period = 1/frequency`` coordinate and sort-by-frequency of
  ``core.py:859-881``; ``frequency/period/values``, slicing (``:897-902``), ``fmax/pmax``
  (``:938-942``), ``find_peaks`` (``:283-317``) and the peak pickers built on it
  (``:944-978``).

Objects of the real ``periodicity.core`` classes are accepted anywhere these are: the scan
classes only use the attributes named above (duck typing).
"""
from numbers import Number

import numpy as np
from scipy import signal as _sps

__all__ = ["TSeries", "FSeries"]


def _is_sorted(a):
    return a.size < 2 or bool(np.all(a[1:] >= a[:-1]))


class Signal(np.lib.mixins.NDArrayOperatorsMixin):
    """A value array labelled by one monotonically increasing coordinate."""

    _dim = "index"
    _handled = (Number, np.ndarray)

    def __init__(self, coord, values, assume_sorted=False):
        coord = np.asarray(coord)
        values = np.asarray(values)
        if coord.ndim != 1 or values.ndim != 1:
            raise ValueError("Only one-dimensional signals are supported.")
        if coord.size != values.size:
            raise ValueError("Input arrays have incompatible lengths.")
        if not assume_sorted and not _is_sorted(coord):
            # xarray's sortby is a lexsort, i.e. stable (core.py:473-477, 877-881)
            order = np.argsort(coord, kind="stable")
            coord, values = coord[order], values[order]
        self._coord = coord
        self._values = values
        self.attrs = {}

    # -- array surface ---------------------------------------------------------------
    @property
    def values(self):
        return self._values

    @values.setter
    def values(self, new):
        new = np.asarray(new)
        if new.shape != self._values.shape:
            raise ValueError("replacement data must match the signal's shape")
        self._values = new

    @property
    def dims(self):
        return (self._dim,)

    @property
    def size(self):
        return self._values.size

    @property
    def shape(self):
        return self._values.shape

    @property
    def ndim(self):
        return 1

    @property
    def dtype(self):
        return self._values.dtype

    def __len__(self):
        return self._values.shape[0]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._values, dtype=dtype)

    def _like(self, values):
        new = type(self)(self._coord, values, assume_sorted=True)
        new.attrs.update(self.attrs)
        return new

    def copy(self):
        return self._like(self._values.copy())

    def __repr__(self):
        return (f"<{type(self).__name__} ({self._dim}: {self.size})>\n"
                f"{self._dim}: {self._coord!r}\nvalues: {self._values!r}")

    def __getitem__(self, key):
        coord = self._coord[key]
        values = self._values[key]
        if np.ndim(values) < 1:
            return values.item()
        new = type(self)(coord, values)
        return new

    # -- numpy protocol: scalars and raw arrays broadcast against the values ------------
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        for x in inputs + tuple(kwargs.get("out", ())):
            if not isinstance(x, self._handled + (Signal,)):
                return NotImplemented
        if "out" in kwargs:
            kwargs["out"] = tuple(x._values if isinstance(x, Signal) else x
                                  for x in kwargs["out"])
        sigs = [x for x in inputs if isinstance(x, Signal)]
        for other in sigs[1:]:
            if other.size != 1 and sigs[0].size != 1 and not np.array_equal(
                    other._coord, sigs[0]._coord):
                raise ValueError("signals are not aligned on the same coordinate")
        raw = tuple(x._values if isinstance(x, Signal) else x for x in inputs)
        result = getattr(ufunc, method)(*raw, **kwargs)
        if method == "at":
            return None
        if method == "__call__":
            host = max(sigs, key=lambda s: s.size)
            if isinstance(result, tuple):
                return tuple(host._like(r) for r in result)
            if np.ndim(result) == 0:
                return result
            return host._like(result)
        if np.ndim(result) == 0:
            return result.item() if hasattr(result, "item") else result
        return result

    # -- NaN-aware reductions (core.py:192-260) -----------------------------------------
    def argmax(self):
        return int(np.nanargmax(self._values))

    def argmin(self):
        return int(np.nanargmin(self._values))

    def amax(self):
        return np.nanmax(self._values)

    def amin(self):
        return np.nanmin(self._values)

    def max(self):
        i = self.argmax()
        return self[i:i + 1]

    def min(self):
        i = self.argmin()
        return self[i:i + 1]

    def mean(self):
        return np.nanmean(self._values)

    def median(self):
        return np.nanmedian(self._values)

    def sum(self):
        return np.nansum(self._values)

    def std(self, **kw):
        return np.nanstd(self._values, **kw)

    def var(self, **kw):
        return np.nanvar(self._values, **kw)

    # -- peaks (core.py:283-341) ------------------------------------------------------------
    def find_peaks(self, include_edges=False, prominence=0.0, **peak_kwargs):
        """Local maxima with their prominences, as ``scipy.signal.find_peaks`` defines them."""
        maxima, res = _sps.find_peaks(self._values, prominence=prominence, **peak_kwargs)
        if include_edges:
            maxima = np.hstack([0, maxima, -1])
            for key, vals in res.items():
                fill = np.nan if vals.dtype.kind == "f" else -1
                res[key] = np.hstack([fill, vals, fill])
        res["indices"] = maxima
        peaks = self[maxima]
        peaks.attrs.update(res)
        return peaks

    def find_dips(self, include_edges=False, prominence=0.0, **dip_kwargs):
        dips = (-self).find_peaks(include_edges, prominence, **dip_kwargs)
        out = -dips
        out.attrs.update(dips.attrs)
        return out

    def find_zero_crossings(self, height=None, delta=0.0):
        if height is None:
            (idx,) = np.where(np.diff(np.signbit(self._values)))
            return idx
        idx, _ = _sps.find_peaks(-np.abs(self._values), height=-height, prominence=delta)
        return idx


class TSeries(Signal):
    """Values sampled at (possibly uneven) times; always held sorted by time."""

    _dim = "time"

    def __init__(self, time=None, values=None, assume_sorted=False):
        if isinstance(time, TSeries) and values is None:
            time, values = time.time, time.values
        if time is None:
            time = np.arange(len(values))
        if values is None:
            values = np.ones(len(time))
        super().__init__(time, values, assume_sorted)

    @property
    def time(self):
        return self._coord

    @property
    def baseline(self):
        return self._coord[-1] - self._coord[0]

    @property
    def median_dt(self):
        return np.median(np.diff(self._coord))

    @property
    def dt(self):
        if np.allclose(np.diff(self._coord), self.median_dt):
            return self.median_dt
        raise AttributeError(
            "The sampling period is only strictly defined for uniformly sampled signals. "
            "Use median_dt for a median value.")

    def tmax(self):
        return self._coord[self.argmax()].item()

    def timeshift(self, t0):
        return TSeries(self._coord + t0, self._values)

    def timescale(self, alpha):
        return TSeries(self._coord * alpha, self._values)

    def fold(self, period, t0=0):
        """Phase-fold: ``((time - t0) / period) % 1`` re-sorted by phase (core.py:543-544)."""
        return TSeries(((self._coord - t0) / period) % 1, self._values)


class FSeries(Signal):
    """A periodogram: values on a frequency grid, with the derived ``period`` coordinate."""

    _dim = "frequency"

    def __init__(self, frequency=None, values=None, assume_sorted=False):
        if values is None:
            values = np.ones(len(frequency))
        super().__init__(frequency, values, assume_sorted)
        with np.errstate(divide="ignore", invalid="ignore"):
            self._period = 1.0 / self._coord

    @property
    def frequency(self):
        return self._coord

    @property
    def period(self):
        return self._period

    @property
    def median_df(self):
        return np.median(np.diff(self._coord))

    @property
    def df(self):
        if np.allclose(np.diff(self._coord), self.median_df):
            return self.median_df
        raise AttributeError(
            "The sampling period is only strictly defined for uniform frequency grids. "
            "Use median_df for a median value.")

    @property
    def median_dp(self):
        return -np.median(np.diff(self._period))

    def fmax(self):
        return self._coord[self.argmax()].item()

    def pmax(self):
        return self._period[self.argmax()].item()

    def psort_by_peak(self):
        peaks = self.find_peaks()
        return peaks.period[peaks.values.argsort()[::-1]]

    def psort_by_prominence(self):
        peaks = self.find_peaks()
        return peaks.period[peaks.attrs["prominences"].argsort()[::-1]]

    @property
    def period_at_highest_peak(self):
        return self.find_peaks().pmax()

    @property
    def period_at_highest_prominence(self):
        peaks = self.find_peaks()
        return peaks.period[np.nanargmax(peaks.attrs["prominences"])]

    def periods_at_half_max(self, peak_order=1, use_prominence=False):
        peaks = self.find_peaks()
        indices = peaks.attrs["indices"]
        heights = peaks.attrs["prominences"] if use_prominence else peaks.values
        jmax = heights.argsort()[-peak_order]
        idmax = indices[jmax]
        half = self._values[idmax] - heights[jmax] / 2
        left, right = self[:idmax], self[idmax:]
        hi = (left - half).find_zero_crossings()[-1]
        lo = (right - half).find_zero_crossings()[0]
        return right.period[lo], left.period[hi]
