"""Generate the golden vectors under tests/golden/ from the reference's own code.

Runs ONLY in the build container, where /root/reference exists: the reference's
``spectral.py`` / ``phase.py`` are executed verbatim through ``oracle/refstub.py`` (which swaps
in numpy-only containers for the xarray-backed ``periodicity.core``).  The vectors are data —
inputs and expected outputs — and travel to the GPU box; the reference never does.

    python tests/golden/make_golden.py

Two flavours of GLS output are stored per case:
  power_ref    the unmodified reference (FFT/extirpolation approximation)        -> Tier R
  power_exact  the reference's ``GLS.__call__`` run verbatim with ``_trig_sum`` replaced by an
               80-bit long-double evaluation of the sums its docstring defines     -> Tier E
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refstub  # noqa: E402
from periodicity_amd.core import TSeries  # noqa: E402

GENERATOR_VERSION = 1
TWO_PI_L = np.longdouble(2) * np.arctan2(np.longdouble(0), np.longdouble(-1))


def exact_trig_sum(t, w, df, nf, fmin, n=5):
    """Drop-in for the reference's ``_trig_sum`` evaluating its docstring directly in long double."""
    tl = np.asarray(t, dtype=np.longdouble)
    wl = np.asarray(w, dtype=np.longdouble)
    f = np.longdouble(fmin) + np.longdouble(df) * np.arange(nf, dtype=np.longdouble)
    S = np.empty(nf, dtype=np.longdouble)
    C = np.empty(nf, dtype=np.longdouble)
    step = max(1, (1 << 20) // max(1, tl.size))
    for a in range(0, nf, step):
        ph = TWO_PI_L * np.outer(f[a:a + step], tl)
        S[a:a + step] = np.sin(ph) @ wl
        C[a:a + step] = np.cos(ph) @ wl
    return S.astype(float), C.astype(float)


def synthetic_curve(n, seed, period=37.3):
    """SURVEY.md §8d recipe (draw order t, dy, noise)."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0, float(n), n))
    dy = rng.uniform(0.05, 0.2, n)
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / period) + dy * rng.standard_normal(n)
    return t, y, dy


def both_powers(spectral, make_gls, call):
    """Run ``call(gls)`` with the stock and with the exact trig sums."""
    stock = spectral._trig_sum
    ref = call(make_gls())
    spectral._trig_sum = exact_trig_sum
    try:
        exact = call(make_gls())
    finally:
        spectral._trig_sum = stock
    return ref, exact


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    arrays["generator_version"] = np.array(GENERATOR_VERSION)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def model_golden(GLS):
    # G9 - GLS.model (spectral.py:169-204): the weighted sinusoid fit at the peak frequency and at an
    # off-peak one, with heteroscedastic errors and with err=None (all-ones errors)
    t, y, dy = synthetic_curve(400, 20241008 + 9, period=17.0)
    tf = np.linspace(t[0] - 3.0, t[-1] + 3.0, 257)
    out = dict(t=t, y=y, dy=dy, tf=tf)
    for tag, err in (("err", dy), ("noerr", None)):
        g = GLS()
        ls = g(TSeries(t, y), err=err)
        f_peak = float(ls.frequency[np.nanargmax(ls.values)])
        out["f0_" + tag] = np.array([f_peak, 0.0371])
        out["yf_" + tag] = np.stack([np.asarray(g.model(tf, f).values) for f in out["f0_" + tag]])
        out["tf_sorted_" + tag] = np.asarray(g.model(tf, f_peak).time)
    save("g9_model", **out)


def main():
    spectral, phase = refstub.load()
    GLS = spectral.GLS
    model_golden(GLS)
    if "--only-g9" in sys.argv:
        return

    # G1 — tests/test_spectral.py:7-24 (default grid; values default to ones)
    time = np.arange(0, 2.5 + 0.1, 0.1)
    ls = GLS(n=1)(TSeries(time))
    save("g1_grid", time=time, frequency=ls.frequency, power_ref=ls.values)

    # G2 — tests/test_spectral.py:27-31 (100-sample sine, 10 cycles)
    values = np.sin((np.arange(100) / 100) * 20 * np.pi)
    ref, exact = both_powers(spectral, lambda: GLS(), lambda g: g(TSeries(values=values)))
    save("g2_sine100", values=values, frequency=ref.frequency, power_ref=ref.values,
         power_exact=exact.values, argmax=np.array(ref.argmax()),
         period_at_highest_peak=np.array(ref.period_at_highest_peak))

    # G3 — SpottedStar (real uneven sampling, heteroscedastic dy), all four flag combinations
    t, y, dy = np.load("/root/reference/src/periodicity/data/spotted_star.npy")
    out = dict(t=t, y=y, dy=dy)
    for fit_mean in (True, False):
        for psd in (False, True):
            ref, exact = both_powers(spectral, lambda: GLS(psd=psd),
                                     lambda g: g(TSeries(t, y), err=dy, fit_mean=fit_mean))
            tag = f"fm{int(fit_mean)}_psd{int(psd)}"
            out["frequency"] = ref.frequency
            out["power_ref_" + tag] = ref.values
            out["power_exact_" + tag] = exact.values
    ref, exact = both_powers(spectral, lambda: GLS(), lambda g: g(TSeries(t, y)))
    out["power_ref_noerr"] = ref.values
    out["power_exact_noerr"] = exact.values
    save("g3_spotted_star", **out)

    # G4 — seeded uneven synthetic curves + the three seam-level trig sums
    for n in (1000, 5000):
        t, y, dy = synthetic_curve(n, 20241008 + n)
        ref, exact = both_powers(spectral, lambda: GLS(),
                                 lambda g: g(TSeries(t, y), err=dy, fit_mean=True))
        freq = ref.frequency
        df = 1.0 / (t[-1] - t[0]) / 5
        fmin = 0.5 * df
        w = dy ** -2.0
        w /= w.sum()
        yc = y - np.dot(w, y)
        seams = {}
        for label, fn in (("fft", spectral._trig_sum), ("exact", exact_trig_sum)):
            seams["Sh_" + label], seams["Ch_" + label] = fn(t, w * yc, df, freq.size, fmin)
            seams["S2_" + label], seams["C2_" + label] = fn(t, w, 2 * df, freq.size, 2 * fmin)
            seams["S_" + label], seams["C_" + label] = fn(t, w, df, freq.size, fmin)
        save(f"g4_synth{n}", t=t, y=y, dy=dy, seed=np.array(20241008 + n), frequency=freq,
             df=np.array(df), fmin=np.array(fmin), power_ref=ref.values,
             power_exact=exact.values, **seams)

    # G5 — window(): all-ones signal, fit_mean=False
    t, y, dy = synthetic_curve(800, 20241008 + 5)

    def window_of(g):
        g(TSeries(t, y), err=dy)
        return g.window()

    ref, exact = both_powers(spectral, lambda: GLS(), window_of)
    save("g5_window", t=t, y=y, dy=dy, frequency=ref.frequency, power_ref=ref.values,
         power_exact=exact.values)

    # G6 — bootstrap(20, random_seed=42): pins the RNG draw order of spectral.py:141-150
    t, y, dy = synthetic_curve(300, 20241008 + 6, period=11.0)

    def boot(g):
        g(TSeries(t, y), err=dy)
        reps = g.bootstrap(20, random_seed=42)
        return np.array(reps), g.fap(0.3), g.fal(0.1)

    (r_ref, fap_ref, fal_ref), (r_ex, fap_ex, fal_ex) = both_powers(spectral, lambda: GLS(), boot)
    save("g6_bootstrap", t=t, y=y, dy=dy, replicates_ref=r_ref, replicates_exact=r_ex,
         fap_at_0p3_ref=np.array(fap_ref), fal_at_0p1_ref=np.array(fal_ref),
         fap_at_0p3_exact=np.array(fap_ex), fal_at_0p1_exact=np.array(fal_ex))

    # G7 — PDM: the per-period seam and the whole call, both bin layouts, sub-harmonics,
    # negative times
    t, y, _ = synthetic_curve(2000, 20241008 + 7, period=13.7)
    out = dict(t=t, y=y)
    for nb, nc in ((5, 2), (10, 3)):
        pdm = phase.PDM(nb=nb, nc=nc, p_min=1.0, p_max=60.0, n_periods=200, cores=1)
        res = pdm(TSeries(t, y))
        out[f"periods_{nb}_{nc}"] = pdm.periods
        out[f"theta_seam_{nb}_{nc}"] = np.array([pdm._pdm(p) for p in pdm.periods])
        out[f"frequency_{nb}_{nc}"] = res.frequency
        out[f"theta_call_{nb}_{nc}"] = res.values
    pdm = phase.PDM(p_min=1.0, p_max=60.0, n_periods=200, do_subharmonic=True, cores=1)
    res = pdm(TSeries(t, y))
    out["frequency_sub"], out["theta_call_sub"] = res.frequency, res.values
    pdm = phase.PDM(n_periods=150, cores=1)  # default p_min / p_max
    res = pdm(TSeries(t, y))
    out["periods_default"], out["theta_call_default"] = pdm.periods, res.values
    out["frequency_default"] = res.frequency
    tneg = t - 700.25
    pdm = phase.PDM(p_min=1.0, p_max=60.0, n_periods=200, cores=1)
    res = pdm(TSeries(tneg, y))
    out["t_negative"] = tneg
    out["theta_call_negative"] = res.values
    out["sigma"] = np.array(pdm.sigma)
    save("g7_pdm", **out)

    # G8 — StringLength: per-period seam on the restated scaling/grid (the call itself is
    # broken upstream)
    t, y, _ = synthetic_curve(2000, 20241008 + 8, period=13.7)
    m = (y - np.nanmax(y)) / (2 * (np.nanmax(y) - np.nanmin(y))) + 0.25     # phase.py:65-66
    df = 0.1 / (t[-1] - t[0])                                                # phase.py:67
    periods = 1 / np.linspace(200 * df, df, 200)                             # phase.py:68
    sl = phase.StringLength(n_periods=200, cores=1)
    sl.m = TSeries(t, m)
    ell = np.array([sl._stringlength(p) for p in periods])
    # evenly sampled: phases repeat exactly, which exercises the stable tie order
    te = np.arange(500.0)
    ye = np.sin(2 * np.pi * te / 12.5) + 0.1 * np.cos(te)
    me = (ye - ye.max()) / (2 * (ye.max() - ye.min())) + 0.25
    pe = np.concatenate([[1.0, 2.0, 2.5, 4.0, 5.0, 12.5, 25.0, 50.0],
                         1 / np.linspace(120 * 0.1 / 499.0, 0.1 / 499.0, 120)])
    sl.m = TSeries(te, me)
    elle = np.array([sl._stringlength(p) for p in pe])
    save("g8_stringlength", t=t, y=y, m=m, periods=periods, ell=ell,
         t_even=te, m_even=me, periods_even=pe, ell_even=elle)


if __name__ == "__main__":
    if not refstub.available():
        print("reference sources not present: nothing to do")
        sys.exit(0)
    main()
