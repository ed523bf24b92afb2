"""Sanity + timing of the StringLength/PDM kernels at large N (many phase slices)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from periodicity_amd import _cabi
from oracle import c_oracle as co, scan_oracle as so
rng = np.random.default_rng(1)
for n, n_per in ((200_000, 2048), (1_000_000, 256)):
    t = np.sort(rng.uniform(0, float(n), n))
    y = np.sin(2 * np.pi * t / 13.7) + 0.1 * rng.standard_normal(n)
    m = so.stringlength_scale(y)
    periods = np.linspace(1.0, 100.0, n_per)
    _cabi.stringlength_scan(t, m, periods[:8])      # first call pays for HIP start-up and the workspace
    _cabi.stringlength_scan(t, m, periods)
    t0 = time.perf_counter(); ell = _cabi.stringlength_scan(t, m, periods); dt_sl = time.perf_counter() - t0
    _cabi.pdm_scan(t, y, periods, 5, 2, np.var(y, ddof=1))
    t0 = time.perf_counter(); th = _cabi.pdm_scan(t, y, periods, 5, 2, np.var(y, ddof=1)); dt_pdm = time.perf_counter() - t0
    pick = np.array([0, n_per // 3, n_per - 1])
    e_sl = np.max(np.abs(ell[pick] - co.stringlength_scan(t, m, periods[pick])) / ell[pick])
    e_pdm = np.max(np.abs(th[pick] - co.pdm_scan(t, y, periods[pick], 5, 2)) / th[pick])
    print(json.dumps({"N": n, "periods": n_per, "SL_ms": round(dt_sl * 1e3, 1), "SL_Gpair_s": round(n * n_per / dt_sl / 1e9, 1),
                      "PDM_ms": round(dt_pdm * 1e3, 1), "PDM_Gpair_s": round(n * n_per / dt_pdm / 1e9, 1),
                      "SL_rel_err": float(e_sl), "PDM_rel_err": float(e_pdm)}))
