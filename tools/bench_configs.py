"""Kernel-level timings of every BASELINE.json config that fits one GPU (resident inputs,
HIP events on the launch stream).  Not the driver's bench (that is bench.py): this feeds
DESIGN.md's per-kernel table.

    python tools/bench_configs.py [--reps 5] [--only c2,c3,c5]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from periodicity_amd import _cabi  # noqa: E402

lib = _cabi.lib()
DEV = 0


def synth(n, k, period=37.3):
    rng = np.random.default_rng(20241008 + k)
    t = np.sort(rng.uniform(0, float(n), n))
    dy = rng.uniform(0.05, 0.2, n)
    y = 1.0 + 0.5 * np.sin(2 * np.pi * t / period) + dy * rng.standard_normal(n)
    return t, y, dy


def grid(t, nf):
    df = 1.0 / (t[-1] - t[0]) / 5
    fmin = 0.5 * df
    f = np.arange(fmin, fmin + (nf - 1.5) * df + df, df)
    assert f.size == nf
    return _cabi.grid_params(f)


class Timer:
    def __init__(self):
        s = C.c_void_p()
        _cabi.check(lib.pdc_stream_create(DEV, C.byref(s)))
        self.stream = s.value
        self.ev = []
        for _ in range(2):
            e = C.c_void_p()
            _cabi.check(lib.pdc_event_create(DEV, C.byref(e)))
            self.ev.append(e.value)

    def time(self, fn, reps):
        fn()
        _cabi.check(lib.pdc_stream_sync(DEV, self.stream))
        out = []
        for _ in range(reps):
            _cabi.check(lib.pdc_event_record(DEV, self.ev[0], self.stream))
            fn()
            _cabi.check(lib.pdc_event_record(DEV, self.ev[1], self.stream))
            ms = C.c_float()
            _cabi.check(lib.pdc_event_elapsed_ms(DEV, self.ev[0], self.ev[1], C.byref(ms)))
            out.append(ms.value)
        return float(np.median(out)), float(np.min(out))


def dbuf(a):
    return _cabi.DeviceBuffer.from_array(a, DEV)


def run_gls(tm, name, t, y, dy, offsets, nb, f0, delta, nf, reps, shared_t=0, peaks_only=False):
    n_total = y.size
    bt, by, bdy = dbuf(t), dbuf(y), (dbuf(dy) if dy is not None else None)
    boff = dbuf(offsets) if offsets is not None else None
    wb = lib.pdc_gls_work_bytes(n_total, nb, nf)
    work = _cabi.DeviceBuffer(wb, DEV)
    power = None if peaks_only else _cabi.DeviceBuffer(nb * nf * 8, DEV)
    amax = _cabi.DeviceBuffer(nb * 8, DEV) if peaks_only else None
    arg = _cabi.DeviceBuffer(nb * 8, DEV) if peaks_only else None

    def fn():
        _cabi.check(lib.pdc_gls_scan_dev(DEV, tm.stream, bt.ptr, by.ptr, bdy.ptr if bdy else None,
                                         boff.ptr if boff else None, n_total, nb, shared_t, f0,
                                         delta, 0, nf, 1, 0, power.ptr if power else None,
                                         amax.ptr if amax else None, arg.ptr if arg else None,
                                         work.ptr, wb))
    med, best = tm.time(fn, reps)
    pairs = float(n_total) * nf
    res = {"config": name, "pairs": pairs, "ms_median": round(med, 4), "ms_min": round(best, 4),
           "Gpair_per_s": round(pairs / med / 1e6, 1), "K": os.environ.get("PDC_GLS_K", "8")}
    for b in (bt, by, bdy, boff, work, power, amax, arg):
        if b:
            b.free()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="c1,c2,c2fft,c3,c3peaks,c3shared,c5pdm,c5sl")
    args = ap.parse_args()
    only = set(args.only.split(","))
    tm = Timer()
    out = []
    if "c1" in only:
        t, y, dy = synth(1000, 1)
        out.append(run_gls(tm, "C1 GLS 1k x 1k", t, y, dy, None, 1, *grid(t, 1000), args.reps))
    if "c2" in only:
        t, y, dy = synth(100_000, 2)
        out.append(run_gls(tm, "C2 GLS 1e5 x 1e6", t, y, dy, None, 1, *grid(t, 1_000_000), args.reps))
    if "c2noerr" in only:
        t, y, dy = synth(100_000, 2)
        out.append(run_gls(tm, "C2 GLS 1e5 x 1e6, err=None (equal weights)", t, y, None, None, 1,
                           *grid(t, 1_000_000), args.reps))
    for tag, n_s, nf_s in (("c2fft", 100_000, 1_000_000), ("c4fft", 1_000_000, 10_000_000)):
        if tag in only:
            t, y, dy = synth(n_s, 2)
            f0, delta, nf = grid(t, nf_s)
            df = 1.0 / (t[-1] - t[0]) / 5
            bt, by, bdy = dbuf(t), dbuf(y), dbuf(dy)
            wb = lib.pdc_gls_fft_work_bytes(n_s, nf)
            work = _cabi.DeviceBuffer(wb, DEV)
            power = _cabi.DeviceBuffer(nf * 8, DEV)
            med, best = tm.time(lambda: _cabi.check(lib.pdc_gls_scan_fft_dev(
                DEV, tm.stream, bt.ptr, by.ptr, bdy.ptr, n_s, 0.5 * df, df, nf, 1, 0, power.ptr,
                work.ptr, wb)), args.reps)
            out.append({"config": f"{tag.upper()} GLS FFT-extirpolation path N={n_s} nf={nf} (Tier F)",
                        "pairs": float(n_s) * nf, "ms_median": round(med, 4), "ms_min": round(best, 4),
                        "effective_Gpair_per_s": round(n_s * nf / med / 1e6, 1),
                        "work_MiB": round(wb / 2**20, 1)})
            for b in (bt, by, bdy, work, power):
                b.free()
    if only & {"c3", "c3peaks", "c3shared", "c3sharednoerr"}:
        B, n, nf = 4096, 2000, 50_000
        rng = np.random.default_rng(20241008 + 3)
        ts, ys, dys = [], [], []
        for b in range(B):
            tb = np.sort(rng.uniform(0, float(n), n))
            db = rng.uniform(0.05, 0.2, n)
            ts.append(tb)
            dys.append(db)
            ys.append(1.0 + 0.5 * np.sin(2 * np.pi * tb / (5.0 + 0.01 * b)) + db * rng.standard_normal(n))
        t, y, dy = np.concatenate(ts), np.concatenate(ys), np.concatenate(dys)
        offsets = np.arange(B + 1, dtype=np.int64) * n
        df = 1.0 / n / 5
        f = np.arange(0.5 * df, 0.5 * df + (nf - 1.5) * df + df, df)
        assert f.size == nf
        gp = _cabi.grid_params(f)
        if "c3" in only:
            out.append(run_gls(tm, "C3 GLS batch 4096 x 2k x 5e4 (power out)", t, y, dy, offsets, B, *gp, args.reps))
        if "c3peaks" in only:
            out.append(run_gls(tm, "C3 GLS batch (amax/argmax only)", t, y, dy, offsets, B, *gp, args.reps, peaks_only=True))
        if "c3sharednoerr" in only:
            out.append(run_gls(tm, "C3 GLS batch shared t, err=None (bootstrap of an unweighted fit, peaks only)", ts[0], y, None, offsets, B, *gp, args.reps, shared_t=1, peaks_only=True))
        if "c3shared" in only:
            out.append(run_gls(tm, "C3 GLS batch shared t (bootstrap shape, peaks only)", ts[0], y, dy, offsets, B, *gp, args.reps, shared_t=1, peaks_only=True))
    if only & {"c5pdm", "c5sl"}:
        n, n_per = 50_000, 100_000
        t, y, _ = synth(n, 5, period=13.7)
        bt = dbuf(t)
        if "c5pdm" in only:
            periods = np.linspace(1.0, 100.0, n_per)
            bx, bp = dbuf(y), dbuf(periods)
            bth = _cabi.DeviceBuffer(n_per * 8, DEV)
            sigma = float(np.var(y, ddof=1))
            med, best = tm.time(lambda: _cabi.check(lib.pdc_pdm_scan_dev(
                DEV, tm.stream, bt.ptr, bx.ptr, n, bp.ptr, n_per, 5, 2, sigma, bth.ptr)), args.reps)
            out.append({"config": "C5 PDM 5e4 x 1e5 (nb=5, nc=2)", "pairs": float(n) * n_per,
                        "ms_median": round(med, 4), "ms_min": round(best, 4),
                        "Gpair_per_s": round(n * n_per / med / 1e6, 1)})
        if "c5sl" in only:
            vmax, vmin = y.max(), y.min()
            m = (y - vmax) / (2 * (vmax - vmin)) + 0.25
            df = 0.1 / (t[-1] - t[0])
            periods = 1 / np.linspace(n_per * df, df, n_per)
            bm, bp = dbuf(m), dbuf(periods)
            be = _cabi.DeviceBuffer(n_per * 8, DEV)
            wb = lib.pdc_stringlength_work_bytes(n, n_per)
            work = _cabi.DeviceBuffer(wb, DEV)
            med, best = tm.time(lambda: _cabi.check(lib.pdc_stringlength_scan_dev(
                DEV, tm.stream, bt.ptr, bm.ptr, n, bp.ptr, n_per, be.ptr, work.ptr, wb)), args.reps)
            out.append({"config": "C5 StringLength 5e4 x 1e5", "pairs": float(n) * n_per,
                        "ms_median": round(med, 4), "ms_min": round(best, 4),
                        "Gpair_per_s": round(n * n_per / med / 1e6, 1)})
    for r in out:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
