/*
 * periodicity_hip.h — C ABI of libperiodicity_hip.so: MI355X (gfx950) trial-frequency scans.
 *
 * The reference (dioph/periodicity) is pure Python and has no FFI of its own; the boundary a
 * native replacement slots into is its callable API plus three private seams.
 * Each entry point below names the reference interface it replaces
 * (paths relative to /root/reference/src/periodicity/).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - fp64 only; every array is C-contiguous; sizes are int64_t.
 *   - Functions return 0 on success, a negative pdc_status otherwise; the message of the last
 *     failure on the calling thread is available from pdc_last_error().
 *   - "host" entry points take host pointers: the caller owns every buffer, the library copies
 *     H2D/D2H itself, keeps no host pointer after return and caches its device workspace per
 *     device (freed by pdc_release()).
 *   - "_dev" entry points take device pointers on `device` and enqueue on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream).  They do not synchronise.
 *   - NaN/Inf in the data are not errors: IEEE results propagate exactly as in numpy
 *     (spectral.py:113-128 has no guards).
 *   - No C++ exception crosses this boundary.
 */
#ifndef PERIODICITY_HIP_H
#define PERIODICITY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pdc_status {
    PDC_OK = 0,
    PDC_ERR_INVALID = -1,   /* bad argument (NULL, negative size, incompatible lengths) -> ValueError */
    PDC_ERR_NODEVICE = -2,  /* no usable gfx950 device                                 -> RuntimeError */
    PDC_ERR_HIP = -3,       /* a HIP runtime call failed                               -> RuntimeError */
    PDC_ERR_RCCL = -4,      /* an RCCL call failed                                     -> RuntimeError */
    PDC_ERR_NOMEM = -5      /* device or host allocation failed                        -> MemoryError  */
} pdc_status;

/* ---- runtime ------------------------------------------------------------------------------- */
const char *pdc_last_error(void);
int pdc_version(void);                                   /* 100*major + minor */
int pdc_device_count(int *count);
int pdc_device_info(int device, char *name, int name_len, int *cu_count, int64_t *hbm_bytes,
                    int *clock_khz);
int pdc_release(void);                                   /* free all cached device workspaces */
/* How many device (hipMalloc) and page-locked host (hipHostMalloc) allocations the library has made in
 * this process so far: a cached path (plans, the one-shot `_multi` entry points, the per-device
 * workspaces) shows no increase on its second call with the same sizes. */
int pdc_alloc_counts(int64_t *device_allocs, int64_t *pinned_allocs);
/* TEST HOOKS (not for callers): pin / unpin the scratch block the library keeps per (device, stream) for the `_dev`
 * entry points that take no workspace.  Rule they let the tests drive: ONE holder at a time per (device, stream) -
 * a second pdc_test_scratch_pin on the same stream WAITS until the first is unpinned (so never pin twice from one
 * thread without unpinning: it would wait for itself). */
int pdc_test_scratch_pin(int device, void *stream, int64_t bytes, void **dptr);
int pdc_test_scratch_unpin(int device, void *stream);

/* device memory + events for callers that keep data resident (bench.py, tests) */
int pdc_malloc(int device, int64_t bytes, void **dptr);
int pdc_free(int device, void *dptr);
int pdc_memcpy_h2d(int device, void *dst, const void *src, int64_t bytes);
int pdc_memcpy_d2h(int device, void *dst, const void *src, int64_t bytes);
int pdc_memset(int device, void *dst, int value, int64_t bytes);
int pdc_stream_create(int device, void **stream);
int pdc_stream_destroy(int device, void *stream);
int pdc_stream_sync(int device, void *stream);
int pdc_device_sync(int device);
int pdc_event_create(int device, void **event);
int pdc_event_destroy(int device, void *event);
int pdc_event_record(int device, void *event, void *stream);
int pdc_event_elapsed_ms(int device, void *start, void *stop, float *ms);  /* syncs on stop */
/* Measurement aid (no counterpart upstream): a fixed count of fp64 fmas, 16 independent chains per lane, 4 waves on
 * every SIMD of the device, timed with HIP events on `stream`.  *wave_instr_per_simd x 4 cycles / *ms = the clock the
 * chip sustains under fp64 load right now (it is power-limited there), the figure bench.py prints beside the
 * headline so that a box-to-box difference of the same kernel can be attributed; *memtime_ratio = s_memtime ticks
 * per s_memrealtime tick (100 MHz) over one wave's loop.  iters ~ 10 000 runs about 10 ms. */
int pdc_clock_probe(int device, void *stream, int iters, float *ms, double *wave_instr_per_simd, double *memtime_ratio);

/* ---- generalized Lomb-Scargle --------------------------------------------------------------
 * Replaces GLS.__call__ from the weights onward (spectral.py:99-132): w = dy^-2 / sum(dy^-2)
 * (dy == NULL -> ones, :99-103), y centred by the weighted mean when fit_mean (:105-108), the
 * three _trig_sum calls (:109-112) evaluated as exact direct sums, and the fused epilogue
 * (:113-132).  The grid is the one np.arange(fmin, fmax + df, df) produced on the host (:97):
 * frequency[j] = f0 + j*delta with f0 = frequency[0], delta = frequency[1] - frequency[0]
 * (numpy's own fill rule), j = j_begin .. j_begin + nf - 1; power_out[0..nf) receives that slab.
 */
int pdc_gls_scan(const double *t, const double *y, const double *dy, int64_t n,
                 double f0, double delta, int64_t j_begin, int64_t nf,
                 int fit_mean, int psd, double *power_out, int device);

/* Batch of independent light curves on one shared grid (the shape of GLS.bootstrap,
 * spectral.py:140-152, when shared_t != 0: one time axis, B resampled (y, dy) rows).
 * Curve b occupies [offsets[b], offsets[b+1]) of y/dy (and of t unless shared_t, in which case
 * t has offsets[1]-offsets[0] entries and every curve must have that length).
 * power_out is [B][nf] or NULL; amax_out / argmax_out are [B] or NULL and receive the NaN-aware
 * maximum and its index (Signal.amax / argmax, core.py:202-215), computed on the device. */
int pdc_gls_scan_batch(const double *t, const double *y, const double *dy,
                       const int64_t *offsets, int64_t n_curves, int shared_t,
                       double f0, double delta, int64_t j_begin, int64_t nf,
                       int fit_mean, int psd,
                       double *power_out, double *amax_out, int64_t *argmax_out, int device);

/* A batch of curves dealt to `n_devices` GPU slots in contiguous groups of ceil(n_curves / n_devices)
 * curves (batches shard over curves, no exchange); arguments and outputs as
 * pdc_gls_scan_batch.  With shared_t != 0 this is GLS.bootstrap (spectral.py:140-152) over several GPUs.
 * A device may be listed more than once; per-slot buffers are kept between calls. */
int pdc_gls_scan_batch_multi(const double *t, const double *y, const double *dy,
                             const int64_t *offsets, int64_t n_curves, int shared_t,
                             double f0, double delta, int64_t nf, int fit_mean, int psd,
                             double *power_out, double *amax_out, int64_t *argmax_out,
                             const int *devices, int n_devices);

/* One periodogram with the grid split into contiguous equal slabs over `n_devices` GPUs of this
 * node (one process, one stream per device), gathered with one RCCL all-gather over xGMI. */
int pdc_gls_scan_multi(const double *t, const double *y, const double *dy, int64_t n,
                       double f0, double delta, int64_t nf, int fit_mean, int psd,
                       double *power_out, const int *devices, int n_devices);

/* GLS.bootstrap (spectral.py:140-152) with the replicates given BY INDEX.  Upstream draws
 * `bs = rng.integers(0, ndata, ndata)` per replicate (:146) and runs a full periodogram of
 * (values[bs], err[bs]) on the unchanged time axis, keeping only `.amax()` (:147-150).  Here the caller
 * draws the same integers and passes them as picks[n_boot][n] (int32, row b = replicate b); only the ONE
 * curve (t, y, dy|NULL) and the indices are uploaded, the device prologue gathers y[picks], dy[picks]
 * while it builds the weight table of the shared-time-axis kernel.  Outputs: the NaN-aware maximum of
 * every replicate's spectrum (amax_out[n_boot]) and/or its bin (argmax_out[n_boot]).
 * method 0: exact direct sums on the grid f0 + j*delta; method 1: the reference's FFT/extirpolation
 * path with fmin = f0, df = delta (first listed device only).  Replicates are dealt to the listed device
 * slots in contiguous groups, no exchange.  Any pick outside 0 .. n-1 is PDC_ERR_INVALID. */
int pdc_gls_bootstrap(const double *t, const double *y, const double *dy, int64_t n, const int32_t *picks,
                      int64_t n_boot, double f0, double delta, int64_t nf, int fit_mean, int psd, int method,
                      double *amax_out, int64_t *argmax_out, const int *devices, int n_devices);
/* Device-resident form: inputs, picks and outputs in HBM, no synchronisation; `work` of at least
 * pdc_gls_bootstrap_work_bytes(n, n_boot, nf) bytes on the same device. */
int64_t pdc_gls_bootstrap_work_bytes(int64_t n, int64_t n_boot, int64_t nf);
int pdc_gls_bootstrap_dev(int device, void *stream, const double *d_t, const double *d_y, const double *d_dy,
                          int64_t n, const int32_t *d_picks, int64_t n_boot, double f0, double delta, int64_t nf,
                          int fit_mean, int psd, double *d_amax, int64_t *d_argmax, void *work,
                          int64_t work_bytes);

/* The same shard + all-gather as a persistent plan for callers that scan repeatedly (bench.py, a
 * survey loop over many light curves): per-device sample/power/work buffers, two streams and the
 * events per device and the RCCL communicators (ncclCommInitAll, one process, N devices) are created
 * ONCE by pdc_gls_plan_create and reused by every scan; this is also what pdc_gls_scan_multi keeps
 * cached between calls.  Replaces the per-call process fan-out of phase.py:69-70,185-186 for the
 * spectral path.  `devices` are distinct ordinals below pdc_device_count() (else PDC_ERR_INVALID).
 *   upload   replicate (t, y, dy|NULL) of n <= n_max samples on every device (async, compute streams)
 *   scan     enqueue: device i scans slab i of the nf <= nf_max grid into generation g of its power
 *            buffer, then the grouped ncclAllGather of generation g runs on the communication streams
 *            while the next scan (generation g^1) may already compute; returns without waiting
 *   wait     drain every stream of the plan
 *   download wait, then copy the latest complete power[nf] from device slot `which` (every device
 *            holds the whole array after the gather)
 *   kernel_ms  HIP-event time of the latest slab scan on devices[0]
 *   slot_ms    the same for every slot (ms_out[n_slots]; the slowest slot bounds a sharded scan) */
int pdc_gls_plan_create(const int *devices, int n_devices, int64_t n_max, int64_t nf_max, void **plan);
/* The same plan with `n_slots` LOGICAL slots on ONE physical device ("loopback"): every slot has its own
 * buffers, streams and events, scans its own slab, and the all-gather is replaced by the equivalent
 * device-to-device copies on the communication streams (RCCL refuses two ranks on one device).  The slab
 * arithmetic, padded tails (nf % n_slots != 0, nf < n_slots), generation reuse and event ordering are
 * those of the N-GPU plan, so they can be tested on a 1-GPU box.  PDC_PLAN_EXCHANGE=copy makes
 * pdc_gls_plan_create use the copy exchange between distinct devices too (hipMemcpyPeerAsync). */
int pdc_gls_plan_create_loopback(int device, int n_slots, int64_t n_max, int64_t nf_max, void **plan);
/* n_slots; the size RCCL reports for the plan's communicator (ncclCommCount; 0 = no communicator);
 * exchange: 0 none (one slot), 1 RCCL all-gather, 2 device-to-device copies.  Any pointer may be NULL. */
int pdc_gls_plan_info(void *plan, int *n_slots, int *rccl_ranks, int *exchange);
/* If ncclCommInitAll failed when the plan was built, the plan does NOT fail: it warns on stderr and exchanges the
 * slabs by device-to-device copies (peer access enabled where the devices allow it) - exchange == 2 above - and the
 * reason is kept here (empty string: the communicators were built, or never needed).  PDC_FORCE_RCCL_FAIL=1 injects
 * the failure (tests, `bench.py --loopback`). */
int pdc_gls_plan_init_error(void *plan, char *buf, int buf_len);
int pdc_gls_plan_upload(void *plan, const double *t, const double *y, const double *dy, int64_t n);
int pdc_gls_plan_scan(void *plan, double f0, double delta, int64_t nf, int fit_mean, int psd);
int pdc_gls_plan_wait(void *plan);
int pdc_gls_plan_download(void *plan, double *power_out, int64_t nf, int which);
int pdc_gls_plan_kernel_ms(void *plan, float *ms);
int pdc_gls_plan_slot_ms(void *plan, float *ms_out, int n_slots);
int pdc_gls_plan_destroy(void *plan);

/* Seam-level: replaces _trig_sum(t, w, df, nf, fmin) (spectral.py:11-40) by what its docstring
 * defines (:13-15): S_j = sum_i w_i sin(2 pi f_j t_i), C_j = sum_i w_i cos(2 pi f_j t_i),
 * f_j = f0 + j*delta. */
int pdc_trig_sums(const double *t, const double *w, int64_t n,
                  double f0, double delta, int64_t nf,
                  double *S_out, double *C_out, int device);

/* ---- BGLST: Bayesian generalised Lomb-Scargle with linear trend -----------------------------------------------
 * The reference exports the name (spectral.py:7 `__all__ = ["GLS", "BGLST"]`) for an empty class (:207-208, README:
 * "Bayesian Lomb-Scargle with linear Trend (soon)"); there is nothing upstream to replace or to pin against -
 * PARITY UNPINNED BY THE REFERENCE.  What is computed is the published statistic (Olspert, Pelt, Kapyla & Lehtinen
 * 2018, A&A 615, A111): per trial frequency f_j = f0 + (j_begin + j) delta the log marginal likelihood of
 *     y_i = A cos(2 pi f t_i) + B sin(2 pi f t_i) + alpha tau_i + beta + eps_i,   eps_i ~ N(0, dy_i^2),
 * with independent zero-mean Gaussian priors on A, B (one sigma_A), alpha, beta integrated out analytically;
 * tau = (t - t_ref) / span.  Same kernel skeleton as pdc_gls_scan (eight running sums per pair instead of six).
 * scalars[12] = {W = sum dy^-2, then over the normalised weights w = dy^-2 / W: sum w y^2, sum w y, sum w tau y,
 * sum w tau^2, sum w tau; (t[0] - t_ref) / span; 1 / span; 1 / sigma_A^2, 1 / sigma_alpha^2 (alpha per span),
 * 1 / sigma_beta^2; sum log(2 pi dy_i^2) + log(sigma_A^4 sigma_alpha^2 sigma_beta^2)} - O(N) host work in fp64
 * (periodicity_amd/spectral.py:BGLST).  dy == NULL: unit uncertainties.  Workspace: pdc_gls_work_bytes(n, 1, nf). */
int pdc_bglst_scan(const double *t, const double *y, const double *dy, int64_t n,
                   double f0, double delta, int64_t j_begin, int64_t nf, const double *scalars,
                   double *loglik_out, int device);
int pdc_bglst_scan_dev(int device, void *stream, const double *d_t, const double *d_y, const double *d_dy, int64_t n,
                       double f0, double delta, int64_t j_begin, int64_t nf, const double *scalars,
                       double *d_loglik, void *work, int64_t work_bytes);

/* Device-resident form used by bench.py: inputs already in HBM, no synchronisation.
 * `work` is scratch of at least pdc_gls_work_bytes(n_total, n_curves, nf) bytes on the same
 * device (non-decreasing in n_total and in nf: a buffer sized for the largest call serves all; for a
 * single curve it includes up to 48 MiB for the partial sums of short grids, whose samples are cut
 * into parts as well).  d_offsets may be NULL for a single curve of n_total samples. */
int64_t pdc_gls_work_bytes(int64_t n_total, int64_t n_curves, int64_t nf);
int pdc_gls_scan_dev(int device, void *stream,
                     const double *d_t, const double *d_y, const double *d_dy,
                     const int64_t *d_offsets, int64_t n_total, int64_t n_curves, int shared_t,
                     double f0, double delta, int64_t j_begin, int64_t nf,
                     int fit_mean, int psd,
                     double *d_power, double *d_amax, int64_t *d_argmax,
                     void *work, int64_t work_bytes);

/* ---- generalized Lomb-Scargle by the reference's own algorithm ("Tier F") ------
 * Same inputs/outputs as pdc_gls_scan, but the three _trig_sum calls (spectral.py:109-112) are
 * evaluated the way the reference evaluates them: Press-Rybicki extirpolation onto a grid of
 * nfft = 2^ceil(log2(5 nf)) points + inverse FFT (spectral.py:18-39), all on the device.  It takes
 * the reference's own scalars: fmin and df (spectral.py:88-96), not the np.arange grid.  Result:
 * the reference's `power` including its approximation error (O(N + nfft log nfft) work).
 * pdc_trig_sums_fft is the seam: _trig_sum(t, h, df, nf, fmin) itself. */
int64_t pdc_gls_fft_work_bytes(int64_t n, int64_t nf);
int pdc_gls_scan_fft(const double *t, const double *y, const double *dy, int64_t n,
                     double fmin, double df, int64_t nf, int fit_mean, int psd,
                     double *power_out, int device);
int pdc_gls_scan_fft_dev(int device, void *stream, const double *d_t, const double *d_y,
                         const double *d_dy, int64_t n, double fmin, double df, int64_t nf,
                         int fit_mean, int psd, double *d_power, void *work, int64_t work_bytes);
int pdc_trig_sums_fft(const double *t, const double *h, int64_t n, double df, int64_t nf,
                      double fmin, double *S_out, double *C_out, int device);
/* Batch of curves through the FFT path in one set of launches (layout and outputs as
 * pdc_gls_scan_batch): with shared_t != 0 this is GLS.bootstrap (spectral.py:140-152) evaluated by
 * the reference's own algorithm; the batch is processed in chunks sized to ~8 GiB of grids. */
int pdc_gls_scan_fft_batch(const double *t, const double *y, const double *dy,
                           const int64_t *offsets, int64_t n_curves, int shared_t,
                           double fmin, double df, int64_t nf, int fit_mean, int psd,
                           double *power_out, double *amax_out, int64_t *argmax_out, int device);

/* ---- peak picking on the device --------------------------------------------------
 * Highest local maximum of each of n_curves spectra of nf bins, as
 * FSeries.period_at_highest_peak finds it (core.py:952-955 -> find_peaks :283-317 ->
 * scipy.signal.find_peaks(prominence=0.0) -> NaN-aware max :202-220): strict local maxima, flat
 * tops resolved to their midpoint, first/last bin never a peak, NaN never part of one, ties ->
 * lowest index.  idx_out[b] = -1 (val NaN) when spectrum b has no peak.
 * pdc_gls_batch_highest_peak runs the batched periodogram and the reduction back to back so the
 * spectra never leave HBM (4096 x 5e4 bins = 1.6 GB stay on the device, 64 KB come back). */
int pdc_highest_peak(const double *power, int64_t n_curves, int64_t nf,
                     int64_t *idx_out, double *val_out, int device);
int pdc_highest_peak_dev(int device, void *stream, const double *d_power, int64_t n_curves,
                         int64_t nf, int64_t *d_idx, double *d_val);
int pdc_gls_batch_highest_peak(const double *t, const double *y, const double *dy,
                               const int64_t *offsets, int64_t n_curves, int shared_t,
                               double f0, double delta, int64_t nf, int fit_mean, int psd,
                               int64_t *idx_out, double *val_out, int device);

/* The k <= 1024 highest (by_prominence == 0: FSeries.psort_by_peak, core.py:944-946) or most prominent
 * (psort_by_prominence :948-950, period_at_highest_prominence :957-961) find_peaks() maxima of each
 * spectrum, found and ranked on the device: count[b] = number of maxima scipy.signal.find_peaks(x,
 * prominence=0.0) reports; idx/height/prominence are [n_curves][k], ranked descending, padded with
 * -1 / NaN; prominences follow scipy.signal.peak_prominences (wlen=None).  half_lo / half_hi
 * ([n_curves][k], -1 when absent; may be NULL) are the two sign changes of x - (x[idx] - key/2), key =
 * the ranking quantity, that FSeries.periods_at_half_max (core.py:963-978) looks up for
 * peak_order = rank + 1: half_hi = the last one inside x[:idx], half_lo = the first one of x[idx:]
 * as an absolute bin (np.where(np.diff(np.signbit(..))), core.py:362); the method returns
 * (period[half_lo], period[half_hi]).  Exactly equal keys rank the lower bin first (upstream's argsort()[::-1]
 * leaves ties to numpy's unstable sort: see INTEGRATION.md).
 * k > 128 runs as launches of 128 ranks (round 5: 64), each ranking what comes AFTER the launch before's last winner in that total
 * order (every launch sweeps the spectra again; measured for 4096 spectra of 5e4 bins: 64 / 128 / 256 ranks 1.1 / 1.5 / 3.0 ms by height, 1.8 / 2.7 / 6.7 ms by prominence - profiles/r06_peaks_timing.txt); it needs idx_out
 * and the ranking key's output (height_out, or prominence_out).  The host methods of FSeries serve any k.
 * pdc_gls_batch_peaks runs the batched periodogram first; the spectra never leave HBM. */
int pdc_peaks_topk(const double *power, int64_t n_curves, int64_t nf, int k, int by_prominence,
                   int64_t *count_out, int64_t *idx_out, double *height_out, double *prominence_out,
                   int64_t *half_lo_out, int64_t *half_hi_out, int device);
int pdc_peaks_topk_dev(int device, void *stream, const double *d_power, int64_t n_curves, int64_t nf,
                       int k, int by_prominence, int64_t *d_count, int64_t *d_idx, double *d_height,
                       double *d_prominence, int64_t *d_half_lo, int64_t *d_half_hi);
int pdc_gls_batch_peaks(const double *t, const double *y, const double *dy, const int64_t *offsets,
                        int64_t n_curves, int shared_t, double f0, double delta, int64_t nf,
                        int fit_mean, int psd, int k, int by_prominence,
                        int64_t *count_out, int64_t *idx_out, double *height_out,
                        double *prominence_out, int64_t *half_lo_out, int64_t *half_hi_out, int device);

/* ---- Phase Dispersion Minimization -----------------------------------------------------------
 * Replaces pool.map(PDM._pdm, periods) (phase.py:128-149, 185-187): theta_out[p] for every trial
 * period, bins phi in [k/m0, (k+nc)/m0) U [0, (k+nc-m0)/m0), m0 = nb*nc, phi = (t/period) % 1
 * with IEEE division and Python modulo; sigma = var(x, ddof=1) is the caller's (phase.py:165). */
int pdc_pdm_scan(const double *t, const double *x, int64_t n,
                 const double *periods, int64_t n_periods, int nb, int nc, double sigma,
                 double *theta_out, int device);
int pdc_pdm_scan_dev(int device, void *stream, const double *d_t, const double *d_x, int64_t n,
                     const double *d_periods, int64_t n_periods, int nb, int nc, double sigma,
                     double *d_theta);

/* The two period searches phase.py:11-15 lists as TODO, on the PDM binning kernel (same phase bins
 * [k/r, (k+1)/r) against the doubles k/r as phase.py:137, same exact phases, one thread per trial
 * period; only the epilogue differs).  The reference has no implementation: parity is pinned to the
 * published formulas, restated in oracle/scan_oracle.py.
 *   Analysis of Variance (Schwarzenberg-Czerny 1989, MNRAS 241, 153, eq. 1-3) over n_bins phase bins:
 *     theta_out[p] = [sum_i n_i (xbar_i - xbar)^2 / (r - 1)] / [sum_i sum_j (x_ij - xbar_i)^2 / (n - r)]
 *   Conditional entropy (Graham et al. 2013, MNRAS 434, 2629, eq. 1) over n_phase x n_mag cells:
 *     entropy_out[p] = sum_ij p(m_j, phi_i) ln(p(phi_i) / p(m_j, phi_i));  mag_bin[i] is the magnitude
 *     bin (0 .. n_mag-1, stored as a double) of sample i, (n_phase + 1) * n_mag <= 191. */
int pdc_aov_scan(const double *t, const double *x, int64_t n, const double *periods, int64_t n_periods,
                 int n_bins, double *theta_out, int device);
int pdc_aov_scan_dev(int device, void *stream, const double *d_t, const double *d_x, int64_t n,
                     const double *d_periods, int64_t n_periods, int n_bins, double *d_theta);
int pdc_cond_entropy_scan(const double *t, const double *mag_bin, int64_t n, const double *periods,
                          int64_t n_periods, int n_phase, int n_mag, double *entropy_out, int device);
int pdc_cond_entropy_scan_dev(int device, void *stream, const double *d_t, const double *d_mag_bin,
                              int64_t n, const double *d_periods, int64_t n_periods, int n_phase,
                              int n_mag, double *d_entropy);

/* The third TODO of phase.py:11-15: the Gregory-Loredo method (Gregory & Loredo 1992, ApJ 398, 146) for
 * ARRIVAL TIMES t (the signal's values are not used): for every trial period the log of
 *     S_m(w) = (1 / 2 pi) Int dphi  m^N n_1! ... n_m! / N!      (their eq. 5.13-5.14 marginalised over the offset;
 * the integrand of the odds ratio O_m1, eq. 5.28), n_j = counts in m phase bins, the offset integral as the
 * mean over n_offsets equally spaced shifts of the bin boundaries; m * n_offsets <= 190.  Counts-only histogram
 * on the PDM binning kernel (same exact phases, same edges f / F).  The reference has no implementation:
 * parity is pinned to the published formula, restated in oracle/scan_oracle.py. */
int pdc_gl_scan(const double *t, int64_t n, const double *periods, int64_t n_periods, int m, int n_offsets,
                double *log_s_out, int device);
int pdc_gl_scan_dev(int device, void *stream, const double *d_t, int64_t n, const double *d_periods,
                    int64_t n_periods, int m, int n_offsets, double *d_log_s);

/* The phase-fold statistics behind one device-resident entry with an EXPLICIT workspace (nothing is
 * cached per stream, nothing is allocated): kind 0 = PDM theta (nb, nc, sigma as pdc_pdm_scan), 1 = AoV
 * (nb = n_bins), 2 = conditional entropy (nb = n_phase, nc = n_mag, v = magnitude bins), 3 = StringLength
 * (v = m; nb, nc, sigma ignored), 4 = Gregory-Loredo (v may be NULL; nb = m * n_offsets, nc = m).  `work` holds at least pdc_phase_work_bytes(kind, n, n_periods, nb, nc)
 * bytes (0 for most PDM shapes: only short period grids over long curves split the samples and need
 * scratch for partial histograms). */
int64_t pdc_phase_work_bytes(int kind, int64_t n, int64_t n_periods, int nb, int nc);
int pdc_phase_scan_dev(int kind, int device, void *stream, const double *d_t, const double *d_v, int64_t n,
                       const double *d_periods, int64_t n_periods, int nb, int nc, double sigma,
                       double *d_out, void *work, int64_t work_bytes);

/* The period grid cut into contiguous slabs over `n_devices` GPUs of this node (one process, one
 * stream per slab; a device may be listed more than once).  Replaces the multiprocessing.Pool
 * fan-out of phase.py:182-186; trial periods are independent, so there is no exchange step.  The
 * per-slot buffers, streams and page-locked staging are kept between calls, keyed by the device list
 * (pdc_release() frees them): a second call of the same size allocates nothing. */
int pdc_pdm_scan_multi(const double *t, const double *x, int64_t n,
                       const double *periods, int64_t n_periods, int nb, int nc, double sigma,
                       double *theta_out, const int *devices, int n_devices);
int pdc_aov_scan_multi(const double *t, const double *x, int64_t n, const double *periods, int64_t n_periods,
                       int n_bins, double *theta_out, const int *devices, int n_devices);
int pdc_cond_entropy_scan_multi(const double *t, const double *mag_bin, int64_t n, const double *periods,
                                int64_t n_periods, int n_phase, int n_mag, double *entropy_out,
                                const int *devices, int n_devices);
int pdc_gl_scan_multi(const double *t, int64_t n, const double *periods, int64_t n_periods, int m, int n_offsets,
                      double *log_s_out, const int *devices, int n_devices);

/* The same fan-out as a persistent plan for callers that scan repeatedly or want the samples resident
 * (bench.py, a survey loop): replaces Pool(cores).map of phase.py:69-70,185-186.
 *   create    streams, events and (for n_max / n_periods_max > 0) buffers per slot; a device may repeat
 *   upload    replicate (t, v) on every slot through page-locked staging (async)
 *   scan      kind as pdc_phase_scan_dev; enqueue per slot: its slab of `periods` H2D, the scan, its slab
 *             of results D2H into page-locked memory; returns without waiting
 *   wait / download   drain; copy the n_periods results of the latest scan to `out`
 *   kernel_ms HIP-event time of the latest scan's kernels, the slowest slot */
int pdc_phase_plan_create(const int *devices, int n_devices, int64_t n_max, int64_t n_periods_max, void **plan);
int pdc_phase_plan_upload(void *plan, const double *t, const double *v, int64_t n);
int pdc_phase_plan_scan(void *plan, int kind, const double *periods, int64_t n_periods, int nb, int nc,
                        double sigma);
int pdc_phase_plan_wait(void *plan);
int pdc_phase_plan_download(void *plan, double *out, int64_t n_periods);
int pdc_phase_plan_kernel_ms(void *plan, float *ms);
int pdc_phase_plan_destroy(void *plan);

/* ---- String Length -----------------------------------------------------------------------------
 * Replaces pool.map(StringLength._stringlength, periods) (phase.py:45-51, 69-70) including the
 * fold ((t - 0)/period) % 1 (core.py:543-544) and the stable sort by phase of the TSeries
 * constructor (core.py:473-477); the closing segment is not phase-wrapped. `m` is the scaled
 * signal of phase.py:65-66.  Samples may come in any order (equal phases keep the order given, as the
 * stable sort does).  Time-ordered samples - what a TSeries holds - are what the kernels are built around: the
 * periods that outlast them need no sort at all (at any size), and from 262 144 samples on a bin's samples are
 * fetched as slices of t / m.  Samples in ANOTHER order, from 262 144 on, are ordered by time on the device first
 * (round 5; a stable radix sort of (t, m), once per call, 0.4 ms at N = 2e6 - csrc/timesort.inc) and take the same
 * kernels: the result is the one for TSeries(t, m), i.e. samples that share a time stamp keep the caller's order, and
 * samples that share a phase at some period without sharing a time stamp are taken in time order there (up to
 * round 4 such samples went through the partition lists and the general kernel - N = 2e6 x 512 periods: 118 ms;
 * now 13.7 against 13.3 ms in order; profiles/r05_sl_shapes.txt).  Non-finite or |t| beyond 1e+-150: no time sort,
 * the lists as before.  Below 262 144 samples the kernels sort any order by phase directly: the periods which outlast
 * the samples are then sorted like any other, and two samples that share a phase WITHOUT sharing a time stamp keep the
 * caller's order there (above: the time order, as the reference's TSeries would have it - core.py:473-477).
 * The `_dev` entry cannot look at the samples from the host: from 262 144 samples on it ALWAYS enqueues the time sort's
 * intake (a 16 n-byte copy of (t, m) and 25 launches that return at once when the device finds the samples in order:
 * 3.19 against 3.15 ms at N = 1e6 x 256 periods) and sizes its workspace for the bin lists; the host entry and the
 * phase plan hold the arrays, check the order and the periods first, and skip both.
 * Workspace (both forms): PDC_WORK_BUDGET_GB=<float> caps it - pdc_stringlength_work_bytes and the scan read the same
 * value -, the host entry also fits it to the device's free memory; smaller batches, same bits; a budget that not
 * even one period fits in is an error whose text names it. */
int pdc_stringlength_scan(const double *t, const double *m, int64_t n,
                          const double *periods, int64_t n_periods,
                          double *ell_out, int device);
int64_t pdc_stringlength_work_bytes(int64_t n, int64_t n_periods);
int pdc_stringlength_scan_dev(int device, void *stream, const double *d_t, const double *d_m,
                              int64_t n, const double *d_periods, int64_t n_periods,
                              double *d_ell, void *work, int64_t work_bytes);

/* As pdc_pdm_scan_multi, for phase.py:69-70. */
int pdc_stringlength_scan_multi(const double *t, const double *m, int64_t n,
                                const double *periods, int64_t n_periods,
                                double *ell_out, const int *devices, int n_devices);

/* Supersmoother period search - the reference names it in one line ("TODO: check out Supersmoother (Reimann
 * 1994)", spectral.py:8) and has no code.  Built from the published algorithm: per trial period the curve is
 * folded and sorted by phase exactly as for StringLength (core.py:543-544, 473-477), Friedman's variable span
 * smoother (Friedman 1984, SLAC PUB-3477: `supsmu` with periodic abscissae, spans 0.05 / 0.2 / 0.5, bass
 * control `alpha` in [0, 10], 0 = off) is fitted to (phase, y), and stat_out[p] = the mean absolute residual
 * about the fit (Reimann 1994): minimal at the period.  Needs n >= 5.  Parity unpinned by the reference; the
 * oracle restates the published Fortran (oracle/scan_oracle.py: supersmoother*) in exact-to-rounding window sums -
 * for phases crowded into a sliver of the cycle (periods thousands of baselines long) the Fortran's own
 * double-precision updating formulas lose the windows' variances; the device follows the oracle there, not them.
 * Samples may come in any order, as for pdc_stringlength_scan (from 262 144 samples on they are ordered by time on the
 * device first).  The device form takes resident inputs and a workspace of pdc_supersmoother_work_bytes(n, n_periods)
 * bytes; PDC_WORK_BUDGET_GB caps it as for pdc_stringlength_scan (smaller sub-batches and pools: the statistic agrees
 * to 1e-12 - the tiles' segment count follows the sub-batch -, a budget nothing fits in is an error that names it). */
int pdc_supersmoother_scan(const double *t, const double *y, int64_t n, const double *periods, int64_t n_periods,
                           double alpha, double *stat_out, int device);
int pdc_supersmoother_scan_multi(const double *t, const double *y, int64_t n, const double *periods, int64_t n_periods,
                                 double alpha, double *stat_out, const int *devices, int n_devices);
int64_t pdc_supersmoother_work_bytes(int64_t n, int64_t n_periods);
int pdc_supersmoother_scan_dev(int device, void *stream, const double *d_t, const double *d_y, int64_t n,
                               const double *d_periods, int64_t n_periods, double alpha, double *d_stat, void *work,
                               int64_t work_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PERIODICITY_HIP_H */
