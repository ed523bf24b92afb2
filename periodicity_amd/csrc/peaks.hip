// Device-side peak picking for FSeries consumers.
//
// Replaces, for a batch of spectra resident in HBM, FSeries.period_at_highest_peak
// (/root/reference/src/periodicity/core.py:952-955): find_peaks (core.py:283-317) =
// scipy.signal.find_peaks(values, prominence=0.0) = scipy's _local_maxima_1d, then the NaN-aware
// maximum over those peaks (core.py:202-220, first maximum on ties).
//
// scipy's rule, restated: i is the left edge of a peak when x[i-1] < x[i]; walk right over the
// flat top while x[j] == x[i]; it is a peak only if the sample after the flat top is lower, and
// the reported index is the midpoint (left + right) // 2 of the flat top.  The first and last
// sample are never peaks; comparisons with NaN are false, so NaN is never part of one.
#include "pdc_internal.h"

using namespace pdc;

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void highest_peak_kernel(const double *power, int64_t nf,
                                                              int64_t *idx_out, double *val_out) {
    __shared__ double red_v[kBlock / 64];
    __shared__ long long red_i[kBlock / 64];
    const double *x = power + (int64_t)blockIdx.x * nf;
    double best = 0.0;
    long long best_i = -1;
    for (int64_t i = 1 + threadIdx.x; i < nf - 1; i += kBlock) {  // ascending per thread
        const double v = x[i];
        if (!(x[i - 1] < v)) continue;
        int64_t ahead = i + 1;
        while (ahead < nf - 1 && x[ahead] == v) ++ahead;
        if (!(x[ahead] < v)) continue;
        const long long mid = (long long)((i + ahead - 1) / 2);
        if (best_i < 0 || v > best) {
            best = v;
            best_i = mid;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(best, o, 64);
        const long long oi = __shfl_down(best_i, o, 64);
        if (oi >= 0 && (best_i < 0 || ov > best || (ov == best && oi < best_i))) {
            best = ov;
            best_i = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        red_v[threadIdx.x >> 6] = best;
        red_i[threadIdx.x >> 6] = best_i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            if (red_i[w] >= 0 && (best_i < 0 || red_v[w] > best || (red_v[w] == best && red_i[w] < best_i))) {
                best = red_v[w];
                best_i = red_i[w];
            }
        }
        if (idx_out) idx_out[blockIdx.x] = best_i;
        if (val_out) val_out[blockIdx.x] = best_i >= 0 ? best : __builtin_nan("");
    }
}

}  // namespace

extern "C" {

int pdc_highest_peak_dev(int device, void *stream, const double *d_power, int64_t n_curves,
                         int64_t nf, int64_t *d_idx, double *d_val) {
    PDC_REQUIRE(d_power || n_curves * nf == 0, "highest_peak: power is NULL");
    PDC_REQUIRE(d_idx || d_val, "highest_peak: no output requested");
    PDC_REQUIRE(n_curves >= 0 && nf >= 0 && n_curves < ((int64_t)1 << 31), "highest_peak: bad size");
    if (n_curves == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    hipLaunchKernelGGL(highest_peak_kernel, dim3((unsigned)n_curves), dim3(kBlock), 0,
                       (hipStream_t)stream, d_power, nf, d_idx, d_val);
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}

int pdc_highest_peak(const double *power, int64_t n_curves, int64_t nf, int64_t *idx_out,
                     double *val_out, int device) {
    PDC_REQUIRE(power || n_curves * nf == 0, "highest_peak: power is NULL");
    PDC_REQUIRE(idx_out || val_out, "highest_peak: no output requested");
    PDC_REQUIRE(n_curves >= 0 && nf >= 0, "highest_peak: negative size");
    if (n_curves == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    void *d_p, *d_i, *d_v;
    PDC_TRY(cached(device, SLOT_OUT0, n_curves * nf * 8, &d_p));
    PDC_TRY(cached(device, SLOT_OUT1, n_curves * 8, &d_i));
    PDC_TRY(cached(device, SLOT_OUT2, n_curves * 8, &d_v));
    hipStream_t st = nullptr;
    PDC_HIP(hipMemcpyAsync(d_p, power, n_curves * nf * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(pdc_highest_peak_dev(device, st, (double *)d_p, n_curves, nf, (int64_t *)d_i, (double *)d_v));
    if (idx_out) PDC_HIP(hipMemcpyAsync(idx_out, d_i, n_curves * 8, hipMemcpyDeviceToHost, st));
    if (val_out) PDC_HIP(hipMemcpyAsync(val_out, d_v, n_curves * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

// Batched periodograms reduced on the device to the highest peak of each (index into the grid and
// power there): the spectra never leave HBM.
int pdc_gls_batch_highest_peak(const double *t, const double *y, const double *dy,
                               const int64_t *offsets, int64_t n_curves, int shared_t, double f0,
                               double delta, int64_t nf, int fit_mean, int psd, int64_t *idx_out,
                               double *val_out, int device) {
    PDC_REQUIRE(t && y && offsets && (idx_out || val_out), "gls_batch_highest_peak: NULL argument");
    PDC_REQUIRE(n_curves >= 1 && nf >= 0, "gls_batch_highest_peak: bad size");
    for (int64_t b = 0; b < n_curves; ++b) {
        PDC_REQUIRE(offsets[b + 1] >= offsets[b], "gls: offsets must be non-decreasing");
        PDC_REQUIRE(!shared_t || offsets[b + 1] - offsets[b] == offsets[1] - offsets[0],
                    "gls: with a shared time axis every curve must have the same length");
    }
    PDC_REQUIRE(offsets[0] == 0, "gls: offsets[0] must be 0");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    const int64_t n_total = offsets[n_curves];
    const int64_t n_t = shared_t ? offsets[1] : n_total;
    const int64_t wb = pdc_gls_work_bytes(n_total, n_curves, nf);
    void *d_t, *d_y, *d_dy = nullptr, *d_off, *d_pow, *d_i, *d_v, *d_work;
    PDC_TRY(cached(device, SLOT_IN0, n_t * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n_total * 8, &d_y));
    if (dy) PDC_TRY(cached(device, SLOT_IN2, n_total * 8, &d_dy));
    PDC_TRY(cached(device, SLOT_IN3, (n_curves + 1) * 8, &d_off));
    PDC_TRY(cached(device, SLOT_OUT0, n_curves * nf * 8, &d_pow));
    PDC_TRY(cached(device, SLOT_OUT1, n_curves * 8, &d_i));
    PDC_TRY(cached(device, SLOT_OUT2, n_curves * 8, &d_v));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_work));
    hipStream_t st = nullptr;
    PDC_HIP(hipMemcpyAsync(d_t, t, n_t * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_y, y, n_total * 8, hipMemcpyHostToDevice, st));
    if (dy) PDC_HIP(hipMemcpyAsync(d_dy, dy, n_total * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_off, offsets, (n_curves + 1) * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(pdc_gls_scan_dev(device, st, (double *)d_t, (double *)d_y, (double *)d_dy, (int64_t *)d_off,
                             n_total, n_curves, shared_t, f0, delta, 0, nf, fit_mean, psd,
                             (double *)d_pow, nullptr, nullptr, d_work, wb));
    PDC_TRY(pdc_highest_peak_dev(device, st, (double *)d_pow, n_curves, nf, (int64_t *)d_i, (double *)d_v));
    if (idx_out) PDC_HIP(hipMemcpyAsync(idx_out, d_i, n_curves * 8, hipMemcpyDeviceToHost, st));
    if (val_out) PDC_HIP(hipMemcpyAsync(val_out, d_v, n_curves * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

}  // extern "C"
