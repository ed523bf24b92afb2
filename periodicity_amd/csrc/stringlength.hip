// Dworetsky String-Length sweep on gfx950.
//
// Replaces pool.map(StringLength._stringlength, periods)
// (/root/reference/src/periodicity/phase.py:45-51, 69-70) together with the fold
// ((t - 0) / period) % 1 (core.py:543-544) and the stable sort by phase done by the TSeries
// constructor (core.py:473-477).  The polygon is closed with np.roll, so the last->first segment
// is included and is NOT phase-wrapped.
//
// The sort order must be exactly numpy's: two samples a rounding apart in phase swap places and
// change the length by O(|dm|).  Phases are therefore computed with an IEEE division and Python
// modulo and compared as their 64-bit patterns (monotone for phi in [0, 1], NaN last), ties
// broken by sample index (= stable sort of the time-ordered input).
//
// Mapping: one workgroup per trial period (persistent grid, periods strided over workgroups).
// N (phase, index) pairs do not fit in 160 KB of LDS, so the phase axis is cut into ranges:
//   1. histogram of phases over 2048 equal buckets (LDS atomics);
//   2. consecutive buckets are grouped greedily into ranges of <= CAP samples;
//   3. per range: re-scan the samples (recomputing the fold — an fp64 division is far cheaper than
//      a round trip through HBM), compact the members into LDS, bitonic-sort them there, and sum
//      the segments, carrying the last point over to the next range.
// A single bucket holding more than CAP samples (evenly sampled data folded at a commensurate
// period: thousands of identical phases) is sorted in this workgroup's global scratch instead.
#include "pdc_internal.h"

using namespace pdc;

namespace {

constexpr int kBlock = 256;
constexpr int kBuckets = 2048;
constexpr int kCap = 4096;
constexpr int kMaxGrid = 1024;

struct SlArgs {
    const double *t, *m, *periods;
    int64_t n, n_periods;
    double *ell;
    unsigned long long *gkeys;  // [grid][n_pad]
    unsigned *gidx;             // [grid][n_pad]
    int64_t n_pad;
};

__device__ __forceinline__ double fold_phase(double t, double period) {
    const double q = (t - 0.0) / period;   // IEEE division (core.py:544)
    return q - __builtin_floor(q);         // == numpy's float % 1 (exact unless -1 < q < 0)
}

__device__ __forceinline__ int bucket_of(double phi) {
    // monotone non-decreasing in phi; NaN and phi == 1.0 land in the last bucket
    const double u = phi * (double)kBuckets;
    int b = (u >= 0.0) ? (int)(u < (double)kBuckets ? u : (double)(kBuckets - 1)) : 0;
    return (phi != phi) ? kBuckets - 1 : b;
}

// Ascending bitonic sort of P (power of two) (key, index) pairs by (key, index).
template <typename KeyPtr, typename IdxPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr K, IdxPtr I, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int c = threadIdx.x; c < (P >> 1); c += kBlock) {
                const int i = ((c & ~(j - 1)) << 1) | (c & (j - 1));
                const int l = i | j;
                const unsigned long long ka = K[i], kb = K[l];
                const unsigned ia = I[i], ib = I[l];
                const bool up = (i & k) == 0;
                const bool a_gt_b = ka > kb || (ka == kb && ia > ib);
                if (a_gt_b == up) {
                    K[i] = kb;
                    K[l] = ka;
                    I[i] = ib;
                    I[l] = ia;
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(kBlock) void sl_scan_kernel(SlArgs a) {
    __shared__ unsigned hist[kBuckets];
    __shared__ unsigned long long keys[kCap];
    __shared__ unsigned idxs[kCap];
    __shared__ int s_hi, s_cnt;
    __shared__ unsigned s_fill;
    __shared__ double red[kBlock / 64];
    const int tid = threadIdx.x;
    unsigned long long *gk = a.gkeys + (int64_t)blockIdx.x * a.n_pad;
    unsigned *gi = a.gidx + (int64_t)blockIdx.x * a.n_pad;

    for (int64_t p = blockIdx.x; p < a.n_periods; p += gridDim.x) {
        const double period = a.periods[p];
        for (int b = tid; b < kBuckets; b += kBlock) hist[b] = 0u;
        __syncthreads();
        for (int64_t i = tid; i < a.n; i += kBlock)
            atomicAdd(&hist[bucket_of(fold_phase(a.t[i], period))], 1u);
        __syncthreads();

        double total = 0.0;                       // this thread's share of the string length
        bool have_prev = false;                   // carry = last point of the previous range
        double prev_phi = 0.0, prev_m = 0.0, first_phi = 0.0, first_m = 0.0;
        int lo = 0;
        while (lo < kBuckets) {
            if (tid == 0) {
                int hi = lo;
                unsigned cnt = 0;
                do {
                    cnt += hist[hi];
                    ++hi;
                } while (hi < kBuckets && cnt + hist[hi] <= (unsigned)kCap);
                s_hi = hi;
                s_cnt = (int)cnt;
                s_fill = 0u;
            }
            __syncthreads();
            const int hi = s_hi, cnt = s_cnt;
            if (cnt > 0) {
                int P = 2;
                while (P < cnt) P <<= 1;
                const bool in_lds = cnt <= kCap;
                // compact the members of [lo, hi) (arrival order is irrelevant: the sort key
                // (phase bits, index) is a total order)
                for (int64_t i = tid; i < a.n; i += kBlock) {
                    const double phi = fold_phase(a.t[i], period);
                    const int b = bucket_of(phi);
                    if (b >= lo && b < hi) {
                        const unsigned slot = atomicAdd(&s_fill, 1u);
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(phi);
                        if (in_lds) {
                            keys[slot] = bits;
                            idxs[slot] = (unsigned)i;
                        } else {
                            gk[slot] = bits;
                            gi[slot] = (unsigned)i;
                        }
                    }
                }
                for (int s = cnt + tid; s < P; s += kBlock) {
                    if (in_lds) {
                        keys[s] = ~0ull;
                        idxs[s] = ~0u;
                    } else {
                        gk[s] = ~0ull;
                        gi[s] = ~0u;
                    }
                }
                __syncthreads();
                if (in_lds) {
                    bitonic_sort(keys, idxs, P);
                } else {
                    bitonic_sort(gk, gi, P);
                }
                // segments inside the range + the link from the previous range
                for (int j = tid; j < cnt; j += kBlock) {
                    const unsigned long long kj = in_lds ? keys[j] : gk[j];
                    const unsigned ij = in_lds ? idxs[j] : gi[j];
                    const double phi = __longlong_as_double((long long)kj);
                    const double mm = a.m[ij];
                    if (j > 0) {
                        const unsigned long long kp = in_lds ? keys[j - 1] : gk[j - 1];
                        const unsigned ip = in_lds ? idxs[j - 1] : gi[j - 1];
                        total += hypot(mm - a.m[ip], phi - __longlong_as_double((long long)kp));
                    } else if (have_prev) {
                        total += hypot(mm - prev_m, phi - prev_phi);
                    }
                }
                // every thread tracks the carry (uniform values)
                {
                    const unsigned long long k0 = in_lds ? keys[0] : gk[0];
                    const unsigned i0 = in_lds ? idxs[0] : gi[0];
                    const unsigned long long k1 = in_lds ? keys[cnt - 1] : gk[cnt - 1];
                    const unsigned i1 = in_lds ? idxs[cnt - 1] : gi[cnt - 1];
                    if (!have_prev) {
                        first_phi = __longlong_as_double((long long)k0);
                        first_m = a.m[i0];
                    }
                    prev_phi = __longlong_as_double((long long)k1);
                    prev_m = a.m[i1];
                    have_prev = true;
                }
            }
            __syncthreads();  // keys/idxs/s_* are reused by the next range
            lo = hi;
        }
        // closing segment of np.roll(-1): first minus last, no phase wrap (phase.py:50)
        if (tid == 0 && have_prev) total += hypot(first_m - prev_m, first_phi - prev_phi);
        total = wave_sum(total);
        if ((tid & 63) == 0) red[tid >> 6] = total;
        __syncthreads();
        if (tid == 0) a.ell[p] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}

int64_t pad_pow2(int64_t n) {
    int64_t p = 2;
    while (p < n) p <<= 1;
    return p;
}

int64_t grid_for(int64_t n_periods) { return n_periods < kMaxGrid ? n_periods : kMaxGrid; }

}  // namespace

extern "C" {

int64_t pdc_stringlength_work_bytes(int64_t n, int64_t n_periods) {
    if (n < 0 || n_periods < 0) return -1;
    return grid_for(n_periods > 0 ? n_periods : 1) * pad_pow2(n) * 12 + 512;
}

int pdc_stringlength_scan_dev(int device, void *stream, const double *d_t, const double *d_m,
                              int64_t n, const double *d_periods, int64_t n_periods, double *d_ell,
                              void *work, int64_t work_bytes) {
    PDC_REQUIRE(d_t && d_m && (d_periods || n_periods == 0) && (d_ell || n_periods == 0),
                "stringlength: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "stringlength: negative size");
    PDC_REQUIRE(n < ((int64_t)1 << 31), "stringlength: at most 2^31-1 samples");
    PDC_REQUIRE(work && work_bytes >= pdc_stringlength_work_bytes(n, n_periods),
                "stringlength: workspace too small");
    if (n_periods == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    const int64_t grid = grid_for(n_periods);
    SlArgs a;
    a.t = d_t;
    a.m = d_m;
    a.periods = d_periods;
    a.n = n;
    a.n_periods = n_periods;
    a.ell = d_ell;
    a.n_pad = pad_pow2(n);
    a.gkeys = reinterpret_cast<unsigned long long *>(work);
    a.gidx = reinterpret_cast<unsigned *>(a.gkeys + grid * a.n_pad);
    hipLaunchKernelGGL(sl_scan_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}

int pdc_stringlength_scan(const double *t, const double *m, int64_t n, const double *periods,
                          int64_t n_periods, double *ell_out, int device) {
    PDC_REQUIRE(t && m && (periods || n_periods == 0) && (ell_out || n_periods == 0),
                "stringlength: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "stringlength: negative size");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    const int64_t wb = pdc_stringlength_work_bytes(n, n_periods);
    void *d_t, *d_m, *d_p, *d_e, *d_w;
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_m));
    PDC_TRY(cached(device, SLOT_IN2, n_periods * 8, &d_p));
    PDC_TRY(cached(device, SLOT_OUT0, n_periods * 8, &d_e));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_w));
    hipStream_t st = nullptr;
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_m, m, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_p, periods, n_periods * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(pdc_stringlength_scan_dev(device, st, (double *)d_t, (double *)d_m, n, (double *)d_p,
                                      n_periods, (double *)d_e, d_w, wb));
    PDC_HIP(hipMemcpyAsync(ell_out, d_e, n_periods * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

}  // extern "C"
