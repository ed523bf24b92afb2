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
#include <type_traits>

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

// ---- top-k peaks with prominences and half-maximum crossings ------------------------------------------
// For every spectrum of a batch resident in HBM: all find_peaks() maxima (core.py:283-317) with their
// scipy prominences (scipy.signal.peak_prominences, wlen=None: walk left and right from the peak while
// x[i] <= x[peak], remember the lowest sample on each side, prominence = x[peak] - max of the two),
// ranked by height (psort_by_peak, core.py:944-946) or by prominence (psort_by_prominence :948-950,
// period_at_highest_prominence :957-961); only the first k <= 8 come back, together with the two
// sign changes of x - (x[peak] - height/2) that periods_at_half_max (:963-978) looks up: the last
// one left of the peak and the first one from the peak rightwards (np.diff(np.signbit(..)), core.py:362).
// Equal heights / prominences rank the lower bin first (numpy's argsort leaves that order open).
//
// One workgroup per spectrum.  The walks are what could cost O(nf) per peak: per-block minima and
// maxima in LDS let a walk hop over every block that cannot stop it (a block holding a NaN never
// hops: a NaN stops a walk, as x[i] <= x[peak] is false).
constexpr int kPkBlock = 256;
constexpr int kPkMaxBlocks = 4096;   // LDS: two doubles per block
constexpr int kPkMaxK = 8;

struct PeakArgs {
    const double *power;
    int64_t nf;
    int k, by_prominence, blk_shift;
    int64_t nblk, tile;
    long long *count, *idx, *half_lo, *half_hi;
    double *height, *prom;
};

struct Cand {
    double key;
    long long idx;
};

__device__ __forceinline__ bool cand_before(double ka, long long ia, double kb, long long ib) {
    return ib < 0 || (ia >= 0 && (ka > kb || (ka == kb && ia < ib)));
}

template <int K>
__global__ __launch_bounds__(kPkBlock) void peaks_topk_kernel(PeakArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *bmin = reinterpret_cast<double *>(lds_raw);   // [nblk]
    double *bmax = bmin + a.nblk;                         // [nblk]
    double *tile = bmax + a.nblk + 1;                     // [-1 .. a.tile]: the part of the spectrum being scanned + halo
    int *plist = reinterpret_cast<int *>(tile + a.tile + 1);  // [a.tile / 2 + 2] peaks of the tile (bin - t0)
    __shared__ int s_npk;
    __shared__ double red_k[kPkBlock / 64];
    __shared__ long long red_i[kPkBlock / 64];
    __shared__ int red_t[kPkBlock / 64];
    __shared__ long long s_count[kPkBlock / 64];
    __shared__ double win_key[K], win_h[K], win_p[K];
    __shared__ long long win_idx[K];
    __shared__ long long s_found;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *x = a.power + (int64_t)blockIdx.x * a.nf;
    const int64_t nf = a.nf;
    const int sh = a.blk_shift;
    const int64_t blk = (int64_t)1 << sh;
    const double inf = __builtin_inf();

    // ---- A: per-block minimum / maximum (a wave per block, coalesced) -----------------------------
    for (int64_t b = wave; b < a.nblk; b += kPkBlock / 64) {
        double mn = inf, mx = -inf;
        bool nan = false;
        const int64_t e = (b + 1) * blk < nf ? (b + 1) * blk : nf;
        for (int64_t i = b * blk + lane; i < e; i += 64) {
            const double v = x[i];
            nan = nan || v != v;
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double omn = __shfl_xor(mn, o, 64), omx = __shfl_xor(mx, o, 64);
            mn = omn < mn ? omn : mn;
            mx = omx > mx ? omx : mx;
        }
        if (__any(nan)) mx = inf;
        if (lane == 0) {
            bmin[b] = mn;
            bmax[b] = mx;
        }
    }
    __syncthreads();

    // The spectrum is scanned tile by tile through LDS (tiles are whole blocks): a dependent chain of
    // global loads costs an L2 round trip per step once 32 waves of walkers thrash the 32 KB L1, an LDS
    // read ~20x less, and nearly every step of nearly every walk stays inside the tile of its peak.
    int64_t t0 = 0, t1 = 0;

    // lowest sample met walking from `from` (a bin of the current tile) in direction dir (+1 / -1) while
    // x[i] <= h (scipy's loop).  Blocks never straddle tiles, so every stepping loop reads either LDS
    // or global memory, never a per-lane mix.
    auto walk = [&](int64_t from, int dir, double h, auto start_in_tile) -> double {
        double low = h;
        int64_t i = from;
        // inside the starting block (in the tile, except for a flat top that ran past the tile's end)
        const int64_t b0 = from >> sh;
        const int64_t edge = dir > 0 ? (((b0 + 1) << sh) < nf ? ((b0 + 1) << sh) : nf) : (b0 << sh) - 1;
        while (i != edge) {
            const double v = decltype(start_in_tile)::value ? tile[i - t0] : x[i];
            if (!(v <= h)) return low;
            low = v < low ? v : low;
            i += dir;
        }
        // whole blocks that cannot stop the walk
        int64_t b = b0 + dir;
        while (b >= 0 && b < a.nblk && bmax[b] <= h) {
            low = bmin[b] < low ? bmin[b] : low;
            b += dir;
        }
        if (b < 0 || b >= a.nblk) return low;   // reached the border of the signal
        i = dir > 0 ? b << sh : (((b + 1) << sh) < nf ? ((b + 1) << sh) : nf) - 1;
        if (i >= t0 && i < t1) {
            for (;;) {                           // this block holds a sample > h (or a NaN): the walk ends in it
                const double v = tile[i - t0];
                if (!(v <= h)) return low;
                low = v < low ? v : low;
                i += dir;
            }
        }
        for (;;) {                               // ... the same in a block of another tile
            const double v = x[i];
            if (!(v <= h)) return low;
            low = v < low ? v : low;
            i += dir;
        }
    };

    // ---- B: peaks of this thread's bins, their prominences, its own top K --------------------------
    Cand best[K];
    double best_h[K], best_p[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        best[q].key = 0.0;
        best[q].idx = -1;
        best_h[q] = best_p[q] = 0.0;
    }
    long long mine = 0;
    for (t0 = 0; t0 < nf; t0 += a.tile) {
    t1 = t0 + a.tile < nf ? t0 + a.tile : nf;
    __syncthreads();   // everyone is done with the previous tile
    for (int64_t i = t0 - 1 + tid; i <= t1; i += kPkBlock) tile[i - t0] = (i >= 0 && i < nf) ? x[i] : 0.0;
    __syncthreads();
    if (tid == 0) s_npk = 0;
    __syncthreads();
    // (1) the tile's maxima, compacted into a list: with lanes = bins only every eighth lane would hold a
    // peak and every 64-bin window would cost the longest walk among its few peaks
    for (int64_t i = (t0 > 0 ? t0 : 1) + tid; i < t1 && i < nf - 1; i += kPkBlock) {
        const double v = tile[i - t0];
        if (!(tile[i - 1 - t0] < v)) continue;
        int64_t ahead = i + 1;   // (a flat top may run past the tile and its halo: global reads from there on)
        while (ahead < nf - 1 && (ahead <= t1 ? tile[ahead - t0] : x[ahead]) == v) ++ahead;
        if (!((ahead <= t1 ? tile[ahead - t0] : x[ahead]) < v)) continue;
        plist[atomicAdd(&s_npk, 1)] = (int)((i + ahead - 1) / 2 - t0);
    }
    __syncthreads();
    const int npk = s_npk;
    mine += tid == 0 ? npk : 0;
    // (2) lanes = consecutive entries of the list: prominence walks, per-thread top K
    for (int pi = tid; pi < npk; pi += kPkBlock) {
        const int64_t mid = t0 + plist[pi];
        const bool inside = mid < t1;
        const double v = inside ? tile[mid - t0] : x[mid];
        double lo, hi;
        if (inside) {
            lo = walk(mid, -1, v, std::true_type{});
            hi = walk(mid, +1, v, std::true_type{});
        } else {
            lo = walk(mid, -1, v, std::false_type{});
            hi = walk(mid, +1, v, std::false_type{});
        }
        const double prom = v - (lo > hi ? lo : hi);
        const double ckey = a.by_prominence ? prom : v;
        // insertion into the sorted list (descending key, then bin): slot = number of entries that stay
        // ahead of the newcomer; the tail moves down one place
        int pos = 0;
#pragma unroll
        for (int q = 0; q < K; ++q) pos += cand_before(ckey, (long long)mid, best[q].key, best[q].idx) ? 0 : 1;
#pragma unroll
        for (int q = K - 1; q > 0; --q) {
            if (q > pos) {
                best[q] = best[q - 1];
                best_h[q] = best_h[q - 1];
                best_p[q] = best_p[q - 1];
            }
        }
#pragma unroll
        for (int q = 0; q < K; ++q) {
            if (q == pos) {
                best[q].key = ckey;
                best[q].idx = mid;
                best_h[q] = v;
                best_p[q] = prom;
            }
        }
    }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (lane == 0) s_count[wave] = mine;

    // ---- C: the K best of the workgroup: K rounds of "best head of any thread" --------------------
    int head = 0;
    for (int round = 0; round < K; ++round) {
        double hk = 0.0, hh = 0.0, hp = 0.0;
        long long hi_ = -1;
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (q == head) {
                hk = best[q].key;
                hi_ = best[q].idx;
                hh = best_h[q];
                hp = best_p[q];
            }
        double wk = hk;
        long long wi = hi_;
        int wt = tid;
        for (int o = 32; o > 0; o >>= 1) {
            const double ok = __shfl_down(wk, o, 64);
            const long long oi = __shfl_down(wi, o, 64);
            const int ot = __shfl_down(wt, o, 64);
            if (cand_before(ok, oi, wk, wi)) {
                wk = ok;
                wi = oi;
                wt = ot;
            }
        }
        __syncthreads();
        if (lane == 0) {
            red_k[wave] = wk;
            red_i[wave] = wi;
            red_t[wave] = wt;
        }
        __syncthreads();
        wk = red_k[0];
        wi = red_i[0];
        wt = red_t[0];
        for (int w = 1; w < kPkBlock / 64; ++w)
            if (cand_before(red_k[w], red_i[w], wk, wi)) {
                wk = red_k[w];
                wi = red_i[w];
                wt = red_t[w];
            }
        if (wi >= 0 && tid == wt) {   // the winner publishes its candidate and moves on
            win_key[round] = hk;
            win_idx[round] = hi_;
            win_h[round] = hh;
            win_p[round] = hp;
            ++head;
        }
        if (wi < 0 && tid == 0) {
            win_key[round] = __builtin_nan("");
            win_idx[round] = -1;
            win_h[round] = win_p[round] = __builtin_nan("");
        }
    }
    __syncthreads();
    const int64_t ob = (int64_t)blockIdx.x * a.k;
    if (tid == 0) {
        long long total = 0;
        for (int w = 0; w < kPkBlock / 64; ++w) total += s_count[w];
        if (a.count) a.count[blockIdx.x] = total;
    }
    if (tid < a.k) {
        const bool ok = tid < K;
        if (a.idx) a.idx[ob + tid] = ok ? win_idx[tid] : -1;
        if (a.height) a.height[ob + tid] = ok ? win_h[tid] : __builtin_nan("");
        if (a.prom) a.prom[ob + tid] = ok ? win_p[tid] : __builtin_nan("");
    }

    // ---- D: half-maximum crossings of every ranked peak (periods_at_half_max) -------------------
    if (!a.half_lo && !a.half_hi) return;
    for (int r = 0; r < a.k && r < K; ++r) {
        const long long idmax = win_idx[r];
        long long lo_abs = -1, hi_abs = -1;
        if (idmax >= 0) {
            const double half = x[idmax] - win_key[r] / 2;     // core.py:972 (height or prominence)
            auto flips = [&](int64_t i) {                       // signbit(x[i]-half) != signbit(x[i+1]-half)
                return (__double_as_longlong(x[i] - half) < 0) != (__double_as_longlong(x[i + 1] - half) < 0);
            };
            // last sign change inside x[:idmax]: pairs (i, i+1), i+1 <= idmax-1; search outwards in chunks
            for (int64_t top = idmax - 2; top >= 0; top -= kPkBlock) {
                const int64_t i = top - tid;
                const bool f = i >= 0 && flips(i);
                __syncthreads();
                if (tid == 0) s_found = -1;
                __syncthreads();
                if (f) atomicMax(&s_found, (long long)i);
                __syncthreads();
                const long long got = s_found;
                if (got >= 0) {   // (workgroup-uniform)
                    hi_abs = got;
                    break;
                }
            }
            // first sign change from the peak rightwards: pairs (idmax+i, idmax+i+1)
            for (int64_t base = idmax; base < nf - 1; base += kPkBlock) {
                const int64_t i = base + tid;
                const bool f = i < nf - 1 && flips(i);
                __syncthreads();
                if (tid == 0) s_found = nf;
                __syncthreads();
                if (f) atomicMin(&s_found, (long long)i);
                __syncthreads();
                const long long got = s_found;
                if (got < nf) {
                    lo_abs = got;
                    break;
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (a.half_lo) a.half_lo[ob + r] = lo_abs;
            if (a.half_hi) a.half_hi[ob + r] = hi_abs;
        }
    }
    if (tid == 0)
        for (int r = K; r < a.k; ++r) {
            if (a.half_lo) a.half_lo[ob + r] = -1;
            if (a.half_hi) a.half_hi[ob + r] = -1;
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

namespace {

int launch_topk(hipStream_t st, PeakArgs a, int64_t n_curves) {
    int sh = 6;
    while (((a.nf + ((int64_t)1 << sh) - 1) >> sh) > kPkMaxBlocks) ++sh;
    a.blk_shift = sh;
    a.nblk = (a.nf + ((int64_t)1 << sh) - 1) >> sh;
    // tile of the spectrum staged in LDS: whole blocks, ~2048 bins (16 KB + halo: five workgroups per CU
    // at 5e4 bins), or the whole row
    const int64_t blk = (int64_t)1 << sh;
    a.tile = blk > 2048 ? blk : 2048;
    if (a.tile > a.nf) a.tile = ((a.nf > 0 ? a.nf : 1) + blk - 1) / blk * blk;
    const size_t lds = (size_t)(a.nblk > 0 ? a.nblk : 1) * 16 + (size_t)(a.tile + 2) * 8 + (size_t)(a.tile / 2 + 4) * 4 + 16;
    static const hipError_t attr[3] = {
        hipFuncSetAttribute((const void *)peaks_topk_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024),
        hipFuncSetAttribute((const void *)peaks_topk_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024),
        hipFuncSetAttribute((const void *)peaks_topk_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)};
    for (hipError_t e : attr) PDC_HIP(e);
    const dim3 grid((unsigned)n_curves), block(kPkBlock);
    if (a.k <= 1) hipLaunchKernelGGL((peaks_topk_kernel<1>), grid, block, lds, st, a);
    else if (a.k <= 4) hipLaunchKernelGGL((peaks_topk_kernel<4>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((peaks_topk_kernel<8>), grid, block, lds, st, a);
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}

}  // namespace

int pdc_peaks_topk_dev(int device, void *stream, const double *d_power, int64_t n_curves, int64_t nf,
                       int k, int by_prominence, int64_t *d_count, int64_t *d_idx, double *d_height,
                       double *d_prominence, int64_t *d_half_lo, int64_t *d_half_hi) {
    PDC_REQUIRE(d_power || n_curves * nf == 0, "peaks_topk: power is NULL");
    PDC_REQUIRE(k >= 1 && k <= kPkMaxK, "peaks_topk: k must be 1..%d", kPkMaxK);
    PDC_REQUIRE(n_curves >= 0 && nf >= 0 && n_curves < ((int64_t)1 << 31), "peaks_topk: bad size");
    PDC_REQUIRE(d_count || d_idx || d_height || d_prominence || d_half_lo || d_half_hi,
                "peaks_topk: no output requested");
    if (n_curves == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    PeakArgs a;
    a.power = d_power;
    a.nf = nf;
    a.k = k;
    a.by_prominence = by_prominence ? 1 : 0;
    a.count = (long long *)d_count;
    a.idx = (long long *)d_idx;
    a.half_lo = (long long *)d_half_lo;
    a.half_hi = (long long *)d_half_hi;
    a.height = d_height;
    a.prom = d_prominence;
    return launch_topk((hipStream_t)stream, a, n_curves);
}

namespace {

// device slots of the k-wide outputs + D2H; shared by the two host entry points
int topk_outputs(int device, hipStream_t st, const double *d_pow, int64_t n_curves, int64_t nf, int k,
                 int by_prominence, int64_t *count, int64_t *idx, double *height, double *prom,
                 int64_t *half_lo, int64_t *half_hi) {
    const int64_t nk = n_curves * k;
    // one cached block: count | idx | lo | hi | height | prom
    void *blockp;
    PDC_TRY(cached(device, SLOT_OUT1, (n_curves + 5 * nk) * 8, &blockp));
    int64_t *d_count = (int64_t *)blockp, *d_idx = d_count + n_curves, *d_lo = d_idx + nk, *d_hi = d_lo + nk;
    double *d_h = (double *)(d_hi + nk), *d_p = d_h + nk;
    PDC_TRY(pdc_peaks_topk_dev(device, st, d_pow, n_curves, nf, k, by_prominence, d_count, d_idx, d_h, d_p,
                               (half_lo || half_hi) ? d_lo : nullptr, (half_lo || half_hi) ? d_hi : nullptr));
    if (count) PDC_HIP(hipMemcpyAsync(count, d_count, n_curves * 8, hipMemcpyDeviceToHost, st));
    if (idx) PDC_HIP(hipMemcpyAsync(idx, d_idx, nk * 8, hipMemcpyDeviceToHost, st));
    if (height) PDC_HIP(hipMemcpyAsync(height, d_h, nk * 8, hipMemcpyDeviceToHost, st));
    if (prom) PDC_HIP(hipMemcpyAsync(prom, d_p, nk * 8, hipMemcpyDeviceToHost, st));
    if (half_lo) PDC_HIP(hipMemcpyAsync(half_lo, d_lo, nk * 8, hipMemcpyDeviceToHost, st));
    if (half_hi) PDC_HIP(hipMemcpyAsync(half_hi, d_hi, nk * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

}  // namespace

int pdc_peaks_topk(const double *power, int64_t n_curves, int64_t nf, int k, int by_prominence,
                   int64_t *count_out, int64_t *idx_out, double *height_out, double *prominence_out,
                   int64_t *half_lo_out, int64_t *half_hi_out, int device) {
    PDC_REQUIRE(power || n_curves * nf == 0, "peaks_topk: power is NULL");
    PDC_REQUIRE(k >= 1 && k <= kPkMaxK, "peaks_topk: k must be 1..%d", kPkMaxK);
    PDC_REQUIRE(n_curves >= 0 && nf >= 0, "peaks_topk: negative size");
    PDC_REQUIRE(count_out || idx_out || height_out || prominence_out || half_lo_out || half_hi_out,
                "peaks_topk: no output requested");
    if (n_curves == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    void *d_p;
    PDC_TRY(cached(device, SLOT_OUT0, n_curves * nf * 8, &d_p));
    hipStream_t st = nullptr;
    PDC_HIP(hipMemcpyAsync(d_p, power, n_curves * nf * 8, hipMemcpyHostToDevice, st));
    return topk_outputs(device, st, (double *)d_p, n_curves, nf, k, by_prominence, count_out, idx_out,
                        height_out, prominence_out, half_lo_out, half_hi_out);
}

// Batched periodograms reduced on the device to their k highest (or most prominent) peaks with
// prominences and half-maximum crossings: 4096 x 5e4 spectra stay in HBM, O(B k) values come back.
int pdc_gls_batch_peaks(const double *t, const double *y, const double *dy, const int64_t *offsets,
                        int64_t n_curves, int shared_t, double f0, double delta, int64_t nf, int fit_mean,
                        int psd, int k, int by_prominence, int64_t *count_out, int64_t *idx_out,
                        double *height_out, double *prominence_out, int64_t *half_lo_out,
                        int64_t *half_hi_out, int device) {
    PDC_REQUIRE(t && y && offsets, "gls_batch_peaks: NULL argument");
    PDC_REQUIRE(n_curves >= 1 && nf >= 0, "gls_batch_peaks: bad size");
    PDC_REQUIRE(k >= 1 && k <= kPkMaxK, "gls_batch_peaks: k must be 1..%d", kPkMaxK);
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
    void *d_t, *d_y, *d_dy = nullptr, *d_off, *d_pow, *d_work;
    PDC_TRY(cached(device, SLOT_IN0, n_t * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n_total * 8, &d_y));
    if (dy) PDC_TRY(cached(device, SLOT_IN2, n_total * 8, &d_dy));
    PDC_TRY(cached(device, SLOT_IN3, (n_curves + 1) * 8, &d_off));
    PDC_TRY(cached(device, SLOT_OUT0, n_curves * nf * 8, &d_pow));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_work));
    hipStream_t st = nullptr;
    PDC_HIP(hipMemcpyAsync(d_t, t, n_t * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_y, y, n_total * 8, hipMemcpyHostToDevice, st));
    if (dy) PDC_HIP(hipMemcpyAsync(d_dy, dy, n_total * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_off, offsets, (n_curves + 1) * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(pdc_gls_scan_dev(device, st, (double *)d_t, (double *)d_y, (double *)d_dy, (int64_t *)d_off,
                             n_total, n_curves, shared_t, f0, delta, 0, nf, fit_mean, psd,
                             (double *)d_pow, nullptr, nullptr, d_work, wb));
    return topk_outputs(device, st, (double *)d_pow, n_curves, nf, k, by_prominence, count_out, idx_out,
                        height_out, prominence_out, half_lo_out, half_hi_out);
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
