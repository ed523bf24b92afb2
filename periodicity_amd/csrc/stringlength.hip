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
// N (phase, index) pairs do not fit in 160 KB of LDS, so the sort is two-level and linear-time:
//   P1  histogram of the phases over 2048 equal coarse buckets (LDS atomics) + exclusive scan;
//   P2  re-fold every sample (an fp64 division is cheaper than keeping phases around) and scatter
//       (phase bits, index) into this workgroup's global scratch, grouped by coarse bucket;
//   P3  consecutive buckets are grouped into ranges of <= 4096 samples (a contiguous slice of the
//       scratch).  Per range: load the slice, rank every element inside one of <= 4096 FINE
//       buckets (LDS atomics; the fine index is a power-of-two refinement of the coarse one, so
//       both are exact and consistent), exclusive scan, place into LDS in fine-bucket order,
//       finish each fine bucket (mean occupancy < 1) with an insertion sort on (bits, index),
//       then sum the segments, carrying the last point over to the next range.
// Clustered phases (evenly sampled data folded at a commensurate period: thousands of identical
// phases) defeat the counting sort; a range whose fullest fine bucket holds more than 24 samples
// is bitonic-sorted instead (in LDS, or in global scratch when one coarse bucket alone exceeds
// 4096 samples).
#include "pdc_internal.h"

#include <cstdlib>

using namespace pdc;

namespace {

constexpr int kBlock = 256;
constexpr int kBuckets = 2048;  // coarse buckets over [0, 1]
constexpr int kCap = 4096;      // samples per range (LDS sort capacity)
constexpr int kFine = 4096;     // fine buckets per range
constexpr int kPer = kCap / kBlock;
constexpr int kInsertMax = 24;  // fullest fine bucket the insertion-sort finish accepts
constexpr int kMaxGrid = 1024;

struct SlArgs {
    const double *t, *m, *periods;
    int64_t n, n_periods;
    double *ell;
    unsigned long long *gkeys;   // [grid][n_pad]   partitioned (phase bits)
    unsigned *gidx;              // [grid][n_pad]   partitioned (sample index)
    unsigned long long *gkeys2;  // [grid][n_pad]   padded copy for the oversized-bucket sort
    unsigned *gidx2;             // [grid][n_pad]
    double *rsum;                // [grid][nr_pad][4]  range summaries (all-LDS kernel)
    int *rcnt;                   // [grid][nr_pad]
    int64_t n_pad, nr_pad;
    int use_lds;   // 1: all-LDS kernel (16-bit indices), 0: global-scratch kernel
};

__device__ __forceinline__ double fold_phase(double t, double period) {
    const double q = (t - 0.0) / period;   // IEEE division (core.py:544)
    return q - __builtin_floor(q);         // == numpy's float % 1 (exact unless -1 < q < 0)
}

// floor(phi * scale) clamped to [0, last]; scale is a power of two, so the product is exact and
// every refinement of the coarse index is consistent with it.  NaN goes last.
__device__ __forceinline__ int scaled_index(double phi, double scale, int last) {
    const double u = phi * scale;
    int b = (u >= 0.0) ? (u < (double)last ? (int)u : last) : 0;
    return (phi != phi) ? last : b;
}

// Exclusive prefix sum of a[0..L) (L <= kScanPer * BLOCK) in place; returns the total.
constexpr int kScanPer = kPer + 1;  // the fine array carries one extra slot for the total
template <int BLOCK>
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned *a, int L, unsigned *wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (L + BLOCK - 1) / BLOCK;
    const int beg = tid * per;
    unsigned local[kScanPer];
    unsigned sum = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
        if (e < per) {
            const int i = beg + e;
            local[e] = i < L ? a[i] : 0u;
            sum += local[e];
        }
    }
    unsigned incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        if (w < wave) base += wave_tot[w];
        total += wave_tot[w];
    }
    unsigned run = base + incl - sum;
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
        if (e < per) {
            const int i = beg + e;
            if (i < L) a[i] = run;
            run += local[e];
        }
    }
    __syncthreads();
    return total;
}

// Ascending bitonic sort of P (power of two) (key, index) pairs by (key, index).
template <int BLOCK, typename IdxT, typename KeyPtr, typename IdxPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr K, IdxPtr I, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int c = threadIdx.x; c < (P >> 1); c += BLOCK) {
                const int i = ((c & ~(j - 1)) << 1) | (c & (j - 1));
                const int l = i | j;
                const unsigned long long ka = K[i], kb = K[l];
                const IdxT ia = I[i], ib = I[l];
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

// Sum of hypot(dm, dphi) over consecutive sorted points j-1 -> j, j in [0, cnt), where point -1
// is the carry (if any).  Strided over the workgroup; the previous point comes from the
// neighbouring lane.
template <int BLOCK, typename KeyPtr, typename IdxPtr>
__device__ __forceinline__ double segment_sum(KeyPtr K, IdxPtr I, int cnt, const double *m,
                                              bool have_prev, double prev_phi, double prev_m) {
    const int lane = threadIdx.x & 63;
    double total = 0.0;
    for (int j0 = 0; j0 < cnt; j0 += BLOCK) {
        const int j = j0 + threadIdx.x;
        const bool live = j < cnt;
        double phi = 0.0, mm = 0.0;
        if (live) {
            phi = __longlong_as_double((long long)K[j]);
            mm = m[I[j]];
        }
        double pphi = __shfl_up(phi, 1, 64);
        double pm = __shfl_up(mm, 1, 64);
        bool ok = live;
        if (lane == 0 && live) {
            if (j > 0) {
                pphi = __longlong_as_double((long long)K[j - 1]);
                pm = m[I[j - 1]];
            } else {
                pphi = prev_phi;
                pm = prev_m;
                ok = have_prev;
            }
        }
        if (ok) total += hypot(mm - pm, phi - pphi);
    }
    return total;
}

__global__ __launch_bounds__(kBlock) void sl_scan_kernel(SlArgs a) {
    __shared__ unsigned hist[kBuckets];           // coarse counts -> starts -> ends
    __shared__ unsigned fine[kFine + 1];          // fine counts -> starts (+ total)
    __shared__ unsigned long long keys[kCap];
    __shared__ unsigned idxs[kCap];
    __shared__ unsigned wave_tot[kBlock / 64];
    __shared__ unsigned s_max;
    __shared__ double red[kBlock / 64];
    const int tid = threadIdx.x;
    unsigned long long *gk = a.gkeys + (int64_t)blockIdx.x * a.n_pad;
    unsigned *gi = a.gidx + (int64_t)blockIdx.x * a.n_pad;
    unsigned long long *gk2 = a.gkeys2 + (int64_t)blockIdx.x * a.n_pad;
    unsigned *gi2 = a.gidx2 + (int64_t)blockIdx.x * a.n_pad;

    for (int64_t p = blockIdx.x; p < a.n_periods; p += gridDim.x) {
        const double period = a.periods[p];
        // ---- P1: coarse histogram -------------------------------------------------------------
        for (int b = tid; b < kBuckets; b += kBlock) hist[b] = 0u;
        __syncthreads();
        for (int64_t i = tid; i < a.n; i += kBlock)
            atomicAdd(&hist[scaled_index(fold_phase(a.t[i], period), (double)kBuckets, kBuckets - 1)], 1u);
        __syncthreads();
        block_exclusive_scan<kBlock>(hist, kBuckets, wave_tot);
        // ---- P2: scatter into the scratch, grouped by coarse bucket ----------------------------
        for (int64_t i = tid; i < a.n; i += kBlock) {
            const double phi = fold_phase(a.t[i], period);
            const unsigned pos = atomicAdd(&hist[scaled_index(phi, (double)kBuckets, kBuckets - 1)], 1u);
            gk[pos] = (unsigned long long)__double_as_longlong(phi);
            gi[pos] = (unsigned)i;
        }
        __syncthreads();  // hist[b] is now the END offset of bucket b; scratch writes are visible

        // ---- P3: ranges --------------------------------------------------------------------------
        double total = 0.0;
        bool have_prev = false;
        double prev_phi = 0.0, prev_m = 0.0, first_phi = 0.0, first_m = 0.0;
        int lo = 0;
        unsigned beg = 0;  // end offset of bucket lo-1
        while (lo < kBuckets && (int64_t)beg < a.n) {
            // largest hi with end(hi-1) - beg <= kCap (uniform binary search); at least lo+1
            int hi;
            {
                int l = lo + 1, r = kBuckets;  // answer in [l, r]
                while (l < r) {
                    const int mid = (l + r + 1) >> 1;
                    if (hist[mid - 1] - beg <= (unsigned)kCap) l = mid; else r = mid - 1;
                }
                hi = l;
            }
            const int cnt = (int)(hist[hi - 1] - beg);
            if (cnt > 0) {
                const unsigned long long *sk = gk + beg;
                const unsigned *si = gi + beg;
                double part;
                double k0_phi, k0_m, k1_phi, k1_m;
                if (cnt > kCap) {
                    // one coarse bucket alone overflows LDS: padded copy + bitonic sort in scratch
                    int P = 2;
                    while (P < cnt) P <<= 1;
                    for (int s = tid; s < P; s += kBlock) {
                        gk2[s] = s < cnt ? sk[s] : ~0ull;
                        gi2[s] = s < cnt ? si[s] : ~0u;
                    }
                    __syncthreads();
                    bitonic_sort<kBlock, unsigned>(gk2, gi2, P);
                    part = segment_sum<kBlock>(gk2, gi2, cnt, a.m, have_prev, prev_phi, prev_m);
                    k0_phi = __longlong_as_double((long long)gk2[0]);
                    k0_m = a.m[gi2[0]];
                    k1_phi = __longlong_as_double((long long)gk2[cnt - 1]);
                    k1_m = a.m[gi2[cnt - 1]];
                } else {
                    // fine counting sort in LDS
                    const int nbk = hi - lo;
                    int g = 1;
                    while (nbk * (g << 1) <= kFine) g <<= 1;
                    const int nfine = nbk * g;
                    const double fscale = (double)kBuckets * (double)g;
                    const int foff = lo * g;
                    for (int f = tid; f <= nfine; f += kBlock) fine[f] = 0u;
                    if (tid == 0) s_max = 0u;
                    __syncthreads();
                    unsigned long long ek[kPer];
                    unsigned ei[kPer], er[kPer];
                    unsigned mymax = 0;
#pragma unroll
                    for (int e = 0; e < kPer; ++e) {
                        const int s = tid + e * kBlock;
                        if (s < cnt) {
                            ek[e] = sk[s];
                            ei[e] = si[s];
                            int fb = scaled_index(__longlong_as_double((long long)ek[e]), fscale,
                                                  foff + nfine - 1) - foff;
                            fb = fb < 0 ? 0 : fb;
                            er[e] = atomicAdd(&fine[fb], 1u);
                            mymax = er[e] + 1 > mymax ? er[e] + 1 : mymax;
                        }
                    }
                    atomicMax(&s_max, mymax);
                    __syncthreads();
                    const unsigned fullest = s_max;
                    if (fullest <= (unsigned)kInsertMax) {
                        block_exclusive_scan<kBlock>(fine, nfine + 1, wave_tot);  // fine[nfine] = cnt
#pragma unroll
                        for (int e = 0; e < kPer; ++e) {
                            const int s = tid + e * kBlock;
                            if (s < cnt) {
                                int fb = scaled_index(__longlong_as_double((long long)ek[e]), fscale,
                                                      foff + nfine - 1) - foff;
                                fb = fb < 0 ? 0 : fb;
                                const unsigned pos = fine[fb] + er[e];
                                keys[pos] = ek[e];
                                idxs[pos] = ei[e];
                            }
                        }
                        __syncthreads();
                        // finish: order each fine bucket by (bits, index)
                        for (int f = tid; f < nfine; f += kBlock) {
                            const int s0 = (int)fine[f], s1 = (int)fine[f + 1];
                            for (int x = s0 + 1; x < s1; ++x) {
                                const unsigned long long kx = keys[x];
                                const unsigned ix = idxs[x];
                                int y = x - 1;
                                while (y >= s0 && (keys[y] > kx || (keys[y] == kx && idxs[y] > ix))) {
                                    keys[y + 1] = keys[y];
                                    idxs[y + 1] = idxs[y];
                                    --y;
                                }
                                keys[y + 1] = kx;
                                idxs[y + 1] = ix;
                            }
                        }
                        __syncthreads();
                    } else {
                        // clustered phases: bitonic sort of the range in LDS
                        int P = 2;
                        while (P < cnt) P <<= 1;
#pragma unroll
                        for (int e = 0; e < kPer; ++e) {
                            const int s = tid + e * kBlock;
                            if (s < P) {
                                keys[s] = s < cnt ? ek[e] : ~0ull;
                                idxs[s] = s < cnt ? ei[e] : ~0u;
                            }
                        }
                        __syncthreads();
                        bitonic_sort<kBlock, unsigned>(keys, idxs, P);
                    }
                    part = segment_sum<kBlock>(keys, idxs, cnt, a.m, have_prev, prev_phi, prev_m);
                    k0_phi = __longlong_as_double((long long)keys[0]);
                    k0_m = a.m[idxs[0]];
                    k1_phi = __longlong_as_double((long long)keys[cnt - 1]);
                    k1_m = a.m[idxs[cnt - 1]];
                }
                total += part;
                if (!have_prev) {
                    first_phi = k0_phi;
                    first_m = k0_m;
                }
                prev_phi = k1_phi;
                prev_m = k1_m;
                have_prev = true;
                __syncthreads();  // keys/idxs/fine are reused by the next range
            }
            beg = hist[hi - 1];
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


// ------------------------------------------------------------------------------------------------
// All-LDS variant for N <= kLdsMaxN.  The coarse-bucket permutation is kept in LDS as 16-bit sample
// indices, so nothing but t[] and m[] (L2-resident, shared by every workgroup) is read from global
// memory.  1024 threads = 16 waves, one workgroup per CU.
//   P1/P2  block-wide: coarse histogram, scan, permutation `order[]` (coarse buckets chosen from
//          t * (1/period) with the same guard band as the PDM kernel: the exact IEEE division runs
//          only when the shortcut lands within its own error of a bucket edge).
//   P3a    WAVE-AUTONOMOUS ranges, no workgroup barrier: the sorted positions are cut into windows
//          of 192; range r = the coarse buckets whose first sorted position falls in window r (a
//          contiguous slice of order[], ~192-230 samples).  Each wave takes ranges r = wave,
//          wave+16, ...: exact fold of its <= 256 samples, rank inside <= 256 fine buckets (LDS
//          atomics, wave-private counters), wave-level exclusive scan, placement, insertion-sort
//          finish, segment sum, and a 4-double summary (first/last point) of the range.
//   P3b    ranges a wave cannot take (more than 256 samples, or a fine bucket fuller than 16:
//          clustered phases) are bitonic-sorted by the whole workgroup (LDS, or global scratch
//          beyond 2048 samples).
//   P3c    links between consecutive ranges and the closing segment from the summaries.
constexpr int kLBlock = 1024;
constexpr int kLWaves = kLBlock / 64;
constexpr int kWin = 192;
constexpr int kRCap = 256;
constexpr int kRPer = kRCap / 64;
constexpr int kWFine = 256;
constexpr int kWInsertMax = 16;
constexpr int kDCap = 2048;
constexpr int kWaveBytes = 3616;   // keys u64[256] | fine u32[260] | idx u16[256]
constexpr int kMaxRanges = 1024;
// the coarse histogram is only alive in P1/P2 and the per-wave scratch only in P3: they share LDS
constexpr int kLdsFixed = kLWaves * kWaveBytes + (kMaxRanges + 8) * 4 + 128;
static_assert(kLWaves * kWaveBytes >= kBuckets * 4, "histogram must fit in the per-wave scratch");
constexpr int kLdsMaxN = (163840 - kLdsFixed - 1024) / 2;

__device__ __forceinline__ int coarse_bucket(double t, double period, double rp, double thr) {
    const double q = t * rp;
    const double u = (q - __builtin_floor(q)) * (double)kBuckets;
    const int b = (int)u;
    if (__builtin_fabs((u - (double)b) - 0.5) < thr) return b;
    return scaled_index(fold_phase(t, period), (double)kBuckets, kBuckets - 1);
}

// LDS traffic between lanes of ONE wave: order the accesses without a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kLBlock) void sl_scan_lds_kernel(SlArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char *wbuf = lds_raw;                                                  // per-wave scratch
    unsigned long long *bkeys = reinterpret_cast<unsigned long long *>(lds_raw);    // P3b alias [kDCap]
    unsigned short *bidx = reinterpret_cast<unsigned short *>(bkeys + kDCap);       // P3b alias [kDCap]
    unsigned *hist = reinterpret_cast<unsigned *>(lds_raw);                         // P1/P2 alias [kBuckets]
    unsigned short *bndb = reinterpret_cast<unsigned short *>(lds_raw + kLWaves * kWaveBytes);  // [kMaxRanges + 8]
    unsigned short *bnds = bndb + kMaxRanges + 8;                                   // [kMaxRanges + 8]
    unsigned *defer = reinterpret_cast<unsigned *>(bnds + kMaxRanges + 8);          // [32]
    unsigned short *order = reinterpret_cast<unsigned short *>(defer + 32);         // [n]
    __shared__ unsigned wave_tot[kLWaves];
    __shared__ double red[kLWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (int)a.n;
    const int nranges = (n + kWin - 1) / kWin;
    unsigned long long *gk2 = a.gkeys2 + (int64_t)blockIdx.x * a.n_pad;
    unsigned *gi2 = a.gidx2 + (int64_t)blockIdx.x * a.n_pad;
    double *rsum = a.rsum + (int64_t)blockIdx.x * a.nr_pad * 4;
    int *rcnt = a.rcnt + (int64_t)blockIdx.x * a.nr_pad;

    // max |t| once per workgroup (guard band of the bucket shortcut)
    double tmax = 0.0;
    for (int i = tid; i < n; i += kLBlock) {
        const double at = __builtin_fabs(a.t[i]);
        tmax = at > tmax ? at : tmax;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_down(tmax, o, 64);
        tmax = u > tmax ? u : tmax;
    }
    if (lane == 0) red[wave] = tmax;
    __syncthreads();
    tmax = red[0];
    for (int w = 1; w < kLWaves; ++w) tmax = red[w] > tmax ? red[w] : tmax;
    __syncthreads();

    unsigned long long *keys_w = reinterpret_cast<unsigned long long *>(wbuf + wave * kWaveBytes);
    unsigned *fine_w = reinterpret_cast<unsigned *>(wbuf + wave * kWaveBytes + kRCap * 8);
    unsigned short *idx_w = reinterpret_cast<unsigned short *>(wbuf + wave * kWaveBytes + kRCap * 8 + (kWFine + 4) * 4);

    for (int64_t p = blockIdx.x; p < a.n_periods; p += gridDim.x) {
        const double period = a.periods[p];
        const double rp = 1.0 / period;
        const double thr = 0.5 - (double)kBuckets * (8.9e-16 * tmax * __builtin_fabs(rp) + 8.9e-16);
        // ---- P1: coarse histogram + exclusive scan ------------------------------------------
        for (int b = tid; b < kBuckets; b += kLBlock) hist[b] = 0u;
        if (tid < 32) defer[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += kLBlock) atomicAdd(&hist[coarse_bucket(a.t[i], period, rp, thr)], 1u);
        __syncthreads();
        block_exclusive_scan<kLBlock>(hist, kBuckets, wave_tot);
        // ---- P2: permutation by coarse bucket, in LDS -------------------------------------------
        for (int i = tid; i < n; i += kLBlock) {
            const unsigned pos = atomicAdd(&hist[coarse_bucket(a.t[i], period, rp, thr)], 1u);
            order[pos] = (unsigned short)i;
        }
        __syncthreads();  // hist[b] = END offset of bucket b
        // range boundaries: range r starts at the first bucket whose start offset is >= r * kWin
        for (int r = tid; r <= nranges; r += kLBlock) {
            int b = 0, s0 = 0;
            if (r > 0) {
                const unsigned x = (unsigned)r * kWin;
                int l = 0, h = kBuckets;  // smallest j with hist[j] >= x, kBuckets if none
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    if (hist[mid] >= x) h = mid; else l = mid + 1;
                }
                b = l < kBuckets ? l + 1 : kBuckets;
                s0 = l < kBuckets ? (int)hist[l] : n;
            }
            bndb[r] = (unsigned short)b;
            bnds[r] = (unsigned short)s0;
        }
        __syncthreads();

        // ---- P3a: wave-autonomous ranges -------------------------------------------------------
        double total = 0.0;
        for (int r = wave; r < nranges; r += kLWaves) {
            const int lo_b = bndb[r], hi_b = bndb[r + 1];
            const int s_lo = bnds[r], cnt = (int)bnds[r + 1] - s_lo;
            if (cnt <= 0) {
                if (lane == 0) rcnt[r] = 0;
                continue;
            }
            if (cnt > kRCap) {
                if (lane == 0) atomicOr(&defer[r >> 5], 1u << (r & 31));
                continue;
            }
            // monotone map of the range's phases onto <= kWFine fine buckets
            const int nbk = hi_b - lo_b;
            int g = 1, shift = 0;
            if (nbk <= kWFine) {
                while (nbk * (g << 1) <= kWFine) g <<= 1;
            } else {
                while ((nbk >> shift) + 1 > kWFine) ++shift;
            }
            const double fscale = (double)kBuckets * (double)g;
            const int foff = lo_b * g;
            const int flast = nbk <= kWFine ? nbk * g - 1 : (nbk >> shift);
            // zero the counters (260 entries)
            reinterpret_cast<uint4 *>(fine_w)[lane] = make_uint4(0u, 0u, 0u, 0u);
            if (lane < 4) fine_w[kWFine + lane] = 0u;
            wave_sync();
            unsigned long long ek[kRPer];
            unsigned ei[kRPer], er[kRPer];
            int ef[kRPer];
#pragma unroll
            for (int e = 0; e < kRPer; ++e) {
                const int s = lane + e * 64;
                if (s < cnt) {
                    ei[e] = order[s_lo + s];
                    const double phi = fold_phase(a.t[ei[e]], period);   // exact sort key
                    ek[e] = (unsigned long long)__double_as_longlong(phi);
                    int fb;
                    if (nbk <= kWFine) {
                        fb = scaled_index(phi, fscale, foff + flast) - foff;
                    } else {
                        fb = (scaled_index(phi, (double)kBuckets, kBuckets - 1) - lo_b) >> shift;
                    }
                    fb = fb < 0 ? 0 : (fb > flast ? flast : fb);
                    ef[e] = fb;
                    er[e] = atomicAdd(&fine_w[fb], 1u);
                }
            }
            wave_sync();
            // wave-level exclusive scan of the 256 counters (4 per lane) + fullest bucket
            const uint4 c = reinterpret_cast<uint4 *>(fine_w)[lane];
            const unsigned mx = max(max(c.x, c.y), max(c.z, c.w));
            if (__any(mx > (unsigned)kWInsertMax)) {
                if (lane == 0) atomicOr(&defer[r >> 5], 1u << (r & 31));
                continue;
            }
            const unsigned sum4 = c.x + c.y + c.z + c.w;
            unsigned incl = sum4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            const unsigned base = incl - sum4;
            wave_sync();
            reinterpret_cast<uint4 *>(fine_w)[lane] = make_uint4(base, base + c.x, base + c.x + c.y,
                                                                 base + c.x + c.y + c.z);
            if (lane == 63) fine_w[kWFine] = incl;  // == cnt
            wave_sync();
#pragma unroll
            for (int e = 0; e < kRPer; ++e) {
                const int s = lane + e * 64;
                if (s < cnt) {
                    const unsigned pos = fine_w[ef[e]] + er[e];
                    keys_w[pos] = ek[e];
                    idx_w[pos] = (unsigned short)ei[e];
                }
            }
            wave_sync();
            // finish: each lane orders its 4 fine buckets by (bits, index)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = lane * 4 + q;
                const int s0 = (int)fine_w[f], s1 = (int)fine_w[f + 1];
                for (int x = s0 + 1; x < s1; ++x) {
                    const unsigned long long kx = keys_w[x];
                    const unsigned short ix = idx_w[x];
                    int y = x - 1;
                    while (y >= s0 && (keys_w[y] > kx || (keys_w[y] == kx && idx_w[y] > ix))) {
                        keys_w[y + 1] = keys_w[y];
                        idx_w[y + 1] = idx_w[y];
                        --y;
                    }
                    keys_w[y + 1] = kx;
                    idx_w[y + 1] = ix;
                }
            }
            wave_sync();
            // segments inside the range
#pragma unroll
            for (int e = 0; e < kRPer; ++e) {
                const int j = lane + e * 64;
                const bool live = j < cnt;
                double phi = 0.0, mm = 0.0;
                if (live) {
                    phi = __longlong_as_double((long long)keys_w[j]);
                    mm = a.m[idx_w[j]];
                }
                double pphi = __shfl_up(phi, 1, 64);
                double pm = __shfl_up(mm, 1, 64);
                bool ok = live;
                if (lane == 0 && live) {
                    if (j > 0) {
                        pphi = __longlong_as_double((long long)keys_w[j - 1]);
                        pm = a.m[idx_w[j - 1]];
                    } else {
                        ok = false;
                    }
                }
                if (ok) total += hypot(mm - pm, phi - pphi);
            }
            if (lane == 0) {
                rsum[r * 4 + 0] = __longlong_as_double((long long)keys_w[0]);
                rsum[r * 4 + 1] = a.m[idx_w[0]];
                rsum[r * 4 + 2] = __longlong_as_double((long long)keys_w[cnt - 1]);
                rsum[r * 4 + 3] = a.m[idx_w[cnt - 1]];
                rcnt[r] = cnt;
            }
            wave_sync();
        }
        __syncthreads();

        // ---- P3b: deferred ranges, whole workgroup ---------------------------------------------
        for (int w32 = 0; w32 < (nranges + 31) / 32; ++w32) {
            unsigned bits = defer[w32];
            while (bits) {
                const int r = w32 * 32 + __builtin_ctz(bits);
                bits &= bits - 1;
                const int s_lo = bnds[r], cnt = (int)bnds[r + 1] - s_lo;
                int P = 2;
                while (P < cnt) P <<= 1;
                double part;
                if (cnt <= kDCap) {
                    for (int s = tid; s < P; s += kLBlock) {
                        if (s < cnt) {
                            const unsigned id = order[s_lo + s];
                            bkeys[s] = (unsigned long long)__double_as_longlong(fold_phase(a.t[id], period));
                            bidx[s] = (unsigned short)id;
                        } else {
                            bkeys[s] = ~0ull;
                            bidx[s] = (unsigned short)0xffff;
                        }
                    }
                    __syncthreads();
                    bitonic_sort<kLBlock, unsigned short>(bkeys, bidx, P);
                    part = segment_sum<kLBlock>(bkeys, bidx, cnt, a.m, false, 0.0, 0.0);
                    if (tid == 0) {
                        rsum[r * 4 + 0] = __longlong_as_double((long long)bkeys[0]);
                        rsum[r * 4 + 1] = a.m[bidx[0]];
                        rsum[r * 4 + 2] = __longlong_as_double((long long)bkeys[cnt - 1]);
                        rsum[r * 4 + 3] = a.m[bidx[cnt - 1]];
                        rcnt[r] = cnt;
                    }
                } else {
                    for (int s = tid; s < P; s += kLBlock) {
                        if (s < cnt) {
                            const unsigned id = order[s_lo + s];
                            gk2[s] = (unsigned long long)__double_as_longlong(fold_phase(a.t[id], period));
                            gi2[s] = id;
                        } else {
                            gk2[s] = ~0ull;
                            gi2[s] = ~0u;
                        }
                    }
                    __syncthreads();
                    bitonic_sort<kLBlock, unsigned>(gk2, gi2, P);
                    part = segment_sum<kLBlock>(gk2, gi2, cnt, a.m, false, 0.0, 0.0);
                    if (tid == 0) {
                        rsum[r * 4 + 0] = __longlong_as_double((long long)gk2[0]);
                        rsum[r * 4 + 1] = a.m[gi2[0]];
                        rsum[r * 4 + 2] = __longlong_as_double((long long)gk2[cnt - 1]);
                        rsum[r * 4 + 3] = a.m[gi2[cnt - 1]];
                        rcnt[r] = cnt;
                    }
                }
                total += part;
                __syncthreads();
            }
        }
        __syncthreads();  // summaries (global, this workgroup's) are visible

        // ---- P3c: links between consecutive non-empty ranges + the closing segment ----------------
        for (int r = tid; r < nranges; r += kLBlock) {
            if (rcnt[r] > 0) {
                int q = r - 1;
                while (q >= 0 && rcnt[q] == 0) --q;
                if (q >= 0)
                    total += hypot(rsum[r * 4 + 1] - rsum[q * 4 + 3], rsum[r * 4 + 0] - rsum[q * 4 + 2]);
            }
        }
        if (tid == 0 && nranges > 0) {
            int f0 = 0, l0 = nranges - 1;
            while (f0 < nranges && rcnt[f0] == 0) ++f0;
            while (l0 >= 0 && rcnt[l0] == 0) --l0;
            // closing segment of np.roll(-1): first minus last, no phase wrap (phase.py:50)
            if (f0 < nranges && l0 >= 0)
                total += hypot(rsum[f0 * 4 + 1] - rsum[l0 * 4 + 3], rsum[f0 * 4 + 0] - rsum[l0 * 4 + 2]);
        }
        total = wave_sum(total);
        if (lane == 0) red[wave] = total;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < kLWaves; ++w) sum += red[w];
            a.ell[p] = sum;
        }
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
    const int64_t nr = (n + 63) / 64 + 8;  // sized for the smallest window
    return grid_for(n_periods > 0 ? n_periods : 1) * (pad_pow2(n) * 24 + nr * 40) + 512;
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
    a.gkeys2 = a.gkeys + grid * a.n_pad;
    a.gidx = reinterpret_cast<unsigned *>(a.gkeys2 + grid * a.n_pad);
    a.gidx2 = a.gidx + grid * a.n_pad;
    a.nr_pad = (n + 63) / 64 + 8;
    a.rsum = reinterpret_cast<double *>(a.gkeys2 + grid * a.n_pad);  // placed after the 8-byte arrays
    a.gidx = reinterpret_cast<unsigned *>(a.rsum + grid * a.nr_pad * 4);
    a.gidx2 = a.gidx + grid * a.n_pad;
    a.rcnt = reinterpret_cast<int *>(a.gidx2 + grid * a.n_pad);
    static const int force_scratch = [] { const char *e = getenv("PDC_SL_SCRATCH"); return e ? atoi(e) : 0; }();
    a.use_lds = (n <= kLdsMaxN && !force_scratch) ? 1 : 0;
    if (a.use_lds) {
        const size_t lds = (size_t)kLdsFixed + (size_t)((n + 7) & ~(int64_t)7) * 2;
        PDC_HIP(hipFuncSetAttribute((const void *)sl_scan_lds_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sl_scan_lds_kernel, dim3((unsigned)grid), dim3(kLBlock), lds,
                           (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(sl_scan_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    }
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
