// Dworetsky String-Length sweep on gfx950.
//
// Replaces pool.map(StringLength._stringlength, periods)
// (/root/reference/src/periodicity/phase.py:45-51, 69-70) together with the fold
// ((t - 0) / period) % 1 (core.py:543-544) and the stable sort by phase done by the TSeries
// constructor (core.py:473-477).  The polygon is closed with np.roll, so the last->first segment
// is included and is NOT phase-wrapped.
//
// The sort order must be exactly numpy's: two samples a rounding apart in phase swap places and
// change the length by O(|dm|).  Sort keys are therefore the 64-bit patterns of phases computed
// with an IEEE division and Python modulo (monotone for phi in [0, 1], NaN last), ties broken by
// sample index (= stable sort of the time-ordered input).
//
// One workgroup (1024 threads = 16 waves, one per CU) per trial period, persistent grid.  N
// (phase, index) pairs do not fit in 160 KB of LDS, so the sort is two-level, linear-time, and
// keeps only sample INDICES in LDS (16-bit when N < 65536, else 32-bit):
//   P1   histogram of the phases over 2048 (8192 for N >= 65536) equal coarse buckets (LDS atomics)
//        + exclusive scan.
//        Coarse buckets come from t * (1/period) with the PDM kernel's guard band: the exact IEEE
//        division runs only when the shortcut lands within its own error of a bucket edge.
//   P2   the permutation `order[]`, grouped by coarse bucket, for as many consecutive buckets as
//        fit in LDS (a "slice"; N <= ~50k samples need one slice; for larger N the grouping is done
//        once per period in global scratch and every slice copies its contiguous piece).
//   P3a  WAVE-AUTONOMOUS ranges, no workgroup barrier: the slice's sorted positions are cut into
//        windows of 192; range r = the coarse buckets whose first sorted position falls in window r
//        (a contiguous piece of order[], ~192-230 samples).  Each wave takes ranges r = wave,
//        wave+16, ...: exact fold of its <= 256 samples (t[] is L2-resident), rank inside <= 256
//        fine buckets (a power-of-two refinement of the coarse index, so both are exact and
//        consistent; wave-private LDS counters), wave-level exclusive scan, placement, then every
//        sample counts the members of its fine bucket that sort before it (a wave-uniform loop
//        over the fullest bucket, mean occupancy < 1) and moves to its slot, segment sum with the previous
//        point taken from the neighbouring lane, and a 4-double summary (first/last point).
//   P3b  ranges a wave cannot take (more than 256 samples, or a fine bucket fuller than 16:
//        clustered phases from evenly sampled data at a commensurate period) are bitonic-sorted by
//        the whole workgroup in LDS (global scratch beyond 2048 samples).  A single coarse bucket
//        larger than a whole slice is gathered straight into the global scratch.
//   P3c  links between consecutive ranges and the closing segment, from the summaries.
// Nothing but t[], m[] (read) and ell[p] (written) touches global memory on the common path.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "pdc_internal.h"

using namespace pdc;

namespace {

constexpr int kBuckets = 2048;       // coarse buckets over [0, 1]
constexpr int kBucketsLarge = 8192;  // ... for N >= 65536, so that a bucket still fits a wave's range
constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
constexpr int kWin = 192;        // sorted positions per range window
constexpr int kRCap = 256;       // samples a wave sorts by itself (4 per lane)
constexpr int kRPer = kRCap / 64;
constexpr int kWFine = 256;      // fine buckets per wave range
constexpr int kCPL = kWFine / 64; // fine counters per lane (multiple of 4)
constexpr int kWInsertMax = 16;  // fullest fine bucket the predecessor-counting finish accepts
constexpr int kDCap = 2048;      // workgroup-level (deferred) LDS sort capacity
constexpr int kMaxRanges = 1024; // ranges per slice
constexpr int kMaxGrid = 1024;
constexpr int kLdsTotal = 163840;

template <typename IdxT>
struct Lds {
    // per wave: keys u64[kRCap] | fine u32[kWFine + 4] | idx IdxT[kRCap]
    static constexpr int wave_bytes = ((kRCap * 8 + (kWFine + 4) * 4 + kRCap * (int)sizeof(IdxT)) + 15) & ~15;
    // the coarse histogram is only alive in P1/P2 and the wave scratch only in P3: they share LDS
    static constexpr int fixed = kWaves * wave_bytes + (kMaxRanges + 8) * 4 + 128;
    static constexpr int capacity = (kLdsTotal - fixed - 1024) / (int)sizeof(IdxT);  // slice size
    static_assert(kWaves * wave_bytes >= kDCap * (8 + (int)sizeof(IdxT)), "deferred sort must fit");
    static_assert(capacity / kWin + 1 <= kMaxRanges, "range table too small");
    static_assert(capacity < 65536, "slice positions are stored as 16-bit offsets");
};

struct SlArgs {
    const double *t, *m, *periods;
    int64_t n, n_periods;
    double *ell;
    unsigned long long *gkeys;  // [grid][n_pad]  scratch for ranges too large for LDS
    unsigned *gidx;             // [grid][n_pad]
    double *rsum;               // [grid][nr_pad][4]  range summaries
    int *rcnt;                  // [grid][nr_pad]
    int64_t n_pad, nr_pad;
    unsigned *gorder;           // [grid][n_pad]  samples grouped by coarse bucket (several slices only)
    unsigned *ghist;            // [grid][kBucketsLarge]  first sorted position of every coarse bucket
    const unsigned char *todo = nullptr;   // NULL, or [n_periods]: only periods with a non-zero entry are worked off
    const unsigned *todo_count = nullptr;  // (with todo) how many entries are non-zero: 0 ends every workgroup at once
    const unsigned char *skip = nullptr;   // NULL, or [n_periods]: periods with a non-zero entry are done already (one cycle)
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

template <int NB>
__device__ __forceinline__ int coarse_bucket(double t, double period, double rp, double thr) {
    const double q = t * rp;
    const double u = (q - __builtin_floor(q)) * (double)NB;
    const int b = (int)u;
    if (__builtin_fabs((u - (double)b) - 0.5) < thr) return b;
    return scaled_index(fold_phase(t, period), (double)NB, NB - 1);
}

// LDS traffic between lanes of ONE wave: order the accesses without a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Exclusive prefix sum of a[0..NB) in place (NB / BLOCK entries per thread); returns the total.
template <int NB, int BLOCK = kBlock>
__device__ __forceinline__ unsigned scan_buckets(unsigned *a, unsigned *wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int per = NB / BLOCK;
    static_assert(per * BLOCK == NB, "bucket count must be a multiple of the block size");
    unsigned local[per], sum = 0;
#pragma unroll
    for (int e = 0; e < per; ++e) {
        local[e] = a[tid * per + e];
        sum += local[e];
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
    unsigned run = incl - sum, all = 0u;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        if (w < wave) run += wave_tot[w];
        all += wave_tot[w];
    }
#pragma unroll
    for (int e = 0; e < per; ++e) {
        a[tid * per + e] = run;
        run += local[e];
    }
    __syncthreads();
    return all;
}

// Ascending bitonic sort of P (power of two) (key, index) pairs by (key, index), whole workgroup.
template <typename IdxT, typename KeyPtr, typename IdxPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr K, IdxPtr I, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int c = threadIdx.x; c < (P >> 1); c += (int)blockDim.x) {
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

// Sum of hypot(dm, dphi) over consecutive sorted points j-1 -> j, j in [1, cnt), whole workgroup.
template <typename KeyPtr, typename IdxPtr>
__device__ __forceinline__ double segment_sum(KeyPtr K, IdxPtr I, int cnt, const double *m) {
    const int lane = threadIdx.x & 63;
    double total = 0.0;
    for (int j0 = 0; j0 < cnt; j0 += (int)blockDim.x) {
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
                ok = false;
            }
        }
        if (ok) total += hypot(mm - pm, phi - pphi);
    }
    return total;
}

template <typename IdxT, int NB>
__global__ __launch_bounds__(kBlock) void sl_scan_kernel(SlArgs a) {
    static_assert(kWaves * Lds<IdxT>::wave_bytes >= NB * 4, "histogram must fit in the wave scratch");
    using L = Lds<IdxT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char *wbuf = lds_raw;                                                  // P3a: per-wave scratch
    unsigned *hist = reinterpret_cast<unsigned *>(lds_raw);                         // P1/P2 alias [NB]
    unsigned long long *bkeys = reinterpret_cast<unsigned long long *>(lds_raw);    // P3b alias [kDCap]
    IdxT *bidx = reinterpret_cast<IdxT *>(bkeys + kDCap);                           // P3b alias [kDCap]
    unsigned short *bndb = reinterpret_cast<unsigned short *>(lds_raw + kWaves * L::wave_bytes);
    unsigned short *bnds = bndb + kMaxRanges + 8;                                   // [kMaxRanges + 8] each
    unsigned *defer = reinterpret_cast<unsigned *>(bnds + kMaxRanges + 8);          // [32]
    IdxT *order = reinterpret_cast<IdxT *>(defer + 32);                             // [slice]
    __shared__ unsigned wave_tot[kWaves];
    __shared__ unsigned s_fill;
    __shared__ double red[kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n = a.n;
    unsigned long long *gk = a.gkeys + (int64_t)blockIdx.x * a.n_pad;
    unsigned *gi = a.gidx + (int64_t)blockIdx.x * a.n_pad;
    double *rsum = a.rsum + (int64_t)blockIdx.x * a.nr_pad * 4;
    int *rcnt = a.rcnt + (int64_t)blockIdx.x * a.nr_pad;
    // more samples than one slice holds: the grouping by coarse bucket is done ONCE per period in
    // global scratch and every slice copies its contiguous piece, instead of a histogram pass and a
    // selection pass over all samples per slice
    const bool multi = n > L::capacity;
    unsigned *gorder = multi ? a.gorder + (int64_t)blockIdx.x * a.n_pad : nullptr;
    unsigned *ghist = multi ? a.ghist + (int64_t)blockIdx.x * NB : nullptr;

    // behind the streamed kernels: usually no period was left over
    if (a.todo && *a.todo_count == 0u) return;
    // max |t| once per workgroup (guard band of the bucket shortcut)
    double tmax = 0.0;
    for (int64_t i = tid; i < n; i += kBlock) {
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
    for (int w = 1; w < kWaves; ++w) tmax = red[w] > tmax ? red[w] : tmax;
    __syncthreads();

    unsigned long long *keys_w = reinterpret_cast<unsigned long long *>(wbuf + wave * L::wave_bytes);
    unsigned *fine_w = reinterpret_cast<unsigned *>(wbuf + wave * L::wave_bytes + kRCap * 8);
    IdxT *idx_w = reinterpret_cast<IdxT *>(wbuf + wave * L::wave_bytes + kRCap * 8 + (kWFine + 4) * 4);

    // one summary per range: {first phi, first m, last phi, last m}
    auto write_summary = [&](int r, double p0, double m0, double p1, double m1, int cnt) {
        rsum[(int64_t)r * 4 + 0] = p0;
        rsum[(int64_t)r * 4 + 1] = m0;
        rsum[(int64_t)r * 4 + 2] = p1;
        rsum[(int64_t)r * 4 + 3] = m1;
        rcnt[r] = cnt;
    };

    for (int64_t p = blockIdx.x; p < a.n_periods; p += gridDim.x) {
        if (a.todo && !a.todo[p]) continue;   // (workgroup-uniform)
        if (a.skip && a.skip[p]) continue;    // (workgroup-uniform)
        const double period = a.periods[p];
        const double rp = 1.0 / period;
        const double thr = 0.5 - (double)NB * (8.9e-16 * tmax * __builtin_fabs(rp) + 8.9e-16);
        double total = 0.0;    // this thread's share of the string length
        int r_base = 0;        // ranges emitted so far
        int64_t consumed = 0;  // samples (in sorted order) already accounted for

        // ---- P1: coarse histogram of ALL samples + exclusive scan -----------------------------
        auto histogram = [&]() {
            for (int b = tid; b < NB; b += kBlock) hist[b] = 0u;
            __syncthreads();
            // (four coalesced loads in flight per thread: an LDS atomic between two loads would
            // otherwise hold the second one back)
            for (int64_t i0 = tid; i0 < n; i0 += 4 * kBlock) {
                double tv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = i0 + (int64_t)u * kBlock;
                    tv[u] = i < n ? a.t[i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + (int64_t)u * kBlock < n) atomicAdd(&hist[coarse_bucket<NB>(tv[u], period, rp, thr)], 1u);
            }
            __syncthreads();
            scan_buckets<NB>(hist, wave_tot);  // hist[b] = first sorted position of bucket b
        };
        if (multi) {
            histogram();
            for (int b = tid; b < NB; b += kBlock) ghist[b] = hist[b];
            __syncthreads();
            for (int64_t i0 = tid; i0 < n; i0 += 4 * kBlock) {
                double tv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = i0 + (int64_t)u * kBlock;
                    tv[u] = i < n ? a.t[i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = i0 + (int64_t)u * kBlock;
                    if (i < n) gorder[atomicAdd(&hist[coarse_bucket<NB>(tv[u], period, rp, thr)], 1u)] = (unsigned)i;
                }
            }
            __syncthreads();
        }

        while (consumed < n) {
            if (tid < 32) defer[tid] = 0u;
            if (tid == 0) s_fill = 0u;
            if (multi) {
                for (int b = tid; b < NB; b += kBlock) hist[b] = ghist[b];
                __syncthreads();
            } else {
                histogram();
            }
            auto end_of = [&](int b) -> int64_t { return b + 1 < NB ? (int64_t)hist[b + 1] : n; };
            // first bucket that still has unconsumed samples (consumed == its start)
            int b0;
            {
                int l = 0, h = NB - 1;
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    if (end_of(mid) > consumed) h = mid; else l = mid + 1;
                }
                b0 = l;
            }
            const int64_t first_cnt = end_of(b0) - consumed;
            if (first_cnt > L::capacity) {
                // ---- a single coarse bucket larger than a slice: sort it in global scratch ------
                const int cnt = (int)first_cnt;
                int P = 2;
                while (P < cnt) P <<= 1;
                __syncthreads();
                for (int64_t i = tid; i < n; i += kBlock) {
                    if (coarse_bucket<NB>(a.t[i], period, rp, thr) == b0) {
                        const unsigned slot = atomicAdd(&s_fill, 1u);
                        gk[slot] = (unsigned long long)__double_as_longlong(fold_phase(a.t[i], period));
                        gi[slot] = (unsigned)i;
                    }
                }
                for (int s = cnt + tid; s < P; s += kBlock) {
                    gk[s] = ~0ull;
                    gi[s] = ~0u;
                }
                __syncthreads();
                bitonic_sort<unsigned>(gk, gi, P);
                total += segment_sum(gk, gi, cnt, a.m);
                if (tid == 0)
                    write_summary(r_base, __longlong_as_double((long long)gk[0]), a.m[gi[0]],
                                  __longlong_as_double((long long)gk[cnt - 1]), a.m[gi[cnt - 1]], cnt);
                __syncthreads();
                r_base += 1;
                consumed += cnt;
                continue;
            }
            // ---- slice = buckets [b0, b1) with at most `capacity` samples -------------------------
            int b1;
            {
                int l = b0 + 1, h = NB;
                while (l < h) {
                    const int mid = (l + h + 1) >> 1;
                    if (end_of(mid - 1) - consumed <= L::capacity) l = mid; else h = mid - 1;
                }
                b1 = l;
            }
            const int slice_n = (int)(end_of(b1 - 1) - consumed);
            const int nranges = (slice_n + kWin - 1) / kWin;
            __syncthreads();  // every thread has read what it needs from the start offsets
            // ---- P2: permutation of the slice, grouped by coarse bucket, in LDS ------------------
            if (multi) {
                for (int sidx = tid; sidx < slice_n; sidx += kBlock) order[sidx] = (IdxT)gorder[consumed + sidx];
                for (int b = b0 + tid; b < b1; b += kBlock) hist[b] = b + 1 < NB ? ghist[b + 1] : (unsigned)n;
            } else
            for (int64_t i0 = tid; i0 < n; i0 += 4 * kBlock) {
                double tv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = i0 + (int64_t)u * kBlock;
                    tv[u] = i < n ? a.t[i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = i0 + (int64_t)u * kBlock;
                    if (i < n) {
                        const int b = coarse_bucket<NB>(tv[u], period, rp, thr);
                        if (b >= b0 && b < b1) {
                            const unsigned pos = atomicAdd(&hist[b], 1u);
                            order[pos - (unsigned)consumed] = (IdxT)i;
                        }
                    }
                }
            }
            __syncthreads();  // for b in [b0, b1): hist[b] = END position of bucket b
            // range r starts at the first bucket whose start position is >= consumed + r * kWin
            for (int r = tid; r <= nranges; r += kBlock) {
                int b = b0, s0 = 0;
                if (r > 0) {
                    const int64_t x = consumed + (int64_t)r * kWin;
                    int l = b0, h = b1;  // smallest j in [b0, b1) with end(j) >= x, b1 if none
                    while (l < h) {
                        const int mid = (l + h) >> 1;
                        if ((int64_t)hist[mid] >= x) h = mid; else l = mid + 1;
                    }
                    b = l < b1 ? l + 1 : b1;
                    s0 = l < b1 ? (int)((int64_t)hist[l] - consumed) : slice_n;
                }
                bndb[r] = (unsigned short)b;
                bnds[r] = (unsigned short)s0;
            }
            __syncthreads();  // (the histogram is dead from here on: its LDS becomes wave scratch)

            // ---- P3a: wave-autonomous ranges ---------------------------------------------------------
            for (int r = wave; r < nranges; r += kWaves) {
                const int lo_b = bndb[r], hi_b = bndb[r + 1];
                const int s_lo = bnds[r], cnt = (int)bnds[r + 1] - s_lo;
                if (cnt <= 0) {
                    if (lane == 0) rcnt[r_base + r] = 0;
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
                const double fscale = (double)NB * (double)g;
                const int foff = lo_b * g;
                const int flast = nbk <= kWFine ? nbk * g - 1 : (nbk >> shift);
#pragma unroll
                for (int q = 0; q < kCPL / 4; ++q)
                    reinterpret_cast<uint4 *>(fine_w)[lane * (kCPL / 4) + q] = make_uint4(0u, 0u, 0u, 0u);
                if (lane < 4) fine_w[kWFine + lane] = 0u;
                wave_sync();
                unsigned long long ek[kRPer];
                unsigned ei[kRPer], er[kRPer];
                int ef[kRPer];
                // all gathers of t[] through the permutation first (one L2 round trip for the four of them,
                // not one per element: the LDS atomics below would otherwise fence them apart)
                double et[kRPer];
#pragma unroll
                for (int e = 0; e < kRPer; ++e) {
                    const int s = lane + e * 64;
                    ei[e] = s < cnt ? (unsigned)order[s_lo + s] : (unsigned)order[s_lo];
                }
#pragma unroll
                for (int e = 0; e < kRPer; ++e) et[e] = a.t[ei[e]];
#pragma unroll
                for (int e = 0; e < kRPer; ++e) {
                    const int s = lane + e * 64;
                    if (s < cnt) {
                        const double phi = fold_phase(et[e], period);   // exact sort key
                        ek[e] = (unsigned long long)__double_as_longlong(phi);
                        int fb;
                        if (nbk <= kWFine) {
                            fb = scaled_index(phi, fscale, foff + flast) - foff;
                        } else {
                            fb = (scaled_index(phi, (double)NB, NB - 1) - lo_b) >> shift;
                        }
                        fb = fb < 0 ? 0 : (fb > flast ? flast : fb);
                        ef[e] = fb;
                        er[e] = atomicAdd(&fine_w[fb], 1u);
                    }
                }
                wave_sync();
                // wave-level exclusive scan of the kWFine counters (kCPL per lane) + fullest bucket
                unsigned c[kCPL];
#pragma unroll
                for (int q = 0; q < kCPL / 4; ++q) {
                    const uint4 v = reinterpret_cast<uint4 *>(fine_w)[lane * (kCPL / 4) + q];
                    c[4 * q] = v.x;
                    c[4 * q + 1] = v.y;
                    c[4 * q + 2] = v.z;
                    c[4 * q + 3] = v.w;
                }
                unsigned mx = 0, sum4 = 0;
#pragma unroll
                for (int q = 0; q < kCPL; ++q) {
                    mx = c[q] > mx ? c[q] : mx;
                    sum4 += c[q];
                }
                if (__any(mx > (unsigned)kWInsertMax)) {
                    if (lane == 0) atomicOr(&defer[r >> 5], 1u << (r & 31));
                    continue;
                }
                unsigned incl = sum4;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned up = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += up;
                }
                unsigned run = incl - sum4;
                wave_sync();
#pragma unroll
                for (int q = 0; q < kCPL; ++q) {
                    const unsigned t = c[q];
                    c[q] = run;
                    run += t;
                }
#pragma unroll
                for (int q = 0; q < kCPL / 4; ++q)
                    reinterpret_cast<uint4 *>(fine_w)[lane * (kCPL / 4) + q] =
                        make_uint4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
                if (lane == 63) fine_w[kWFine] = incl;  // == cnt
                wave_sync();
                // Every sample knows its fine bucket [b0, b1) of the sorted range; its final slot is b0 +
                // (number of bucket members that sort before it).  First pass: members are parked in
                // arrival order; second pass: each sample counts its predecessors (a wave-uniform loop
                // over the fullest bucket, <= kWInsertMax trips, no divergent per-lane sorting) and moves
                // to its slot.  Ties in phase fall back to the sample index (stable sort).
                // (LDS reads of one step are issued together, ahead of the step's LDS writes: the compiler
                // cannot reorder them across a possibly aliasing store itself)
                unsigned eb0[kRPer], eb1[kRPer];
#pragma unroll
                for (int e = 0; e < kRPer; ++e) {
                    const bool live = lane + e * 64 < cnt;
                    const int f = live ? ef[e] : 0;
                    eb0[e] = fine_w[f];
                    eb1[e] = fine_w[f + 1];
                    if (!live) eb0[e] = eb1[e] = 0u;
                }
#pragma unroll
                for (int e = 0; e < kRPer; ++e) {
                    if (lane + e * 64 < cnt) {
                        keys_w[eb0[e] + er[e]] = ek[e];
                        idx_w[eb0[e] + er[e]] = (IdxT)ei[e];
                    }
                }
                wave_sync();
                unsigned before[kRPer];
#pragma unroll
                for (int e = 0; e < kRPer; ++e) before[e] = 0u;
                for (unsigned j = 0; __any(mx > j); ++j) {
                    unsigned long long ky[kRPer];
                    bool tie = false;
#pragma unroll
                    for (int e = 0; e < kRPer; ++e) {
                        const bool in = eb0[e] + j < eb1[e];
                        ky[e] = keys_w[in ? eb0[e] + j : 0u];
                        if (in && ky[e] < ek[e]) ++before[e];
                        tie = tie || (in && ky[e] == ek[e] && j != er[e]);  // (slot er[e] is the sample itself)
                    }
                    if (__any(tie)) {  // equal phases of two different samples (rare): the index decides
#pragma unroll
                        for (int e = 0; e < kRPer; ++e) {
                            const bool in = eb0[e] + j < eb1[e];
                            if (in && ky[e] == ek[e] && (unsigned)idx_w[eb0[e] + j] < ei[e]) ++before[e];
                        }
                    }
                }
                wave_sync();
#pragma unroll
                for (int e = 0; e < kRPer; ++e) {
                    const int s = lane + e * 64;
                    if (s < cnt) {
                        keys_w[eb0[e] + before[e]] = ek[e];
                        idx_w[eb0[e] + before[e]] = (IdxT)ei[e];
                    }
                }
                wave_sync();
                // segments inside the range
                // (same for m[]: the four gathers go out together)
                double sphi[kRPer], sm[kRPer];
#pragma unroll
                for (int e = 0; e < kRPer; ++e) {
                    const int j = lane + e * 64;
                    const int jj = j < cnt ? j : 0;
                    sphi[e] = __longlong_as_double((long long)keys_w[jj]);
                    sm[e] = a.m[idx_w[jj]];
                }
                // the predecessor of a lane's point sits in the neighbouring lane; lane 0 takes it from
                // lane 63 of the previous row (carried as a wave-uniform pair), so nothing is re-read
                double carry_phi = 0.0, carry_m = 0.0, last_phi = 0.0, last_m = 0.0;
#pragma unroll
                for (int e = 0; e < kRPer; ++e) {
                    const int j = lane + e * 64;
                    const bool live = j < cnt;
                    const double phi = live ? sphi[e] : 0.0, mm = live ? sm[e] : 0.0;
                    double pphi = __shfl_up(phi, 1, 64);
                    double pm = __shfl_up(mm, 1, 64);
                    if (lane == 0) {
                        pphi = carry_phi;
                        pm = carry_m;
                    }
                    if (live && j > 0) total += hypot(mm - pm, phi - pphi);
                    carry_phi = __shfl(phi, 63, 64);
                    carry_m = __shfl(mm, 63, 64);
                    if (e == ((cnt - 1) >> 6)) {  // wave-uniform: the row that holds the range's last point
                        last_phi = __shfl(phi, (cnt - 1) & 63, 64);
                        last_m = __shfl(mm, (cnt - 1) & 63, 64);
                    }
                }
                if (lane == 0) write_summary(r_base + r, sphi[0], sm[0], last_phi, last_m, cnt);
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
                    double p0, m0, p1, m1;
                    if (cnt <= kDCap) {
                        for (int s = tid; s < P; s += kBlock) {
                            if (s < cnt) {
                                const IdxT id = order[s_lo + s];
                                bkeys[s] = (unsigned long long)__double_as_longlong(fold_phase(a.t[id], period));
                                bidx[s] = id;
                            } else {
                                bkeys[s] = ~0ull;
                                bidx[s] = (IdxT)~0u;
                            }
                        }
                        __syncthreads();
                        bitonic_sort<IdxT>(bkeys, bidx, P);
                        total += segment_sum(bkeys, bidx, cnt, a.m);
                        p0 = __longlong_as_double((long long)bkeys[0]);
                        m0 = a.m[bidx[0]];
                        p1 = __longlong_as_double((long long)bkeys[cnt - 1]);
                        m1 = a.m[bidx[cnt - 1]];
                    } else {
                        for (int s = tid; s < P; s += kBlock) {
                            if (s < cnt) {
                                const unsigned id = order[s_lo + s];
                                gk[s] = (unsigned long long)__double_as_longlong(fold_phase(a.t[id], period));
                                gi[s] = id;
                            } else {
                                gk[s] = ~0ull;
                                gi[s] = ~0u;
                            }
                        }
                        __syncthreads();
                        bitonic_sort<unsigned>(gk, gi, P);
                        total += segment_sum(gk, gi, cnt, a.m);
                        p0 = __longlong_as_double((long long)gk[0]);
                        m0 = a.m[gi[0]];
                        p1 = __longlong_as_double((long long)gk[cnt - 1]);
                        m1 = a.m[gi[cnt - 1]];
                    }
                    if (tid == 0) write_summary(r_base + r, p0, m0, p1, m1, cnt);
                    __syncthreads();
                }
            }
            __syncthreads();
            r_base += nranges;
            consumed += slice_n;
        }
        __syncthreads();  // every summary of this period (global, this workgroup's) is visible

        // ---- P3c: links between consecutive non-empty ranges + the closing segment ----------------
        for (int r = tid; r < r_base; r += kBlock) {
            if (rcnt[r] > 0) {
                int q = r - 1;
                while (q >= 0 && rcnt[q] == 0) --q;
                if (q >= 0)
                    total += hypot(rsum[(int64_t)r * 4 + 1] - rsum[(int64_t)q * 4 + 3],
                                   rsum[(int64_t)r * 4 + 0] - rsum[(int64_t)q * 4 + 2]);
            }
        }
        if (tid == 0 && r_base > 0) {
            int f0 = 0, l0 = r_base - 1;
            while (f0 < r_base && rcnt[f0] == 0) ++f0;
            while (l0 >= 0 && rcnt[l0] == 0) --l0;
            // closing segment of np.roll(-1): first minus last, no phase wrap (phase.py:50)
            if (f0 < r_base && l0 >= 0)
                total += hypot(rsum[(int64_t)f0 * 4 + 1] - rsum[(int64_t)l0 * 4 + 3],
                               rsum[(int64_t)f0 * 4 + 0] - rsum[(int64_t)l0 * 4 + 2]);
        }
        total = wave_sum(total);
        if (lane == 0) red[wave] = total;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < kWaves; ++w) sum += red[w];
            a.ell[p] = sum;
        }
        __syncthreads();
    }
}

// =====================================================================================================
// Fast path: N <= Fast::capacity samples (one slice, 16-bit sample indices; C5's N = 5e4 fits).
//
// What bounds a sort-by-phase on this chip is the gather that brings the samples into phase order:
// a random access into an L2-resident table costs ~2.3 cycles per lane per CU whatever its width
// (tools/ubench/gather_rate.hip: the 16 L2 channels of an XCD serve 32 CUs), ~146 cycles per
// wave-instruction against ~24 for a coalesced one.  This kernel therefore gathers ONCE per sample and
// period - a 16-byte (t, m) record (AoS table built by sl_prep_kernel) - where the general kernel above
// gathers t[] and later m[]; everything else is arranged to run beside that stream:
//   P1   exact fold of every sample, read coalesced from t[] (the quotient t/period comes from a
//        correctly rounded reciprocal and two fma corrections, `exact_quotient`, 5 instructions, not the
//        ~12 of an IEEE division); coarse histogram over 2048 buckets by LDS atomics; the bucket of each
//        of a thread's <= 52 samples stays in registers (two 16-bit ids per VGPR),
//   P2   so the permutation order[] (grouped by coarse bucket) is filled without touching t[] again;
//   P3a  wave-autonomous ranges of <= 256 sorted positions as above, with the records of the NEXT range
//        requested before the current one is ranked; rank = fine-bucket start + number of bucket members
//        that sort before (LDS); the sorted neighbour's (phase, m) is taken by writing one's own at the
//        final slot and reading slot - 1, first the phases, then m through the same 2 KB;
//        segment length = sqrt(dm^2 + dphi^2) by rsq + two refinement steps (<= 1 ulp);
//   P3b/P3c  deferred ranges and the links between ranges, as above.
namespace fast {

// samples per software-pipelined group in P1 (loads + folds) and P2 (rank atomics + index stores)
#ifndef PDC_SL_G1
#define PDC_SL_G1 4
#endif
// PMC / timing experiments only (results are garbage): 1 = every period stops after P1 (fold, histogram, scan),
// 2 = after P2 and the range table - what each phase costs and how many LDS bank-conflict cycles it makes
#ifndef PDC_SL_STOP
#define PDC_SL_STOP 0
#endif
#ifndef PDC_SL_G2
#define PDC_SL_G2 4
#endif
constexpr int kNB = kBuckets;
constexpr int kFWin = 216;             // sorted positions per range window (a range = window + < one bucket)
constexpr int kFCap = 255;             // samples a wave ranks by itself: counts and starts fit in bytes
constexpr int kFine = 1024;            // fine buckets per range, 8-bit counters packed four to a word
constexpr int kRanges = 320;           // >= capacity / kFWin + 2
constexpr int kKMax = 52;              // samples per thread in P1/P2 (capacity <= kKMax * kBlock)
constexpr int kNBLarge = kBucketsLarge; // coarse buckets of the several-slice instance (N above one slice)
constexpr int kStatic = 512;           // static __shared__ below, rounded up
constexpr int kLdsTotalDyn = kLdsTotal - kStatic;
// LDS layout per index type: 16-bit sample indices while one slice holds all samples, 32-bit beyond
template <typename IdxT>
struct FL {
    static constexpr int wave_bytes = Lds<IdxT>::wave_bytes;
    static constexpr int fixed = kWaves * wave_bytes + 2 * (kRanges + 8) * 2 + 64;
    static constexpr int capacity = ((((kLdsTotal - fixed - kStatic) / (int)sizeof(IdxT)) & ~7) - 64);   // (+ 64 dummy slots)
};
constexpr int kCapacity = FL<unsigned short>::capacity;   // samples the one-slice instances take
// slices with bit planes: 16 bits per entry + one / two plane bits, beside the 32-bit instance's wave scratch
constexpr int kCapacity17 = ((((kLdsTotal - FL<unsigned>::fixed - kStatic) * 8 / 17) & ~31) - 64 - 32);
constexpr int kCapacity18 = ((((kLdsTotal - FL<unsigned>::fixed - kStatic) * 8 / 18) & ~31) - 64 - 32);
static_assert(kCapacity17 < 65536 && kCapacity18 > 40000, "slice positions are 16-bit");
static_assert(kCapacity / kFWin + 2 <= kRanges, "range table too small");
static_assert(kFine + 16 <= (kWFine + 4) * 4, "byte counters live in the general kernel's counter area");
static_assert(kCapacity <= kKMax * kBlock, "P1 keeps one bucket id per sample in registers");
static_assert(kCapacity < 65536 && FL<unsigned>::capacity < 65536, "slice positions are 16-bit");
static_assert(kWaves * FL<unsigned short>::wave_bytes >= (kNB + 64) * 4 && kWaves * FL<unsigned short>::wave_bytes >= kDCap * 10,
              "aliases must fit");
static_assert(kWaves * FL<unsigned>::wave_bytes >= kNBLarge * 4 && kWaves * FL<unsigned>::wave_bytes >= kDCap * 12,
              "aliases must fit");

typedef double rec_t __attribute__((ext_vector_type(2)));   // (t, m)

struct FastArgs {
    const double *t, *m, *periods;
    const rec_t *rec;
    const unsigned *flags;      // [0] != 0: every t is 0 or 1e-150 <= |t| <= 1e150
    int64_t n, n_periods;
    double *ell;
    unsigned long long *gkeys;  // [grid][n_pad]  deferred ranges larger than kDCap
    unsigned *gidx;             // [grid][n_pad]
    double *rsum;               // [grid][nr_pad][4]
    int *rcnt;                  // [grid][nr_pad]
    double *rlen;               // [grid][nr_pad]  string length inside each range
    int64_t n_pad, nr_pad;
    unsigned *ghist;            // [grid][kNBLarge]  first sorted position of every coarse bucket (several slices)
    unsigned short *gbucket;    // [grid][2 n_pad]   coarse bucket of every sample, written by P1 (several slices)
    int slice_cap;              // samples per LDS slice (several slices)
    const unsigned char *todo;  // NULL, or [n_periods]: only periods with a non-zero entry are worked off
    const unsigned *todo_count; // (with todo) how many entries are non-zero: 0 ends every workgroup at once
    const unsigned char *skip;  // NULL, or [n_periods]: periods with a non-zero entry are done already (one cycle)
    rec_t *sorted;              // (EMIT instances) [n_periods][n]: every period's (phase, m) in sorted order
};

// RN(t / period) without the division: y = RN(1 / period); q0 = RN(t y) is within 1.5 ulp of the
// quotient, one fma correction makes it faithful, and for a faithful q and a correctly rounded
// reciprocal Markstein's theorem (IBM J. Res. Dev. 34, 1990; Muller et al., Handbook of
// Floating-Point Arithmetic, "division with an FMA") says the second correction IS the correctly
// rounded quotient.  The theorem needs no over/underflow and a divisor whose significand is not all
// ones: `safe` (wave-uniform) says so, otherwise the IEEE division runs.
__device__ __forceinline__ double exact_quotient(double t, double period, double y, bool safe) {
    if (!safe) return t / period;
    const double q0 = t * y;
    const double r0 = __builtin_fma(-period, q0, t);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-period, q1, t);
    return __builtin_fma(r1, y, q1);
}

__device__ __forceinline__ bool period_is_safe(double period, bool t_safe) {
    const double ap = __builtin_fabs(period);
    const unsigned long long frac = (unsigned long long)__double_as_longlong(period) & 0xFFFFFFFFFFFFFull;
    return t_safe && ap >= 1e-150 && ap <= 1e150 && frac != 0xFFFFFFFFFFFFFull;
}

// ((t - 0) / period) % 1 as numpy computes it (core.py:544); NaN when the quotient is not finite.
__device__ __forceinline__ double fast_phase(double t, double period, double y, bool safe) {
    const double q = exact_quotient(t, period, y, safe);
    return q - __builtin_floor(q);
}

__device__ __forceinline__ unsigned long long phase_key(double phi) {
    // monotone for phi in [0, 1]; every NaN gets one pattern, above all numbers, so that NaN phases
    // keep their time order (stable sort)
    return phi == phi ? (unsigned long long)__double_as_longlong(phi) : 0x7FF8000000000000ull;
}

template <int NB = kNB>
__device__ __forceinline__ int coarse_of(double phi) {
    const unsigned b = (unsigned)(phi * (double)NB);            // exact product; v_cvt_u32 saturates
    const unsigned c = b < (unsigned)(NB - 1) ? b : (unsigned)(NB - 1);
    return phi == phi ? (int)c : NB - 1;
}

// Wave-wide inclusive scan / maximum by DPP row shifts and row broadcasts (gfx9 encodings: row_shr:n =
// 0x110 + n, row_bcast15 = 0x142, row_bcast31 = 0x143); no LDS permutes.  The maximum ends in lane 63.
#define PDC_DPP(old, src, ctrl, rmask) \
    (unsigned)__builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), (rmask), 0xf, false)
__device__ __forceinline__ unsigned wave_scan_add(unsigned v) {
    v += PDC_DPP(0u, v, 0x111, 0xf);
    v += PDC_DPP(0u, v, 0x112, 0xf);
    v += PDC_DPP(0u, v, 0x114, 0xf);
    v += PDC_DPP(0u, v, 0x118, 0xf);
    v += PDC_DPP(0u, v, 0x142, 0xa);
    v += PDC_DPP(0u, v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {   // valid in lane 63 -> broadcast
    unsigned u;
    u = PDC_DPP(0u, v, 0x111, 0xf); v = u > v ? u : v;
    u = PDC_DPP(0u, v, 0x112, 0xf); v = u > v ? u : v;
    u = PDC_DPP(0u, v, 0x114, 0xf); v = u > v ? u : v;
    u = PDC_DPP(0u, v, 0x118, 0xf); v = u > v ? u : v;
    u = PDC_DPP(0u, v, 0x142, 0xa); v = u > v ? u : v;
    u = PDC_DPP(0u, v, 0x143, 0xc); v = u > v ? u : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Wave-wide sum of doubles in a FIXED order (the scan pattern above on both halves of the double; the
// total ends in lane 63 and is broadcast): no LDS permutes on the critical path of a range.
#define PDC_DPP_F64(v, ctrl, rmask)                                                                          \
    __longlong_as_double(((long long)PDC_DPP(0u, (unsigned)(__double_as_longlong(v) >> 32), ctrl, rmask) << 32) | \
                         (long long)PDC_DPP(0u, (unsigned)__double_as_longlong(v), ctrl, rmask))
__device__ __forceinline__ double wave_sum_fixed(double v) {
    v += PDC_DPP_F64(v, 0x111, 0xf);
    v += PDC_DPP_F64(v, 0x112, 0xf);
    v += PDC_DPP_F64(v, 0x114, 0xf);
    v += PDC_DPP_F64(v, 0x118, 0xf);
    v += PDC_DPP_F64(v, 0x142, 0xa);
    v += PDC_DPP_F64(v, 0x143, 0xc);
    const long long b = __double_as_longlong(v);
    return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), 63) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)b, 63));
}

// x of the lane below (lane 0: `first`), by DPP wave_shr:1 on both halves - no LDS permute
__device__ __forceinline__ double lane_below(double x, double first) {
    const long long xb = __double_as_longlong(x), fb = __double_as_longlong(first);
    const int lo = __builtin_amdgcn_update_dpp((int)fb, (int)xb, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(fb >> 32), (int)(xb >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double read_lane(double x, int l) {   // l wave-uniform
    const long long xb = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)xb, l), hi = __builtin_amdgcn_readlane((int)(xb >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// sqrt(x^2 + y^2) for |x|, |y| <= ~1e150 whose squares do not underflow to a subnormal that matters:
// here |dm| <= 0.5 and |dphi| <= 1.  rsq seed, one coupled Goldschmidt step, one Newton step.
__device__ __forceinline__ double short_hypot(double x, double y) {
    const double s = __builtin_fma(x, x, y * y);
    const double r = __builtin_amdgcn_rsq(s);
    const double g0 = s * r, h0 = 0.5 * r;
    const double e0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, e0, g0), h1 = __builtin_fma(h0, e0, h0);
    const double d1 = __builtin_fma(-g1, g1, s);
    const double g2 = __builtin_fma(d1, h1, g1);
    return (s > 0.0 && s < __builtin_inf()) ? g2 : s;        // 0, +inf and NaN pass through
}

// Four phases at once: ONE wave-uniform branch on `safe` per group, so that the four dependent fma
// chains sit in one basic block and overlap.
__device__ __forceinline__ void phases4(const double (&t)[4], double period, double y, bool safe,
                                        double (&phi)[4]) {
    double q[4];
    if (safe) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double q0 = t[u] * y;
            const double r0 = __builtin_fma(-period, q0, t[u]);
            const double q1 = __builtin_fma(r0, y, q0);
            const double r1 = __builtin_fma(-period, q1, t[u]);
            q[u] = __builtin_fma(r1, y, q1);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = t[u] / period;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) phi[u] = q[u] - __builtin_floor(q[u]);
}

// MULTI = false: all samples fit one LDS slice (KMAX = samples per thread kept in registers between P1
// and P2, 16-bit indices, NB = 2048).  MULTI = true (kCapacity < N <= 16 slices): one histogram pass over
// all samples (rolled loop) gives the bucket starts and leaves every sample's bucket id in global scratch,
// then the period is worked off in slices of consecutive buckets that fit LDS - the bucket ids read back
// and this slice's samples scattered into order[],
// range table, P3a, P3b - with 32-bit indices and NB = 8192 coarse buckets.
// PL = 1 or 2 (several slices, 65 536 <= N < 131 072 / 262 144): the permutation keeps 16 bits of a sample index per
// entry plus PL bits in planes beside it (an LDS atomic OR per set bit): 2.125 / 2.25 bytes per entry instead of 4,
// i.e. slices of ~45 000 / ~42 700 samples instead of 23 976 - the reference's SunSpots curve (74 326) takes two
// slices instead of four.
// EMIT: besides the length, the sorted curve itself goes to a.sorted - the Supersmoother's sort below 262 144 samples
// (one workgroup sorts a period of 5e4 samples in 69 us: 4096 periods 1.1 ms, where the streamed kernels - built for
// curves that outgrow L2 - took 6.3).  Its own instances: the StringLength instances keep their registers.
template <int KMAX, typename IdxT = unsigned short, int NB = kNB, bool MULTI = false, int PL = 0, bool EMIT = false>
__global__ __launch_bounds__(kBlock) void sl_fast_kernel(FastArgs a) {
    constexpr bool P17 = PL > 0;
    static_assert(PL >= 0 && PL <= 2 && (!P17 || (MULTI && sizeof(IdxT) == 4)), "the bit planes belong to the several-slice instance");
    constexpr int kWaveBytes = FL<IdxT>::wave_bytes;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char *wbuf = lds_raw;                                                  // P3a: per-wave scratch
    unsigned *hist = reinterpret_cast<unsigned *>(lds_raw);                         // P1/P2 alias [NB + 64]
    unsigned long long *bkeys = reinterpret_cast<unsigned long long *>(lds_raw);    // P3b alias [kDCap]
    IdxT *bidx = reinterpret_cast<IdxT *>(bkeys + kDCap);                           // P3b alias [kDCap]
    unsigned short *bndb = reinterpret_cast<unsigned short *>(lds_raw + kWaves * kWaveBytes);
    unsigned short *bnds = bndb + kRanges + 8;
    unsigned *defer = reinterpret_cast<unsigned *>(bnds + kRanges + 8);             // [16]
    IdxT *order = reinterpret_cast<IdxT *>(defer + 16);                             // [slice + 64]
    unsigned short *order16 = reinterpret_cast<unsigned short *>(defer + 16);       // P17: low 16 bits per entry ...
    const int plane_words = P17 ? (a.slice_cap + 64 + 31) / 32 : 0;                 // per plane
    unsigned *plane = reinterpret_cast<unsigned *>(order16 + ((a.slice_cap + 64 + 7) & ~7));   // ... + bits 16 (, 17)
    auto order_get = [&](int pos) -> unsigned {
        if constexpr (P17) {
            unsigned v = (unsigned)order16[pos] | (((plane[pos >> 5] >> (pos & 31)) & 1u) << 16);
            if constexpr (PL > 1) v |= ((plane[plane_words + (pos >> 5)] >> (pos & 31)) & 1u) << 17;
            return v;
        } else {
            return (unsigned)order[pos];
        }
    };
#define PDC_ORDER_GET(pos) order_get(pos)
    __shared__ unsigned wave_tot[kWaves];
    __shared__ double red[kWaves];
    __shared__ unsigned s_fill;
    const int tid0 = threadIdx.x, lane = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int n = (int)a.n;
    unsigned long long *gk = a.gkeys + (int64_t)blockIdx.x * a.n_pad;
    unsigned *gi = a.gidx + (int64_t)blockIdx.x * a.n_pad;
    double *rsum = a.rsum + (int64_t)blockIdx.x * a.nr_pad * 4;
    int *rcnt = a.rcnt + (int64_t)blockIdx.x * a.nr_pad;
    double *rlen = a.rlen + (int64_t)blockIdx.x * a.nr_pad;
    const bool t_safe = a.flags[0] != 0u;

    unsigned long long *keys_w = reinterpret_cast<unsigned long long *>(wbuf + wave * kWaveBytes);
    unsigned *fine_w = reinterpret_cast<unsigned *>(wbuf + wave * kWaveBytes + kRCap * 8);
    IdxT *idx_w = reinterpret_cast<IdxT *>(wbuf + wave * kWaveBytes + kRCap * 8 + (kWFine + 4) * 4);
    unsigned *ghist = MULTI ? a.ghist + (int64_t)blockIdx.x * NB : nullptr;
    unsigned short *gb = MULTI ? a.gbucket + (int64_t)blockIdx.x * a.n_pad * 2 : nullptr;
    static_assert(!MULTI || NB <= 65536, "bucket ids are kept as 16-bit numbers");

    // the usual case behind sl_duo_kernel: nothing was left over - no walk over todo[], no LDS set-up
    if (a.todo && *a.todo_count == 0u) return;
    for (int64_t p = blockIdx.x; p < a.n_periods; p += gridDim.x) {
        if (a.todo && !a.todo[p]) continue;   // (workgroup-uniform)
        if (a.skip && a.skip[p]) continue;    // (workgroup-uniform)
        const double period = a.periods[p];
        const double y = 1.0 / period;
        const bool safe = period_is_safe(period, t_safe);
        double total = 0.0;

        int consumed = 0, r_base = 0, b0 = 0, b1 = NB, slice_n = n, nranges = 0;
        if constexpr (!MULTI) {
        // ---- P1: exact phases, coarse histogram; bucket ids stay in registers --------------------
        // (the thread id goes through an opaque copy once per period: the <= 52 per-sample addresses and
        // bounds masks are loop-invariant, and hoisted out of the period loop they would all be spilled)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        for (int b = tid; b < NB + 64; b += kBlock) hist[b] = 0u;
        if (tid < 16) defer[tid] = tid == 15 ? (unsigned)kWaves : 0u;   // [15]: next range to hand out (P3a)
        __syncthreads();
        // Samples past the end (the last trip of a thread) go to one of 64 dummy buckets behind the
        // histogram instead of being branched around: everything below is straight-line code.
        unsigned pk[(KMAX + 1) / 2];
#pragma unroll
        for (int k = 0; k < (KMAX + 1) / 2; ++k) pk[k] = 0u;
        // (groups of G1 coalesced loads, the next group requested before the current one is folded;
        // the scheduling barrier keeps the compiler from hoisting all <= 52 loads to the top)
        constexpr int G1 = PDC_SL_G1, G2 = PDC_SL_G2;
        static_assert(G1 % 4 == 0, "phases are computed four at a time");
        double tv[G1], tn[G1];
#pragma unroll
        for (int u = 0; u < G1; ++u) {
            const int i = u * kBlock + tid;
            tn[u] = a.t[i < n ? i : n - 1];
        }
#pragma unroll
        for (int k0 = 0; k0 < KMAX; k0 += G1) {
            if (k0 * kBlock < n) {   // workgroup-uniform
#pragma unroll
                for (int u = 0; u < G1; ++u) tv[u] = tn[u];
                if (k0 + G1 < KMAX && (k0 + G1) * kBlock < n) {
#pragma unroll
                    for (int u = 0; u < G1; ++u) {
                        const int i = (k0 + G1 + u) * kBlock + tid;
                        if (k0 + G1 + u < KMAX) tn[u] = a.t[i < n ? i : n - 1];
                    }
                }
                double phi[G1];
#pragma unroll
                for (int q = 0; q < G1; q += 4) {
                    double t4[4] = {tv[q], tv[q + 1], tv[q + 2], tv[q + 3]}, p4[4];
                    phases4(t4, period, y, safe, p4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) phi[q + u] = p4[u];
                }
#pragma unroll
                for (int u = 0; u < G1; ++u) {
                    if (k0 + u < KMAX) {
                        const int i = (k0 + u) * kBlock + tid;
                        const int b = i < n ? coarse_of<NB>(phi[u]) : NB + lane;
                        atomicAdd(&hist[b], 1u);
                        pk[(k0 + u) >> 1] |= (unsigned)b << (((k0 + u) & 1) * 16);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        scan_buckets<NB>(hist, wave_tot);   // hist[b] = first sorted position of bucket b
#if PDC_SL_STOP == 1
        continue;
#endif

        // ---- P2: the permutation, grouped by coarse bucket ----------------------------------------
#pragma unroll
        for (int k0 = 0; k0 < KMAX; k0 += G2) {
            if (k0 * kBlock < n) {   // workgroup-uniform
                unsigned pos[G2];
#pragma unroll
                for (int u = 0; u < G2; ++u) {
                    if (k0 + u < KMAX) {
                        const unsigned b = (pk[(k0 + u) >> 1] >> (((k0 + u) & 1) * 16)) & 0xFFFFu;
                        pos[u] = atomicAdd(&hist[b], 1u);
                    }
                }
#pragma unroll
                for (int u = 0; u < G2; ++u) {
                    if (k0 + u < KMAX) {
                        const int i = (k0 + u) * kBlock + tid;
                        order[i < n ? pos[u] : (unsigned)(n + lane)] = (IdxT)i;
                    }
                }
            }
        }
        } else {
        // ---- P1 (several slices): histogram of ALL samples, then the grouping by coarse bucket in global
        // scratch, once per period -------------------------------------------------------------------------
        int tid = tid0;   // (opaque copy, as above: addresses built from the thread id are not hoisted out of the period loop and spilled)
        asm volatile("" : "+v"(tid));
        for (int b = tid; b < NB; b += kBlock) hist[b] = 0u;
        __syncthreads();
        // (eight coalesced loads per trip, the next trip's requested before this one is folded)
        auto load8 = [&](int i0, double (&v)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * kBlock;
                v[u] = a.t[i < n ? i : n - 1];
            }
        };
        auto fold8 = [&](const double (&v)[8], double (&phi)[8]) {
            double a4[4], p4[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a4[u] = v[4 * h + u];
                phases4(a4, period, y, safe, p4);
#pragma unroll
                for (int u = 0; u < 4; ++u) phi[4 * h + u] = p4[u];
            }
        };
        {
            double nx[8];
            load8(tid, nx);
            for (int i0 = tid; i0 < n; i0 += 8 * kBlock) {
                double tv[8], phi[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tv[u] = nx[u];
                if (i0 + 8 * kBlock < n) load8(i0 + 8 * kBlock, nx);
                fold8(tv, phi);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u * kBlock < n) {
                        const int b = coarse_of<NB>(phi[u]);
                        atomicAdd(&hist[b], 1u);
                        gb[i0 + u * kBlock] = (unsigned short)b;   // (read back once per slice: no second fold)
                    }
            }
        }
        __syncthreads();
        scan_buckets<NB>(hist, wave_tot);   // hist[b] = first sorted position of bucket b
        for (int b = tid; b < NB; b += kBlock) ghist[b] = hist[b];
        __syncthreads();
        }
        int tid = tid0;   // (opaque copy, as above: addresses built from the thread id are not hoisted out of the period loop and spilled)
        asm volatile("" : "+v"(tid));
        do {   // one trip per slice
        asm volatile("" : "+v"(tid));   // (... nor out of the slice loop)
        if constexpr (MULTI) {
            __syncthreads();
            for (int b = tid; b < NB; b += kBlock) hist[b] = ghist[b];
            __syncthreads();
            auto end_of = [&](int b) -> int { return b + 1 < NB ? (int)hist[b + 1] : n; };
            {   // first bucket that still has unconsumed samples
                int l = 0, h = NB - 1;
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    if (end_of(mid) > consumed) h = mid; else l = mid + 1;
                }
                b0 = l;
            }
            const int first_cnt = end_of(b0) - consumed;
            if (first_cnt > a.slice_cap) {
                // a single coarse bucket larger than a slice (clustered phases): sorted in global scratch
                const int cnt = first_cnt;
                int P = 2;
                while (P < cnt) P <<= 1;
                __syncthreads();
                // collect the bucket's samples by folding everything once more
                if (tid == 0) s_fill = 0u;
                __syncthreads();
                for (int i = tid; i < n; i += kBlock) {
                    const double phi = fast_phase(a.t[i], period, y, safe);
                    if (coarse_of<NB>(phi) == b0) {
                        const unsigned slot = atomicAdd(&s_fill, 1u);
                        gk[slot] = phase_key(phi);
                        gi[slot] = (unsigned)i;
                    }
                }
                for (int sI = cnt + tid; sI < P; sI += kBlock) {
                    gk[sI] = ~0ull;
                    gi[sI] = ~0u;
                }
                __syncthreads();
                bitonic_sort<unsigned>(gk, gi, P);
                total += segment_sum(gk, gi, cnt, a.m);
                if (EMIT) {
                    rec_t *const row = a.sorted + p * a.n;
                    for (int s = tid; s < cnt; s += kBlock) {
                        rec_t o;
                        o.x = __longlong_as_double((long long)gk[s]);
                        o.y = a.m[gi[s]];
                        row[consumed + s] = o;
                    }
                }
                if (tid == 0) {
                    rsum[(int64_t)r_base * 4 + 0] = __longlong_as_double((long long)gk[0]);
                    rsum[(int64_t)r_base * 4 + 1] = a.m[gi[0]];
                    rsum[(int64_t)r_base * 4 + 2] = __longlong_as_double((long long)gk[cnt - 1]);
                    rsum[(int64_t)r_base * 4 + 3] = a.m[gi[cnt - 1]];
                    rcnt[r_base] = cnt;
                    rlen[r_base] = 0.0;
                }
                __syncthreads();
                r_base += 1;
                consumed += cnt;
                continue;
            }
            {   // as many consecutive buckets as fit the slice
                int l = b0 + 1, h = NB;
                while (l < h) {
                    const int mid = (l + h + 1) >> 1;
                    if (end_of(mid - 1) - consumed <= a.slice_cap) l = mid; else h = mid - 1;
                }
                b1 = l;
            }
            slice_n = end_of(b1 - 1) - consumed;
            __syncthreads();   // every thread has read what it needs from the start offsets
            if (tid < 16) defer[tid] = tid == 15 ? (unsigned)kWaves : 0u;
            if constexpr (P17) {
                for (int x = tid; x < PL * plane_words; x += kBlock) plane[x] = 0u;
                __syncthreads();
            }
            {
                // hist[b] still holds the START of every bucket: every sample's bucket id is read back (2
                // bytes, coalesced) and this slice's samples go to order[].  (Grouping all samples once in
                // global scratch instead costs a random 4-byte store per sample and period - 1.05e11/s, 2.6x
                // slower than a random load; folding every sample again per slice - the first version of this
                // path - cost as much as the ranges from five slices on.)
                for (int i0 = tid; i0 < n; i0 += 8 * kBlock) {
                    unsigned short bk[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * kBlock;
                        bk[u] = gb[i < n ? i : n - 1];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * kBlock;
                        const int b = (int)bk[u];
                        if (i < n && b >= b0 && b < b1) {
                            const unsigned pos = atomicAdd(&hist[b], 1u) - (unsigned)consumed;
                            if constexpr (P17) {
                                order16[pos] = (unsigned short)i;
                                if ((i >> 16) & 1) atomicOr(&plane[pos >> 5], 1u << (pos & 31));
                                if constexpr (PL > 1)
                                    if ((i >> 17) & 1) atomicOr(&plane[plane_words + (pos >> 5)], 1u << (pos & 31));
                            } else {
                                order[pos] = (IdxT)i;
                            }
                        }
                    }
                }
            }
        }
        nranges = (slice_n + kFWin - 1) / kFWin;
        __syncthreads();   // hist[b] = END position of bucket b
        // range r starts at the first bucket whose start position is >= r * kWin
        for (int r = tid; r <= nranges; r += kBlock) {
            int b = b0, s0 = 0;
            if (r > 0) {
                const unsigned x = (unsigned)consumed + (unsigned)r * kFWin;
                int l = b0, h = b1;   // smallest j in [b0, b1) with end(j) >= x, b1 if none
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    if (hist[mid] >= x) h = mid; else l = mid + 1;
                }
                b = l < b1 ? l + 1 : b1;
                s0 = l < b1 ? (int)hist[l] - consumed : slice_n;
            }
            bndb[r] = (unsigned short)b;
            bnds[r] = (unsigned short)s0;
        }
        __syncthreads();   // (the histogram is dead from here on: its LDS becomes wave scratch)

#if PDC_SL_STOP == 2
        continue;
#endif
        // ---- P3a: wave-autonomous ranges -------------------------------------------------------------
        constexpr bool kEmitSorted = EMIT;
        rec_t *const emit_row = EMIT ? a.sorted + p * a.n : nullptr;
        const int emit_at = consumed;
#include "sl_ranges.inc"
        if (wave < nranges) request(wave);
        for (int r = wave; r < nranges; r = r_next) {
            if (n_cnt > 192) process(r, std::true_type{});
            else process(r, std::false_type{});
        }
        __syncthreads();

        // ---- P3b: deferred ranges, whole workgroup ---------------------------------------------------
        for (int w32 = 0; w32 < (nranges + 31) / 32; ++w32) {
            unsigned bits = defer[w32];
            while (bits) {
                const int r = w32 * 32 + __builtin_ctz(bits);
                bits &= bits - 1;
                const int s_lo = bnds[r], cnt = (int)bnds[r + 1] - s_lo;
                int P = 2;
                while (P < cnt) P <<= 1;
                double p0, m0, p1, m1;
                if (cnt <= kDCap) {
                    for (int s = tid; s < P; s += kBlock) {
                        if (s < cnt) {
                            const IdxT id = (IdxT)PDC_ORDER_GET(s_lo + s);
                            bkeys[s] = phase_key(fast_phase(a.t[id], period, y, safe));
                            bidx[s] = id;
                        } else {
                            bkeys[s] = ~0ull;
                            bidx[s] = (IdxT)~0u;
                        }
                    }
                    __syncthreads();
                    bitonic_sort<IdxT>(bkeys, bidx, P);
                    total += segment_sum(bkeys, bidx, cnt, a.m);
                    if (EMIT)
                        for (int s = tid; s < cnt; s += kBlock) {
                            rec_t o;
                            o.x = __longlong_as_double((long long)bkeys[s]);
                            o.y = a.m[bidx[s]];
                            emit_row[consumed + s_lo + s] = o;
                        }
                    p0 = __longlong_as_double((long long)bkeys[0]);
                    m0 = a.m[bidx[0]];
                    p1 = __longlong_as_double((long long)bkeys[cnt - 1]);
                    m1 = a.m[bidx[cnt - 1]];
                } else {
                    for (int s = tid; s < P; s += kBlock) {
                        if (s < cnt) {
                            const unsigned id = PDC_ORDER_GET(s_lo + s);
                            gk[s] = phase_key(fast_phase(a.t[id], period, y, safe));
                            gi[s] = id;
                        } else {
                            gk[s] = ~0ull;
                            gi[s] = ~0u;
                        }
                    }
                    __syncthreads();
                    bitonic_sort<unsigned>(gk, gi, P);
                    total += segment_sum(gk, gi, cnt, a.m);
                    if (EMIT)
                        for (int s = tid; s < cnt; s += kBlock) {
                            rec_t o;
                            o.x = __longlong_as_double((long long)gk[s]);
                            o.y = a.m[gi[s]];
                            emit_row[consumed + s_lo + s] = o;
                        }
                    p0 = __longlong_as_double((long long)gk[0]);
                    m0 = a.m[gi[0]];
                    p1 = __longlong_as_double((long long)gk[cnt - 1]);
                    m1 = a.m[gi[cnt - 1]];
                }
                if (tid == 0) {
                    const int64_t g = r_base + r;
                    rsum[g * 4 + 0] = p0;
                    rsum[g * 4 + 1] = m0;
                    rsum[g * 4 + 2] = p1;
                    rsum[g * 4 + 3] = m1;
                    rcnt[g] = cnt;
                    rlen[g] = 0.0;   // (its segments were added to `total` by the whole workgroup, in a fixed order)
                }
                __syncthreads();
            }
        }
        r_base += nranges;
        consumed += slice_n;
        } while (MULTI && consumed < n);
        const int nr_all = r_base;
        __syncthreads();  // every summary of this period (global, this workgroup's) is visible

        // ---- P3c: links between consecutive non-empty ranges + the closing segment ----------------
        for (int r = tid; r < nr_all; r += kBlock) {
            if (rcnt[r] > 0) {
                total += rlen[r];
                int q = r - 1;
                while (q >= 0 && rcnt[q] == 0) --q;
                if (q >= 0)
                    total += hypot(rsum[(int64_t)r * 4 + 1] - rsum[(int64_t)q * 4 + 3],
                                   rsum[(int64_t)r * 4 + 0] - rsum[(int64_t)q * 4 + 2]);
            }
        }
        if (tid == 0 && nr_all > 0) {
            int f0 = 0, l0 = nr_all - 1;
            while (f0 < nr_all && rcnt[f0] == 0) ++f0;
            while (l0 >= 0 && rcnt[l0] == 0) --l0;
            // closing segment of np.roll(-1): first minus last, no phase wrap (phase.py:50)
            if (f0 < nr_all && l0 >= 0)
                total += hypot(rsum[(int64_t)f0 * 4 + 1] - rsum[(int64_t)l0 * 4 + 3],
                               rsum[(int64_t)f0 * 4 + 0] - rsum[(int64_t)l0 * 4 + 2]);
        }
        total = wave_sum(total);
        if (lane == 0) red[wave] = total;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < kWaves; ++w) sum += red[w];
            a.ell[p] = sum;
        }
        __syncthreads();
    }
}

#undef PDC_ORDER_GET

// AoS (t, m) table + the "every t is tame" flag (one workgroup; N <= 52k)
__global__ __launch_bounds__(kBlock) void sl_prep_kernel(const double *t, const double *m, int n,
                                                         rec_t *rec, unsigned *flags) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const double tv = t[i];
        rec_t v;
        v.x = tv;
        v.y = m[i];
        rec[i] = v;
        const double at = __builtin_fabs(tv);
        if (!(at == 0.0 || (at >= 1e-150 && at <= 1e150))) mine = 1;
    }
    if (mine) atomicOr(&bad, 1);
    __syncthreads();
    if (threadIdx.x == 0) flags[0] = bad ? 0u : 1u;
}

}  // namespace fast


// =====================================================================================================
// Two workgroups per CU ("duo"), for N <= 26 048 samples: while a workgroup folds its samples and builds the
// permutation (P1 + P2 + range table) the gather pipe of its CU idles, and the ranges (P3a) leave a third of
// the VALU slots empty.  Here a workgroup is 512 threads and takes HALF of LDS, so two are resident per CU,
// each on its own period (handed out by a global ticket counter): one's P1/P2 runs under the other's ranges,
// and a short period no longer pays for 16 waves' worth of barriers.  Same algorithm, same arithmetic and
// summation order per range as sl_fast_kernel.  Measured against it on one box (tools/sl_shapes.py):
// N = 25 000 x 1e5 periods 15.07 -> 13.24 ms, 10 000 x 2e4 1.81 -> 1.25, 2000 x 1e5 3.99 -> 2.32, 500 x 1e5
// 3.20 -> 1.80 ms.  Above one half-LDS permutation (C5's N = 5e4) the one-workgroup kernel stays: a period cut
// into two phase halves, one per workgroup, was built and measured (31.2 against 27.3 ms) - each half has to
// fold ALL samples to find its own, and the range phase is latency-bound per wave (16 waves per CU either
// way), so the doubled P1 is not hidden.
// A deferred range beyond the LDS sort (heavily clustered phases) marks the period in todo[], and the
// one-workgroup kernel works off the marked periods afterwards.
namespace duo {
using namespace fast;

constexpr int kB = 512;
constexpr int kW = kB / 64;
constexpr int kLdsWg = kLdsTotal / 2;   // two workgroups per CU
constexpr int kWaveB = Lds<unsigned short>::wave_bytes;
constexpr int kRangesD = 136;           // >= kCapD / kFWin + 2
constexpr int kStaticD = 256;           // static __shared__ below, rounded up
constexpr int kFixedD = kW * kWaveB + 2 * (kRangesD + 8) * 2 + 64;
constexpr int kCapD = ((((kLdsWg - kFixedD - kStaticD) / 2) & ~7) - 64);   // samples per period (+ 64 dummy slots)
static_assert(kCapD / kFWin + 2 <= kRangesD, "range table too small");
static_assert(kW * kWaveB >= (kNB + 64) * 4 && kW * kWaveB >= kDCap * 10, "aliases must fit");
static_assert(kRangesD <= 15 * 32, "deferred-range bits live in defer[0..14]");
constexpr int kCapQ = 4096;             // samples per period of the 256-thread instance (four workgroups per CU)

struct DuoArgs {
    const double *t, *m, *periods;
    const rec_t *rec;
    const unsigned *flags;      // [0] != 0: every t is 0 or 1e-150 <= |t| <= 1e150
    int n;
    int64_t n_periods;
    unsigned *ticket;           // [0] next period, [1] periods marked in todo[] (both zeroed before the launch)
    double *ell;
    unsigned char *todo;        // [n_periods]: 1 = left to the one-workgroup kernel
    const unsigned char *skip;  // NULL, or [n_periods]: periods with a non-zero entry are done already (one cycle)
    double *rsum;               // [grid][nr_pad][4]
    int *rcnt;                  // [grid][nr_pad]
    double *rlen;               // [grid][nr_pad]
    int64_t nr_pad;
};

// BLK = 512: two workgroups per CU (N <= kCapD); BLK = 256: four (N <= kCapQ - short periods have too few
// ranges for eight waves: 10 at N = 2000, 3 at N = 500)
// NBL: coarse buckets (512 suffice for the short periods of the 256-thread instance: their zeroing, scan and
// binary searches are fixed work per period)
template <int KMAX, int BLK = duo::kB, int NBL = kNB>
__global__ __launch_bounds__(BLK, 4) void sl_duo_kernel(DuoArgs a) {
    constexpr int kB = BLK, kW = BLK / 64;                 // (shadow the namespace's 512-thread values)
    constexpr int kDCap = BLK >= 512 ? ::kDCap : 1024;     // deferred LDS sort: keys + indices alias the wave scratch
    static_assert(kW * kWaveB >= (NBL + 64) * 4 && kW * kWaveB >= kDCap * 10, "aliases must fit");
    typedef unsigned short IdxT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned char *wbuf = lds_raw;                                                  // P3a: per-wave scratch
    unsigned *hist = reinterpret_cast<unsigned *>(lds_raw);                         // P1/P2 alias [NBL + 64]
    unsigned long long *bkeys = reinterpret_cast<unsigned long long *>(lds_raw);    // P3b alias [kDCap]
    IdxT *bidx = reinterpret_cast<IdxT *>(bkeys + kDCap);                           // P3b alias [kDCap]
    unsigned short *bndb = reinterpret_cast<unsigned short *>(lds_raw + kW * kWaveB);
    unsigned short *bnds = bndb + kRangesD + 8;
    unsigned *defer = reinterpret_cast<unsigned *>(bnds + kRangesD + 8);            // [16]
    IdxT *order = reinterpret_cast<IdxT *>(defer + 16);                             // [cap + 64]
    __shared__ unsigned wave_tot[kW];
    __shared__ double red[kW];
    __shared__ unsigned s_item, s_bad;
    const int tid0 = threadIdx.x, lane = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int n = a.n;
    double *rsum = a.rsum + (int64_t)blockIdx.x * a.nr_pad * 4;
    int *rcnt = a.rcnt + (int64_t)blockIdx.x * a.nr_pad;
    double *rlen = a.rlen + (int64_t)blockIdx.x * a.nr_pad;
    const bool t_safe = a.flags[0] != 0u;
    const unsigned n_items = (unsigned)a.n_periods;

    unsigned long long *keys_w = reinterpret_cast<unsigned long long *>(wbuf + wave * kWaveB);
    unsigned *fine_w = reinterpret_cast<unsigned *>(wbuf + wave * kWaveB + kRCap * 8);
    IdxT *idx_w = reinterpret_cast<IdxT *>(wbuf + wave * kWaveB + kRCap * 8 + (kWFine + 4) * 4);

    for (;;) {
        __syncthreads();   // the previous item is done with LDS
        if (tid0 == 0) {
            s_item = atomicAdd(a.ticket, 1u);
            s_bad = 0u;
        }
        __syncthreads();
        const unsigned item = s_item;
        if (item >= n_items) break;
        const int64_t p = (int64_t)item;
        if (a.skip && a.skip[p]) continue;   // (workgroup-uniform)
        const double period = a.periods[p];
        const double y = 1.0 / period;
        const bool safe = period_is_safe(period, t_safe);
        double total = 0.0;

        // ---- P1: exact phases, coarse histogram; bucket ids in registers ------------------------------
        // (as in sl_fast_kernel: opaque copy of the thread id, dummy buckets for samples past the end)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        for (int b = tid; b < NBL + 64; b += kB) hist[b] = 0u;
        if (tid < 16) defer[tid] = tid == 15 ? (unsigned)kW : 0u;   // [15]: next range to hand out (P3a)
        __syncthreads();
        unsigned pk[(KMAX + 1) / 2];
#pragma unroll
        for (int k = 0; k < (KMAX + 1) / 2; ++k) pk[k] = 0u;
        double tv[4], tn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = u * kB + tid;
            tn[u] = a.t[i < n ? i : n - 1];
        }
#pragma unroll
        for (int k0 = 0; k0 < KMAX; k0 += 4) {
            if (k0 * kB < n) {   // workgroup-uniform
#pragma unroll
                for (int u = 0; u < 4; ++u) tv[u] = tn[u];
                if (k0 + 4 < KMAX && (k0 + 4) * kB < n) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = (k0 + 4 + u) * kB + tid;
                        tn[u] = a.t[i < n ? i : n - 1];
                    }
                }
                double phi[4];
                phases4(tv, period, y, safe, phi);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u < KMAX) {
                        const int i = (k0 + u) * kB + tid;
                        const int b = i < n ? coarse_of<NBL>(phi[u]) : NBL + lane;
                        atomicAdd(&hist[b], 1u);
                        pk[(k0 + u) >> 1] |= (unsigned)b << (((k0 + u) & 1) * 16);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        scan_buckets<NBL, kB>(hist, wave_tot);   // hist[b] = first sorted position of bucket b
        const int slice_n = n;

        // ---- P2: the permutation, grouped by coarse bucket ----------------------------------------
#pragma unroll
        for (int k0 = 0; k0 < KMAX; k0 += 4) {
            if (k0 * kB < n) {   // workgroup-uniform
                unsigned pos[4], bb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u < KMAX) {
                        bb[u] = (pk[(k0 + u) >> 1] >> (((k0 + u) & 1) * 16)) & 0xFFFFu;
                        pos[u] = atomicAdd(&hist[bb[u]], 1u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u < KMAX) {
                        const int i = (k0 + u) * kB + tid;
                        order[i < n ? pos[u] : (unsigned)(n + lane)] = (IdxT)i;
                    }
                }
            }
        }
        const int nranges = (slice_n + kFWin - 1) / kFWin;
        __syncthreads();   // hist[b] = END position of bucket b
        // range r starts at the first bucket whose start position is >= r * kFWin
        for (int r = tid; r <= nranges; r += kB) {
            int b = 0, s0 = 0;
            if (r > 0) {
                const unsigned x = (unsigned)r * kFWin;
                int l = 0, h = NBL;   // smallest j in [0, NBL) with end(j) >= x, NBL if none
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    if (hist[mid] >= x) h = mid; else l = mid + 1;
                }
                b = l < NBL ? l + 1 : NBL;
                s0 = l < NBL ? (int)hist[l] : slice_n;
            }
            bndb[r] = (unsigned short)b;
            bnds[r] = (unsigned short)s0;
        }
        __syncthreads();   // (the histogram is dead from here on: its LDS becomes wave scratch)

        // ---- P3a: wave-autonomous ranges (the same code as sl_fast_kernel's, on this workgroup's 8 waves) --
        constexpr int NB = NBL;
        const int r_base = 0;
#define PDC_ORDER_GET(pos) ((unsigned)order[pos])
        constexpr bool kEmitSorted = false;
        rec_t *const emit_row = nullptr;
        const int emit_at = 0;
#include "sl_ranges.inc"
#undef PDC_ORDER_GET
        if (wave < nranges) request(wave);
        for (int r = wave; r < nranges; r = r_next) {
            if (n_cnt > 192) process(r, std::true_type{});
            else process(r, std::false_type{});
        }
        __syncthreads();

        // ---- P3b: deferred ranges, whole workgroup (LDS sort; larger ones mark the period) ----------
        for (int w32 = 0; w32 < (nranges + 31) / 32; ++w32) {
            unsigned bits = defer[w32];
            while (bits) {
                const int r = w32 * 32 + __builtin_ctz(bits);
                bits &= bits - 1;
                const int s_lo = bnds[r], cnt = (int)bnds[r + 1] - s_lo;
                if (cnt > kDCap) {      // (workgroup-uniform)
                    if (tid == 0) s_bad = 1u;
                    continue;
                }
                int P = 2;
                while (P < cnt) P <<= 1;
                for (int sI = tid; sI < P; sI += kB) {
                    if (sI < cnt) {
                        const IdxT id = order[s_lo + sI];
                        bkeys[sI] = phase_key(fast_phase(a.t[id], period, y, safe));
                        bidx[sI] = id;
                    } else {
                        bkeys[sI] = ~0ull;
                        bidx[sI] = (IdxT)~0u;
                    }
                }
                __syncthreads();
                bitonic_sort<IdxT>(bkeys, bidx, P);
                total += segment_sum(bkeys, bidx, cnt, a.m);
                if (tid == 0) {
                    rsum[(int64_t)r * 4 + 0] = __longlong_as_double((long long)bkeys[0]);
                    rsum[(int64_t)r * 4 + 1] = a.m[bidx[0]];
                    rsum[(int64_t)r * 4 + 2] = __longlong_as_double((long long)bkeys[cnt - 1]);
                    rsum[(int64_t)r * 4 + 3] = a.m[bidx[cnt - 1]];
                    rcnt[r] = cnt;
                    rlen[r] = 0.0;   // (its segments were added to `total` by the whole workgroup, in a fixed order)
                }
                __syncthreads();
            }
        }
        __syncthreads();  // every summary of this item (global, this workgroup's) is visible

        // ---- P3c: links between consecutive non-empty ranges + the closing segment -----------------------
        for (int r = tid; r < nranges; r += kB) {
            if (rcnt[r] > 0) {
                total += rlen[r];
                int q = r - 1;
                while (q >= 0 && rcnt[q] == 0) --q;
                if (q >= 0)
                    total += hypot(rsum[(int64_t)r * 4 + 1] - rsum[(int64_t)q * 4 + 3],
                                   rsum[(int64_t)r * 4 + 0] - rsum[(int64_t)q * 4 + 2]);
            }
        }
        if (tid == 0 && nranges > 0) {
            int f0 = 0, l0 = nranges - 1;
            while (f0 < nranges && rcnt[f0] == 0) ++f0;
            while (l0 >= 0 && rcnt[l0] == 0) --l0;
            // closing segment of np.roll(-1): first minus last, no phase wrap (phase.py:50)
            if (f0 < nranges && l0 >= 0)
                total += hypot(rsum[(int64_t)f0 * 4 + 1] - rsum[(int64_t)l0 * 4 + 3],
                               rsum[(int64_t)f0 * 4 + 0] - rsum[(int64_t)l0 * 4 + 2]);
        }
        total = wave_sum(total);
        if (lane == 0) red[wave] = total;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < kW; ++w) sum += red[w];
            a.todo[p] = s_bad ? 1 : 0;
            if (s_bad) atomicAdd(&a.ticket[1], 1u);
            else a.ell[p] = sum;
        }
    }
}

}  // namespace duo

// =====================================================================================================
// Streamed path ("stream"), for light curves whose (t, m) table no longer fits an XCD's L2 (N > ~2.4e5):
// there every random gather of the kernels above misses L2 (62 ps per pair at N = 1e6 against 5.5 at C5).
// Here NOTHING is gathered: the samples of a period are counting-sorted by phase bin through global scratch
// with coalesced stores, and every bin is then sorted in LDS.  Per batch of trial periods, five launches:
//   sl_hist_kernel   workgroup (period, group g of W; a group = a contiguous run of tiles of 2048 samples): folds
//                    the group's samples exactly as the other kernels do and leaves ITS histogram over 4096 coarse
//                    phase buckets (LDS atomics, no global ones);
//   sl_lut_kernel    one workgroup per period: adds the W histograms, cuts [0, 1] into bins of consecutive coarse
//                    buckets that each hold just under 6144 samples - whatever the phase distribution (periods
//                    beyond the baseline fill only part of [0, 1]) - and, from the groups' own histograms, where
//                    every group's records start in every bin's list: the lists are packed exactly, nothing can
//                    overflow; a coarse bucket too heavy for a bin (clustered phases: evenly sampled data at a
//                    commensurate period) marks the period for the general kernel;
//   sl_part_kernel   workgroup (period, group): folds again, tile by tile (the next tile's samples requested
//                    before the current one is worked off), groups a tile's records by bin IN LDS and appends
//                    every group as one contiguous run of (phase, m) + sample index to its place in the bin's
//                    list - coalesced stores, no global atomics, consecutive tiles extend the same cache lines;
//   sl_sort_kernel   persistent, one workgroup per CU over the (period, bin) items: loads a bin's records
//                    (contiguous), ranks them in LDS - 8192 fine buckets, arrival ranks from packed 16-bit LDS
//                    counters, exclusive scan, then every record counts the members of its fine bucket that sort
//                    before it, by (64-bit phase pattern, sample index) = numpy's stable order - and adds the
//                    segments of its bin in sorted order (fixed summation order: bitwise reproducible whatever
//                    order the atomics delivered the records in);
//   sl_link_kernel   one wave per period: the bins' lengths in bin order, the links between consecutive
//                    non-empty bins and the closing segment (phase.py:50, not phase-wrapped).
// Traffic: 20 bytes written and read per (sample, period) pair, sequential - HBM-bound at ~40 B per pair.
//
// SLICES MODE (t non-decreasing, checked on the device): the samples of one cycle of the period are consecutive AND
// in phase order, so the samples of (cycle c, bin b) are ONE SLICE of t[] / m[] and nothing has to be partitioned:
//   sl_bound_kernel  in place of sl_part_kernel: a table of the first sample of every (cycle, bin) cell, written
//                    where key(i) = cycle * bins + bin steps up along i (one word per cell, t read once);
//   sl_sort_kernel   fetches a bin's records as its K slices (prefix over their lengths in LDS, one search per wave,
//                    then a walk) and computes the phases itself - from there on the same kernel: the two modes agree
//                    bit for bit;
//   sl_direct_kernel periods of ONE cycle: phase order = sample order, the segments are summed as the samples stand.
// The bin table kernel picks the mode per period (cells of >= 4 samples on average, < 1024 cycles); periods that miss
// it, or samples in any order, take the lists above.  PDC_SL_SLICES=0 switches the mode off (A/B, tests).
namespace stream {
using namespace fast;

#ifndef PDC_SL_CAP
#define PDC_SL_CAP 6144
#define PDC_SL_BB 1024
#define PDC_SL_FINE 8192
#define PDC_SL_MINFILL 2700
#endif
constexpr int kTA = 2048;          // samples per partition tile
constexpr int kBA = 512;           // threads of the histogram / partition kernels (4 samples each per tile)
constexpr int kNC = 4096;          // coarse phase buckets of the histogram and the bin table
constexpr int kS1Max = 2048;       // bins per period at most (5.5 M samples: ~1350 to a coarse bucket - twice that and the
                                   // buckets themselves crowd the bins' capacity)
constexpr int kCap = PDC_SL_CAP;   // records a bin's list (and the sort kernel's LDS) holds
constexpr int kSlack = 16;         // a period's bins are filled to kCap - kSlack - (its heaviest coarse bucket) ...
constexpr int kMinFill = PDC_SL_MINFILL;   // ... and not below this: heavier coarse buckets (clustered phases) go to the general kernel
constexpr int kBB = PDC_SL_BB;     // threads of the sort kernel
constexpr int kPerB = kCap / kBB;  // records per thread
constexpr int kFineB = PDC_SL_FINE; // fine buckets per bin, 16-bit counters packed two to a word (measured, slices mode, N = 1e6 x
                                    // 2048: 8192 buckets 24.5 ms, 4096 25.9, 16384 with bins of 5120 25.4)
static_assert(kCap % kBB == 0 && kCap < 65536, "slice positions are 16-bit");
static_assert(kTA == 4 * kBA && kTA < 65536, "four samples per thread");
constexpr size_t lds_part(int s1p) { return (size_t)kTA * (8 + 8 + 2) + (size_t)kNC * 2 + (size_t)s1p * 4 * 4; }
constexpr int kCycS = 1024;        // slices mode: entries per row of a period's boundary table = cycles the period may span + 1
constexpr int kMinSlice = 4;       // slices mode: samples per (cycle, bin) cell on average, at least (measured, N = 1e6 x 8192
                                   // periods: 16 -> 135 ms, 8 -> 116, 4 -> 111.5; short slices still beat the lists)
constexpr int kDirectW = 32;        // workgroups per one-cycle period in sl_direct_kernel
constexpr unsigned kFlagDirect = 16u; // flag[] value: one cycle, summed as the samples stand
constexpr int kBatchMax = 768;     // periods per batch at most (the sort kernel keeps a prefix over them in LDS)
constexpr size_t kLdsB = (size_t)kCap * (8 + 8 + 4) + (size_t)kFineB * 2;

struct StreamArgs {
    const double *t, *m, *periods;
    int64_t n;
    int64_t p0;                 // first period of this batch
    int batch;                  // periods in this batch
    int s1;                     // bins reserved per period (the table may use fewer)
    int slices;                 // != 0: periods whose cells are long enough take the slices mode (PDC_SL_SLICES=0: none)
    int no_lists;               // != 0: the workspace holds no lists (the host saw that every period takes the slices or the
                                //       one-cycle mode): a period that does not after all goes to the general kernel
    int direct;                 // != 0: periods that outlast the samples are summed as the samples stand (not when the
                                //       sorted curve itself is wanted: Supersmoother)
    int groups;                 // W: workgroups per period in the histogram / partition kernels
    int tiles_w;                // tiles of kTA samples per group
    const unsigned *bad_t;      // [0] != 0: some |t| outside {0} u [1e-150, 1e150]; [1] != 0: t is not non-decreasing
    const rec_t *tm;            // [n] (t, m) side by side, written by sl_tame_kernel: the sort kernel's slices come as
                                //     one 16-byte load per record
    unsigned *hist;             // [batch][W][kNC]   the groups' histograms
    unsigned short *lut;        // [batch][kNC]      coarse bucket -> bin
    unsigned short *clo;        // [batch][s1 + 1]   first coarse bucket of every bin
    unsigned *boff;             // [batch][s1][W]    where group w's records start in the bin's list (exact: from
                                //                   the groups' own histograms - no slack, no overflow)
    unsigned *bcnt;             // [batch][s1]       records of every bin
    unsigned *bstart;           // [batch][s1]       sorted position of every bin's first record
    unsigned *nbins;            // [batch]           bins the period's table uses (0: left to the general kernel)
    int *ncyc;                  // [batch]           slices mode: cycles the samples span at this period (0: lists mode)
    double *cyc0;               // [batch]           slices mode: floor(t[0] / period)
    unsigned *bnd;              // [batch][s1][kCycS] slices mode: first sample of cell (cycle c, bin b) at [b][c]; cell
                                //                   (c, b) ends where (c, b + 1) - or (c + 1, 0) - starts; [0][K] = n
    rec_t *sorted;              // NULL, or [batch][n]: every period's (phase, m) in sorted order (Supersmoother)
    unsigned *flag;             // [batch]           != 0: left to the general kernel
    rec_t *pm;                  // [batch][s1][kCap] (phase, m)
    unsigned *ix;               // [batch][s1][kCap] sample index (ties)
    double *ssum;               // [batch][s1][4]    first / last (phase, m) of every bin in sorted order
    double *slen;               // [batch][s1]       string length inside every bin
    double *dpart;              // [batch][kDirectW] one-cycle periods: partial sums of sl_direct_kernel's workgroups
    double *ell;
    unsigned char *todo;        // [n_periods]       1 = left to the general kernel
    unsigned *todo_count;
};

__global__ __launch_bounds__(256) void sl_tame_kernel(const double *t, const double *m, rec_t *tm, double *t_copy, double *m_copy,
                                                      int64_t n, unsigned *bad) {
    bool mine = false, unsorted = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (tm) {
            rec_t r;
            r.x = t[i];
            r.y = m[i];
            tm[i] = r;
        }
        if (t_copy) {   // (samples that may be in any order: the time sort works on copies)
            t_copy[i] = t[i];
            m_copy[i] = m[i];
        }
        const double at = __builtin_fabs(t[i]);
        mine = mine || !(at == 0.0 || (at >= 1e-150 && at <= 1e150));
        unsorted = unsorted || (i > 0 && !(t[i - 1] <= t[i]));   // (NaN counts as unsorted)
    }
    if (__any(mine) && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
    if (__any(unsorted) && (threadIdx.x & 63) == 0) atomicOr(bad + 1, 1u);
}

// sample index of local position l of tile kappa of group w: every group owns a contiguous run of tiles_w tiles
__device__ __forceinline__ int64_t sample_of(int64_t kappa, int l, int w, int tiles_w) {
    return ((int64_t)w * tiles_w + kappa) * kTA + l;
}

// Workgroup -> (period q, group w) for the kernels that stream a group's samples (histogram, partition, boundaries).
// Workgroups go to the eight XCDs round robin by their index, and every XCD has an L2 of its own: with the group a
// function of the XCD (w = xcd mod W, W <= 8 groups) an L2 only ever sees its group's share of t[] (and m[]) - 2 MB of
// t at N = 1e6 with four groups - and every period after the first reads it from there.  (Numbered period-major the
// workgroups of all groups ran side by side on every XCD and t came from the Infinity Cache again and again.)
__device__ __forceinline__ bool period_and_group(const StreamArgs &a, int &q, int &w) {
    if (a.groups > 8) {
        q = (int)(blockIdx.x % (unsigned)a.batch);
        w = (int)(blockIdx.x / (unsigned)a.batch);
        return true;
    }
    const int per_group = 8 / a.groups, xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
    w = xcd % a.groups;
    q = j * per_group + xcd / a.groups;
    return q < a.batch;
}

// One LDS add per RUN of equal buckets (round 6).  A thread folds four CONSECUTIVE samples, consecutive lanes the next
// four: time-ordered samples arrive in phase order inside a cycle, and at the periods this path exists for (thousands
// of samples per cycle) a thread's four buckets are one value and a wave's 64 values a handful of runs - where 256
// `ds_add_u32` on one or two addresses serialised (r05 PMC: 95 % of the kernel's LDS-active cycles were bank
// conflicts).  Thread level: four equal buckets become one weight-4 value.  Wave level: a lane heads a run when the
// lane below holds another value (DPP wave_shr:1, lane 0 always); the run's length is the distance to the next head
// in the ballot; one add of 4 x length per run.  A thread whose four buckets differ (a bucket edge, a cycle wrap,
// short periods, samples in any order) adds its own runs and stands out of the wave's: any order stays correct.
// `bucket` = ~0u: nothing to add.
__device__ __forceinline__ void hist_add_runs(unsigned *h, unsigned bucket, unsigned weight, int lane) {
    const unsigned below = PDC_DPP(~bucket, bucket, 0x138, 0xf);
    const bool head = below != bucket;
    const unsigned long long heads = __ballot(head);
    if (head && bucket != ~0u) {
        const unsigned long long rest = (heads >> lane) >> 1;
        const unsigned len = rest ? (unsigned)__builtin_ctzll(rest) + 1u : 64u - (unsigned)lane;
        atomicAdd(&h[bucket], len * weight);
    }
}

// (Two periods per workgroup - every sample loaded once for both - was built and measured: 244 against 236 us per 256
// periods at N = 1e6; the kernel is not short of L2 bandwidth.)
__global__ __launch_bounds__(kBA) void sl_hist_kernel(StreamArgs a) {
    __shared__ unsigned h[kNC];
    const int tid = threadIdx.x;
    int q, w;
    if (!period_and_group(a, q, w)) return;               // (workgroup-uniform)
    const double period = a.periods[a.p0 + q];
    const double y = 1.0 / period;
    const bool safe = period_is_safe(period, a.bad_t[0] == 0u);
    for (int c = tid; c < kNC; c += kBA) h[c] = 0u;
    __syncthreads();
    const int64_t g0 = (int64_t)w * a.tiles_w * kTA;      // the group's first sample (the same run of tiles as before)
    for (int64_t kappa = 0; kappa < a.tiles_w; ++kappa) {
        const int64_t i0 = g0 + kappa * kTA + (int64_t)tid * 4;   // four consecutive samples per thread
        if (g0 + kappa * kTA >= a.n) break;                // (workgroup-uniform)
        double tv[4], phi[4];
        if (i0 + 4 <= a.n) {   // two 16-byte loads (8-byte aligned: t is the caller's pointer)
            typedef double pair_t __attribute__((ext_vector_type(2), aligned(8)));
            const pair_t t01 = *reinterpret_cast<const pair_t *>(a.t + i0), t23 = *reinterpret_cast<const pair_t *>(a.t + i0 + 2);
            tv[0] = t01.x; tv[1] = t01.y; tv[2] = t23.x; tv[3] = t23.y;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) tv[u] = a.t[i0 + u < a.n ? i0 + u : a.n - 1];
        }
        phases4(tv, period, y, safe, phi);
        unsigned b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = i0 + u < a.n ? (unsigned)coarse_of<kNC>(phi[u]) : ~0u;
        const bool same = b[0] == b[1] && b[1] == b[2] && b[2] == b[3];
        hist_add_runs(h, same ? b[0] : ~0u, 4u, tid & 63);
        if (!same) {
            unsigned cnt = 1u;
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                if (b[u] == b[u - 1]) {
                    ++cnt;
                } else {
                    if (b[u - 1] != ~0u) atomicAdd(&h[b[u - 1]], cnt);
                    cnt = 1u;
                }
            }
            if (b[3] != ~0u) atomicAdd(&h[b[3]], cnt);
        }
    }
    __syncthreads();
    unsigned *out = a.hist + ((int64_t)q * a.groups + w) * kNC;
    for (int c = tid; c < kNC; c += kBA) out[c] = h[c];
}

__global__ __launch_bounds__(256) void sl_lut_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned *gcnt = reinterpret_cast<unsigned *>(lds_raw);   // [s1][W] records of group w in bin b
    __shared__ unsigned short lut[kNC + 1];
    __shared__ unsigned short clo[kS1Max + 1];
    __shared__ unsigned wave_tot[4], wave_max[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = blockIdx.x;
    constexpr int kPer = kNC / 256;   // consecutive coarse buckets per thread
    unsigned h[kPer], sum = 0u, mx = 0u;
#pragma unroll
    for (int x = 0; x < kPer; ++x) {
        unsigned v = 0u;
        for (int w = 0; w < a.groups; ++w) v += a.hist[((int64_t)q * a.groups + w) * kNC + tid * kPer + x];
        h[x] = v;
        sum += v;
        mx = v > mx ? v : mx;
    }
    const unsigned incl = wave_scan_add(sum);
    mx = wave_max_u32(mx);
    if (lane == 63) wave_tot[wave] = incl;
    if (lane == 0) wave_max[wave] = mx;
    for (int b = tid; b <= a.s1; b += 256) clo[b] = (unsigned short)kNC;
    __syncthreads();
    unsigned run = incl - sum, h_max = 0u;
    for (int x = 0; x < 4; ++x) {
        if (x < wave) run += wave_tot[x];
        h_max = wave_max[x] > h_max ? wave_max[x] : h_max;
    }
    // Bins of consecutive coarse buckets, whatever the phase distribution: bucket c goes to bin
    // (samples before c) / fill, so a bin holds fewer than fill + (one coarse bucket) <= kCap - kSlack samples
    // and there are at most n / fill + 1 of them.
    const int fill = kCap - kSlack - (int)h_max;
    const bool general = fill < kMinFill;
    const unsigned div = general ? (unsigned)kMinFill : (unsigned)fill;
#pragma unroll
    for (int x = 0; x < kPer; ++x) {
        const unsigned b = run / div;
        lut[tid * kPer + x + 1] = (unsigned short)(b < (unsigned)(a.s1 - 1) ? b : (unsigned)(a.s1 - 1));
        run += h[x];
    }
    if (tid == 0) {
        lut[0] = 0;
        a.flag[q] = general ? 1u : 0u;
    }
    __syncthreads();
    // first coarse bucket of every bin (a heavy bucket may skip a bin: it stays empty, c_lo == c_hi)
#pragma unroll
    for (int x = 0; x < kPer; ++x) {
        const int c = tid * kPer + x;
        const int prev = c == 0 ? -1 : (int)lut[c], cur = (int)lut[c + 1];   // lut[c + 1] = bin of bucket c
        for (int b = prev + 1; b <= cur; ++b) clo[b] = (unsigned short)c;
    }
    __syncthreads();
    for (int c = tid; c < kNC; c += 256) a.lut[(int64_t)q * kNC + c] = lut[c + 1];
    for (int b = tid; b <= a.s1; b += 256) a.clo[(int64_t)q * (a.s1 + 1) + b] = clo[b];
    if (tid == 0) {
        const unsigned nb = general ? 0u : (unsigned)lut[kNC] + 1u;
        a.nbins[q] = nb;
        // Slices mode: with t non-decreasing the samples of one cycle of the period are consecutive AND in phase
        // order, so the samples of (cycle c, bin b) are one slice of t[] / m[] and the partition kernel's lists are
        // not needed - a table of the cells' first samples is (sl_bound_kernel).  Taken when the cells hold >=
        // kMinSlice samples on average and the table row holds the cycles.
        const double period = a.periods[a.p0 + q];
        const bool safe = period_is_safe(period, a.bad_t[0] == 0u);
        const double y = 1.0 / period;
        const double q0 = exact_quotient(a.t[0], period, y, safe), q1 = exact_quotient(a.t[a.n - 1], period, y, safe);
        const double c0 = __builtin_floor(q0), c1 = __builtin_floor(q1);
        const double cycles = c1 - c0 + 1.0;
        // (a negative period runs the phases DOWN along t: neither mode applies)
        const bool slices = a.slices != 0 && a.bad_t[1] == 0u && period > 0.0 && nb > 0u && cycles >= 1.0 && cycles < (double)kCycS &&
                            cycles * (double)nb * (double)kMinSlice <= (double)a.n;
        a.ncyc[q] = slices ? (int)cycles : 0;
        a.cyc0[q] = c0;
        // One cycle (the period outlasts the samples) and t non-decreasing: the phases are in order as the samples
        // stand - no sort, one pass (sl_direct_kernel).  These are also the periods whose phases pile up in a few
        // coarse buckets (p >> baseline), which no bin table can take.
        // (... or less than one cycle across a cycle boundary: the later samples' phases lie below the first sample's)
        if (a.direct != 0 && a.bad_t[1] == 0u && period > 0.0 && (cycles == 1.0 || (cycles == 2.0 && q1 - c1 < q0 - c0))) {
            a.flag[q] = kFlagDirect;
            a.nbins[q] = 0u;
            a.ncyc[q] = 0;
        } else if (a.no_lists != 0 && !slices && nb > 0u) {
            a.flag[q] = 32u;             // (cannot happen: the host's test is the stricter one)
            a.nbins[q] = 0u;
        }
    }
    // how many records every group contributes to every bin - from the groups' own histograms, so the partition
    // kernel's runs are packed exactly
    const int W = a.groups;
    for (int x = tid; x < a.s1 * W; x += 256) gcnt[x] = 0u;
    __syncthreads();
    for (int w = 0; w < W; ++w) {
        // (a thread's 16 consecutive buckets lie in one or two bins: one atomic per bin, not per bucket)
        unsigned acc = 0u;
        int bin = (int)lut[tid * kPer + 1];
#pragma unroll
        for (int x = 0; x < kPer; ++x) {
            const int c = tid * kPer + x;
            const int bc = (int)lut[c + 1];
            if (bc != bin) {
                if (acc) atomicAdd(&gcnt[bin * W + w], acc);
                acc = 0u;
                bin = bc;
            }
            acc += a.hist[((int64_t)q * W + w) * kNC + c];
        }
        if (acc) atomicAdd(&gcnt[bin * W + w], acc);
    }
    __syncthreads();
    for (int b = tid; b < a.s1; b += 256) {
        unsigned at = 0u;
        for (int w = 0; w < W; ++w) {
            a.boff[((int64_t)q * a.s1 + b) * W + w] = at;
            at += gcnt[b * W + w];
        }
        a.bcnt[(int64_t)q * a.s1 + b] = at;
    }
    // sorted position of every bin's first record (the Supersmoother path lays the sorted curve out in one piece)
    __syncthreads();
    {
        constexpr int kBins = kS1Max / 256;   // consecutive bins per thread
        unsigned c[kBins], tot = 0u;
#pragma unroll
        for (int x = 0; x < kBins; ++x) {
            const int b = tid * kBins + x;
            c[x] = 0u;
            if (b < a.s1)
                for (int w = 0; w < W; ++w) c[x] += gcnt[b * W + w];
            tot += c[x];
        }
        const unsigned inc = wave_scan_add(tot);
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        unsigned at = inc - tot;
        for (int x = 0; x < wave; ++x) at += wave_tot[x];
#pragma unroll
        for (int x = 0; x < kBins; ++x) {
            const int b = tid * kBins + x;
            if (b < a.s1) a.bstart[(int64_t)q * a.s1 + b] = at;
            at += c[x];
        }
    }
}

template <int S1P>
__global__ __launch_bounds__(kBA) void sl_part_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *st_phi = reinterpret_cast<double *>(lds_raw);          // [kTA] the tile's records grouped by bin
    double *st_m = st_phi + kTA;
    unsigned *tcnt = reinterpret_cast<unsigned *>(st_m + kTA);      // [2][S1P] records of a tile per bin (tile parity)
    unsigned *start = tcnt + 2 * S1P;                                // [S1P] first staged position of every bin
    unsigned *run = start + S1P;                                     // [S1P] next free place in the bin's list
    unsigned short *st_l = reinterpret_cast<unsigned short *>(run + S1P);   // [kTA] position in the tile (-> sample index)
    unsigned short *lut = st_l + kTA;                                // [kNC]
    __shared__ unsigned wave_tot[kBA / 64];
    __shared__ int s_over;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int q, w;
    if (!period_and_group(a, q, w)) return;               // (workgroup-uniform)
    if (a.flag[q] != 0u || a.ncyc[q] != 0) return;        // (workgroup-uniform) the general kernel, or sl_bound_kernel, takes this period
    const int s1 = a.s1;
    const double period = a.periods[a.p0 + q];
    const double y = 1.0 / period;
    const bool safe = period_is_safe(period, a.bad_t[0] == 0u);
    for (int c = tid; c < kNC; c += kBA) lut[c] = a.lut[(int64_t)q * kNC + c];
    for (int b = tid; b < S1P; b += kBA) {
        run[b] = b < s1 ? a.boff[((int64_t)q * s1 + b) * a.groups + w] : 0u;   // this group's place in the bin's list
        tcnt[b] = 0u;
        tcnt[S1P + b] = 0u;
    }
    if (tid == 0) s_over = 0;
    const int64_t list0 = ((int64_t)q * s1) * kCap;   // + b * kCap: the list of bin b
    constexpr int kPer = S1P / kBA;
    // the next tile's samples are requested before this tile is worked off
    double tn[4], mn[4];
    auto request = [&](int64_t kappa) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = sample_of(kappa, u * kBA + tid, w, a.tiles_w);
            const int64_t ic = i < a.n ? i : a.n - 1;
            tn[u] = a.t[ic];
            mn[u] = a.m[ic];
        }
    };
    request(0);
    __syncthreads();
    for (int64_t kappa = 0; kappa < a.tiles_w; ++kappa) {
        unsigned *tc = tcnt + (kappa & 1) * S1P;
        double tv[4], mv[4], phi[4];
        bool live[4];
        int bin[4];
        unsigned arr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            tv[u] = tn[u];
            mv[u] = mn[u];
            live[u] = sample_of(kappa, u * kBA + tid, w, a.tiles_w) < a.n;
        }
        if (kappa + 1 < a.tiles_w) request(kappa + 1);
        phases4(tv, period, y, safe, phi);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bin[u] = lut[coarse_of<kNC>(phi[u])];
            arr[u] = 0u;
            if (live[u]) arr[u] = atomicAdd(&tc[bin[u]], 1u);
        }
        __syncthreads();
        // exclusive scan of the tile's counts -> start[]
        unsigned c[kPer], sum = 0u;
#pragma unroll
        for (int x = 0; x < kPer; ++x) {
            c[x] = tc[tid * kPer + x];
            sum += c[x];
        }
        const unsigned incl = wave_scan_add(sum);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned pos = incl - sum, tile_n = 0u;
#pragma unroll
        for (int x = 0; x < kBA / 64; ++x) {
            if (x < wave) pos += wave_tot[x];
            tile_n += wave_tot[x];
        }
#pragma unroll
        for (int x = 0; x < kPer; ++x) {
            start[tid * kPer + x] = pos;
            pos += c[x];
        }
        // (the other parity's counts were last read - by this thread - in the previous tile: zero them for the next)
#pragma unroll
        for (int x = 0; x < kPer; ++x) tcnt[((kappa + 1) & 1) * S1P + tid * kPer + x] = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (live[u]) {
                const unsigned l = start[bin[u]] + arr[u];
                st_phi[l] = phi[u];
                st_m[l] = mv[u];
                st_l[l] = (unsigned short)(u * kBA + tid);
            }
        }
        __syncthreads();
        bool over = false;
        for (unsigned j = tid; j < tile_n; j += kBA) {
            const double ph = st_phi[j];
            const int b = lut[coarse_of<kNC>(ph)];
            const unsigned dst = run[b] + (j - start[b]);
            if (dst < (unsigned)kCap) {   // (always: the offsets are exact; a guard against writing outside the list)
                const int64_t at = list0 + (int64_t)b * kCap + dst;
                rec_t r;
                r.x = ph;
                r.y = st_m[j];
                a.pm[at] = r;
                a.ix[at] = (unsigned)sample_of(kappa, (int)st_l[j], w, a.tiles_w);
            } else {
                over = true;
            }
        }
        if (over) s_over = 1;
        __syncthreads();
        // the tile's runs are appended: advance the places (own elements only)
#pragma unroll
        for (int x = 0; x < kPer; ++x) run[tid * kPer + x] += c[x];
    }
    __syncthreads();
    if (tid == 0 && s_over) atomicOr(&a.flag[q], 2u);
}

// Slices mode, in place of the partition kernel: the table of the cells' first samples.  With t non-decreasing
// key(i) = (cycle of sample i) * bins + (bin of its phase) never falls along i, so every cell (cycle, bin) is one run
// of consecutive samples and the table is written where the key steps up (a step over several keys = empty cells,
// all starting at the same sample).  Same groups of tiles as the histogram kernel; nothing but t is read (8 bytes a
// sample, from the caches) and one word per cell written - the partition kernel moves 36 bytes a sample.
// (Round 6 experiment, measured and withdrawn - profiles/r06_sl_bound_wave_runs_experiment.patch: every WAVE walking its
// own contiguous run of the group's samples, the neighbour's key by DPP and the carry in an SGPR, no LDS exchange and
// no barrier per tile, the next step's stamps requested ahead: 364 against 331 us per 256 periods at N = 1e6.  The
// kernel is bound by its ~75 VALU instructions per sample - the exact quotient, floor, the 64-bit index arithmetic -,
// not by the two barriers per tile.)
__global__ __launch_bounds__(kBA) void sl_bound_kernel(StreamArgs a) {
    __shared__ unsigned short lut[kNC];
    __shared__ int last[kBA];
    const int tid = threadIdx.x;
    int q, w;
    if (!period_and_group(a, q, w)) return;               // (workgroup-uniform)
    const int K = a.ncyc[q];
    if (a.flag[q] != 0u || K == 0) return;                // (workgroup-uniform)
    const int nb = (int)a.nbins[q];
    const double cyc0 = a.cyc0[q];
    const double period = a.periods[a.p0 + q];
    const double y = 1.0 / period;
    const bool safe = period_is_safe(period, a.bad_t[0] == 0u);
    for (int c = tid; c < kNC; c += kBA) lut[c] = a.lut[(int64_t)q * kNC + c];
    unsigned *tb = a.bnd + (int64_t)q * a.s1 * kCycS;
    const int key_end = K * nb;                            // (cycle K, bin 0): where the last cell ends
    auto key_of = [&](double tv) -> int {
        const double qq = exact_quotient(tv, period, y, safe);
        const double fl = __builtin_floor(qq);
        const int c = (int)(fl - cyc0);
        return c * nb + (int)lut[coarse_of<kNC>(qq - fl)];
    };
    auto put = [&](int from, int to, int64_t i) {          // cells (from, to] start at sample i
        for (int kk = from + 1; kk <= to; ++kk) {
            const int c = kk / nb, b = kk - c * nb;
            tb[b * kCycS + c] = (unsigned)i;
        }
    };
    const int64_t g0 = (int64_t)w * a.tiles_w * kTA;
    if (g0 >= a.n) return;
    __syncthreads();                                       // (the table read by key_of is complete)
    int carry = g0 == 0 ? -1 : 0;                          // key of the sample before this thread's first
    if (g0 > 0) carry = key_of(a.t[g0 - 1]);
    for (int64_t kappa = 0; kappa < a.tiles_w; ++kappa) {
        const int64_t i0 = g0 + kappa * kTA + (int64_t)tid * 4;   // four consecutive samples per thread
        if (g0 + kappa * kTA >= a.n) break;                // (workgroup-uniform)
        int key[4];
        if (i0 + 4 <= a.n) {   // two 16-byte loads (8-byte aligned: t is the caller's pointer)
            typedef double pair_t __attribute__((ext_vector_type(2), aligned(8)));
            const pair_t t01 = *reinterpret_cast<const pair_t *>(a.t + i0), t23 = *reinterpret_cast<const pair_t *>(a.t + i0 + 2);
            key[0] = key_of(t01.x);
            key[1] = key_of(t01.y);
            key[2] = key_of(t23.x);
            key[3] = key_of(t23.y);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) key[u] = i0 + u < a.n ? key_of(a.t[i0 + u]) : key_end;
        }
        last[tid] = key[3];
        __syncthreads();
        int prev = tid > 0 ? last[tid - 1] : carry;
        carry = last[kBA - 1];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // (a sample past the end carries the key behind the last cell: the step up to it closes the table at n)
            if (key[u] > prev && i0 + u <= a.n) put(prev, key[u], i0 + u < a.n ? i0 + u : a.n);
            prev = key[u] > prev ? key[u] : prev;
        }
    }
    // (n a multiple of the tile: no sample past the end stands in - the last sample's thread closes the table)
    if ((a.n % kTA) == 0 && g0 + (int64_t)a.tiles_w * kTA >= a.n && g0 < a.n) {
        const int64_t il = a.n - 1 - g0;                   // the last sample, counted inside this group
        if ((il % kTA) / 4 == tid) put(key_of(a.t[a.n - 1]), key_end, a.n);
    }
}

// Persistent: a workgroup per CU walks the (period, bin) items.  Built and measured against this kernel (N = 1e6 x
// 2048 periods, 802 us per batch of 96 periods before the metadata table below, 745 us with it):
//  - reading the first four members of every record's fine bucket side by side instead of the data-dependent loop:
//    no gain;
//  - nothing rewritten after the scatter - every record finds its rank AND its sorted predecessor (largest smaller
//    member of its bucket, else the largest member of the nearest non-empty bucket below) and adds its own segment,
//    four barriers instead of eight: 1010 us (six records per thread each walk three or four dependent LDS
//    round trips one after the other), and the sum depends on the arrival order of the partition's atomics;
//  - alternating counter halves + the bin's sum written one barrier later (two barriers fewer): no change - kept;
//  - the workgroup only groups the bin by 2048 mid buckets and every WAVE ranks and sums ranges of <= 255 records by
//    itself (the several-slice kernels' sl_ranges.inc on records already in LDS, no barrier in the data-dependent
//    part): 1055 us, 921 us with a parallel link step and a range table built by the counters' owners - the byte
//    counters' scan per range and rows filled to two thirds cost ~900 instructions per 170 records, and without a
//    second register set for the next bin's records every wave waits 6.6 k cycles for them.
// The cycle stamps taken for the last variant showed what is fixed here: the next bins' counts and coarse ranges
// were SCALAR loads, and a scalar load comes back with the first LDS wait behind it (one counter for both): 6 - 8 k
// cycles of every bin's ~27 k.
__global__ __launch_bounds__(kBB) void sl_sort_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned long long *key_t = reinterpret_cast<unsigned long long *>(lds_raw);   // [kCap] by fine bucket, then sorted
    double *m_s = reinterpret_cast<double *>(key_t + kCap);                          // [kCap] m in sorted order
    unsigned *i_t = reinterpret_cast<unsigned *>(m_s + kCap);                        // [kCap] sample index by fine bucket
    unsigned *fcnt = i_t + kCap;                                                     // [kFineB / 2] packed 16-bit
    __shared__ unsigned wave_tot[kBB / 64];
    __shared__ double red[2][kBB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s1 = a.s1;
    // The items of the batch = the bins every period's table actually uses (about half of the s1 reserved per
    // period), numbered consecutively through a prefix over the periods kept in LDS: the walk touches no empty
    // bin and reads nothing from global memory to find its next item.  (The first version walked all batch x s1
    // slots and read two dependent words per slot to find out whether it was empty: 2.4 us per slot, a third of
    // the kernel's time.)
    __shared__ unsigned pre[kBatchMax + 1];
    {
        unsigned v = 0u;
        if (tid < a.batch) v = a.flag[tid] != 0u ? 0u : a.nbins[tid];
        const unsigned inc = wave_scan_add(v);
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        unsigned at = inc - v;
        for (int x = 0; x < wave; ++x) at += wave_tot[x];
        if (tid < a.batch) pre[tid] = at;
        if (tid == a.batch - 1) pre[a.batch] = at + v;
        __syncthreads();
    }
    const int64_t n_items = (int64_t)pre[a.batch];
    // This workgroup's k-th item is live item blockIdx + k * gridDim.  Slot, record count and coarse range of 64 items
    // at a time are looked up by 64 threads (a binary search over the prefix, three global loads) and kept in LDS:
    // the walk itself then reads nothing from global memory.  (Fetched per item they were scalar loads, and a scalar
    // load comes back with the first LDS wait behind it - they share one counter -: 8 k cycles per bin, measured.)
    constexpr int kRing = 128;         // two halves of 64: the walk looks two items ahead
    __shared__ int meta_slot[kRing], meta_k[kRing];   // (meta_k: cycles, negative when the item is its period's last bin)
    __shared__ unsigned meta_n[kRing], meta_c[kRing];
    __shared__ double meta_per[kRing];
    auto refill = [&](int trip0) {       // items of trips [trip0, trip0 + 64)
        __syncthreads();
        if (tid < 64) {
            const int64_t kk = (int64_t)blockIdx.x + (int64_t)(trip0 + tid) * (int64_t)gridDim.x;
            int slot = -1, cyc = 0;
            unsigned n = 0u, c = 0x10000u;
            double per = 1.0;
            if (kk < n_items) {
                int lo = 0, hi = a.batch;     // largest q with pre[q] <= kk
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if ((int64_t)pre[mid] <= kk) lo = mid; else hi = mid;
                }
                slot = lo * s1 + (int)(kk - (int64_t)pre[lo]);
                const int64_t at = (int64_t)lo * (s1 + 1) + (kk - (int64_t)pre[lo]);
                n = a.bcnt[slot];
                c = (unsigned)a.clo[at] | ((unsigned)a.clo[at + 1] << 16);
                cyc = a.ncyc[lo];
                if (kk + 1 == (int64_t)pre[lo + 1]) cyc = -cyc;   // the period's last bin: its slices end where the next cycle's first bin starts
                per = a.periods[a.p0 + lo];
            }
            const int at = (trip0 + tid) & (kRing - 1);
            meta_slot[at] = slot;
            meta_n[at] = n;
            meta_c[at] = c;
            meta_k[at] = cyc;
            meta_per[at] = per;
        }
        __syncthreads();
    };
    // The NEXT item's records are requested (a second register set) before the current one is worked off.  Two
    // things make that prefetch real: vector loads return in order, so whatever else the current item reads from
    // global memory (its bin's coarse range, the count two items ahead) is requested BEFORE it - a younger load's
    // wait would drain the prefetch -, and a scheduling barrier pins the requests where they are written (their
    // results are first used one trip later; the compiler otherwise sinks them down to that use).  It needs a kernel
    // without register spills: a scratch reload is a vector load too.  (Measured: the lists come from the Infinity
    // Cache the partition kernel just wrote them through, so the prefetch itself buys little - 6 us of 445.)
    rec_t rn[kPerB];
    unsigned idn[kPerB];
    // (a wave takes kPerB CONSECUTIVE rows of 64 records: record rb + 64 e of the list, rb = the wave's first + lane -
    // in slices mode the slice of row e + 1 is then a step or two behind that of row e, no second search)
    auto request = [&](const int tt, int64_t slot, int n) {
        if (slot >= 0 && n > 0 && n <= kCap) {   // (workgroup-uniform)
            const int w0 = __builtin_amdgcn_readfirstlane((tt >> 6) * (kPerB * 64));
#pragma unroll
            for (int e = 0; e < kPerB; ++e) {
                // (whole rows past the bin's count are skipped; the last row reads on inside the list's kCap places -
                // whatever lies there is never used)
                if (w0 + e * 64 < n) {   // (wave-uniform)
                    rn[e] = a.pm[slot * kCap + w0 + e * 64 + (tt & 63)];
                    idn[e] = a.ix[slot * kCap + w0 + e * 64 + (tt & 63)];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // Slices mode (the period's ncyc > 0): the bin's records are K slices of t[] / m[], one per cycle - [vs, ve) of the
    // table rows fetched a trip ahead by the first K threads.  An exclusive prefix over the slices' lengths goes to
    // LDS, every thread finds the slice of each of its records by a binary search over it (K is the same for all:
    // a uniform loop) and requests t and m there; the phase is computed when the records are taken up.
    __shared__ unsigned short sl_pre[kCycS];
    __shared__ unsigned sl_start[kCycS];
    __shared__ unsigned sl_wtot[kCycS / 64];
    unsigned vs = 0u, ve = 0u;
    auto rows_request = [&](const int tt, int64_t slot, int Ks) {   // the table rows of an item two trips ahead
        const int K = Ks < 0 ? -Ks : Ks;
        if (slot >= 0 && K > 0 && tt < K) {
            const int64_t q = slot / s1;
            const unsigned *row = a.bnd + slot * kCycS;
            vs = row[tt];
            ve = Ks < 0 ? a.bnd[q * s1 * kCycS + tt + 1] : row[kCycS + tt];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto request_slices = [&](const int tt, int n, int K) {             // (workgroup-uniform; two barriers)
        const unsigned len = tt < K ? ve - vs : 0u;
        const unsigned incl = wave_scan_add(len);
        if (lane == 63 && wave < kCycS / 64) sl_wtot[wave] = incl;
        __syncthreads();
        if (tt < K) {
            unsigned ex = incl - len;
            for (int x = 0; x < wave; ++x) ex += sl_wtot[x];
            sl_pre[tt] = (unsigned short)ex;
            sl_start[tt] = vs;
        }
        __syncthreads();
        int top = 1;
        while (top * 2 <= K - 1) top *= 2;                // largest power of two <= K - 1 (no step for K = 1)
        const int w0 = __builtin_amdgcn_readfirstlane((tt >> 6) * (kPerB * 64));
        const unsigned last = (unsigned)(n > 0 ? n - 1 : 0);
        unsigned sidx = 0u;                                // largest s with sl_pre[s] <= r: searched for the first row ...
        {
            const unsigned r = (unsigned)(w0 + (tt & 63)) < last ? (unsigned)(w0 + (tt & 63)) : last;
            for (int step = K > 1 ? top : 0; step > 0; step >>= 1) {
                const unsigned probe = sidx + (unsigned)step;
                const unsigned pv = sl_pre[probe < (unsigned)K ? probe : 0u];
                if (probe < (unsigned)K && pv <= r) sidx = probe;
            }
        }
#pragma unroll
        for (int e = 0; e < kPerB; ++e) {
            if (w0 + e * 64 < n) {   // (wave-uniform)
                const unsigned rr = (unsigned)(w0 + e * 64 + (tt & 63));
                const unsigned r = rr < last ? rr : last;
                // ... and walked up from the row before (64 records further on: a slice or two)
                while (sidx + 1u < (unsigned)K && (unsigned)sl_pre[sidx + 1u] <= r) ++sidx;
                unsigned at = sl_start[sidx] + (r - (unsigned)sl_pre[sidx]);
                at = at < (unsigned)a.n ? at : (unsigned)(a.n - 1);   // (always inside: the table and the count come from the same phases)
                rn[e] = a.tm[at];
                idn[e] = at;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const bool t_safe = a.bad_t[0] == 0u;
    refill(0);
    refill(64);
    int trip = 0;
    int64_t sl0 = __builtin_amdgcn_readfirstlane(meta_slot[0]);
    int n0 = __builtin_amdgcn_readfirstlane((int)meta_n[0]);
    unsigned c0 = (unsigned)__builtin_amdgcn_readfirstlane((int)meta_c[0]);
    int k0 = __builtin_amdgcn_readfirstlane(meta_k[0]);   // (|k0| cycles; 0: lists mode)
    if (k0 != 0) {
        rows_request(tid, sl0, k0);
        request_slices(tid, sl0 >= 0 ? n0 : 0, k0 < 0 ? -k0 : k0);
    } else {
        request(tid, sl0, n0);
    }
    rows_request(tid, __builtin_amdgcn_readfirstlane(meta_slot[1]), __builtin_amdgcn_readfirstlane(meta_k[1]));
    int par = 0;                  // which half of red[] this bin's waves sum into
    int64_t pend = -1;            // the bin whose waves' sums wait in red[pend_par] (written out one barrier later)
    int pend_par = 0;
    auto flush = [&]() {
        if (tid == 0 && pend >= 0) {
            double sum = 0.0;
            for (int x = 0; x < kBB / 64; ++x) sum += red[pend_par][x];
            a.slen[pend] = sum;
        }
        pend = -1;
    };
    for (; sl0 >= 0; ++trip) {
        // (the thread id goes through an opaque copy once per trip: the LDS addresses built from it are loop-invariant,
        // and hoisted out of the loop they are spilled - a scratch reload is a vector load, and waiting for it
        // drains the prefetch)
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const int rb = (tl >> 6) * (kPerB * 64) + (tl & 63);   // this thread's record of row e: rb + 64 e
        const int64_t item = sl0;
        const int n_s = n0;
        const double c_lo = (double)(c0 & 0xFFFFu), c_hi = (double)(c0 >> 16);
        rec_t rec[kPerB];
        unsigned id[kPerB];
#pragma unroll
        for (int e = 0; e < kPerB; ++e) {
            rec[e] = rn[e];
            id[e] = idn[e];
        }
        if (k0 != 0) {   // (workgroup-uniform) slices mode: what came is t, not the phase
            const double per = meta_per[trip & (kRing - 1)];
            const double y = 1.0 / per;
            const bool safe = period_is_safe(per, t_safe);
#pragma unroll
            for (int e = 0; e < kPerB; ++e) rec[e].x = fast_phase(rec[e].x, per, y, safe);
        }
        // The NEXT item's records are requested once this item's have been scattered to LDS (its phases, keys and
        // indices are dead by then: two full register sets side by side - the request used to go out up here - do
        // not fit 128 registers together with the slices mode's state, and a scratch reload is a vector load whose
        // wait drains the prefetch).  Ranking, rewriting and the segments lie between the request and its use.
        auto advance = [&]() {
            if (((trip + 2) & 63) == 0) refill(trip + 2);     // (workgroup-uniform)
            const int r1 = (trip + 1) & (kRing - 1), r2 = (trip + 2) & (kRing - 1);
            sl0 = __builtin_amdgcn_readfirstlane(meta_slot[r1]);
            n0 = __builtin_amdgcn_readfirstlane((int)meta_n[r1]);
            c0 = (unsigned)__builtin_amdgcn_readfirstlane((int)meta_c[r1]);
            k0 = __builtin_amdgcn_readfirstlane(meta_k[r1]);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 != 0) request_slices(tl, sl0 >= 0 ? n0 : 0, k0 < 0 ? -k0 : k0);
            else request(tl, sl0, n0);
            rows_request(tl, __builtin_amdgcn_readfirstlane(meta_slot[r2]), __builtin_amdgcn_readfirstlane(meta_k[r2]));
        };
        advance();
        if (n_s <= 0) continue;                           // (workgroup-uniform)
        if (n_s > kCap) {                                 // (cannot happen: the bin table keeps bins below kCap)
            if (tl == 0) atomicOr(&a.flag[item / s1], 4u);
            continue;
        }
        const int q = (int)(item / s1);
        for (int x = tl; x < kFineB / 2; x += kBB) fcnt[x] = 0u;
        __syncthreads();
        const double fscale = (double)kFineB / (c_hi - c_lo > 0.0 ? c_hi - c_lo : 1.0);
        // fine bucket: a monotone refinement of the coarse bucket (phi * 4096 is exact, the scale positive);
        // arrival rank from the packed counters
        unsigned long long key[kPerB];
        unsigned fb[kPerB], arr[kPerB];
#pragma unroll
        for (int e = 0; e < kPerB; ++e) {
            const double phi = rec[e].x;
            key[e] = phase_key(phi);
            const double rel = (phi * (double)kNC - c_lo) * fscale;
            const unsigned f = (unsigned)rel;                                    // (saturating; negative -> 0)
            fb[e] = phi == phi ? (f < (unsigned)(kFineB - 1) ? f : (unsigned)(kFineB - 1)) : (unsigned)(kFineB - 1);
            arr[e] = 0u;
            if (rb + e * 64 < n_s) {
                const unsigned sh = (fb[e] & 1u) * 16u;
                arr[e] = (atomicAdd(&fcnt[fb[e] >> 1], 1u << sh) >> sh) & 0xFFFFu;
            }
        }
        __syncthreads();
        flush();            // every wave is past the bin before: its sums are complete
        {   // exclusive scan of the fine counters (8 per thread), written back as packed starts
            constexpr int kW = kFineB / 2 / kBB;   // words per thread
            unsigned c[2 * kW], sum = 0u;
#pragma unroll
            for (int x = 0; x < kW; ++x) {
                const unsigned v = fcnt[tl * kW + x];
                c[2 * x] = v & 0xFFFFu;
                c[2 * x + 1] = v >> 16;
                sum += c[2 * x] + c[2 * x + 1];
            }
            const unsigned incl = wave_scan_add(sum);
            if (lane == 63) wave_tot[wave] = incl;
            __syncthreads();
            unsigned run = incl - sum;
            for (int x = 0; x < wave; ++x) run += wave_tot[x];
#pragma unroll
            for (int x = 0; x < kW; ++x) {
                const unsigned lo = run;
                run += c[2 * x];
                const unsigned hi = run;
                run += c[2 * x + 1];
                fcnt[tl * kW + x] = lo | (hi << 16);
            }
        }
        __syncthreads();
        auto start_of = [&](unsigned f) -> unsigned {
            return f < (unsigned)kFineB ? (fcnt[f >> 1] >> ((f & 1u) * 16u)) & 0xFFFFu : (unsigned)n_s;
        };
        unsigned se[kPerB];   // the fine bucket's first slot | one past its last << 16
#pragma unroll
        for (int e = 0; e < kPerB; ++e) {
            const unsigned st = start_of(fb[e]);
            se[e] = st | (start_of(fb[e] + 1u) << 16);
            if (rb + e * 64 < n_s) {
                key_t[st + arr[e]] = key[e];
                i_t[st + arr[e]] = id[e];
            }
        }
        __syncthreads();
        // final position = start of the fine bucket + members that sort before (phase pattern, then sample index)
        unsigned fin[kPerB];
#pragma unroll
        for (int e = 0; e < kPerB; ++e) {
            unsigned before = 0u;
            const unsigned st = se[e] & 0xFFFFu, en = se[e] >> 16;
            if (rb + e * 64 < n_s) {
                for (unsigned o = st; o < en; ++o) {
                    const unsigned long long ko = key_t[o];
                    before += (ko < key[e] || (ko == key[e] && i_t[o] < id[e])) ? 1u : 0u;
                }
            }
            fin[e] = st + before;
        }
        __syncthreads();   // every thread is done reading the bucket-ordered keys: they become the sorted ones
#pragma unroll
        for (int e = 0; e < kPerB; ++e) {
            if (rb + e * 64 < n_s) {
                key_t[fin[e]] = key[e];
                m_s[fin[e]] = rec[e].y;
            }
        }
        __syncthreads();
        // segments in sorted order: position p against p - 1 (the lane below; lane 0 reads it)
        const int64_t sorted_at = a.sorted ? (int64_t)q * a.n + a.bstart[item] : 0;
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < kPerB; ++e) {
            const int p = tl + e * kBB;
            const bool live = p < n_s;
            const int pc = live ? p : 0;
            const double phi = __longlong_as_double((long long)key_t[pc]), mm = m_s[pc];
            double pphi = 0.0, pmm = 0.0;
            if (lane == 0 && live && p > 0) {
                pphi = __longlong_as_double((long long)key_t[p - 1]);
                pmm = m_s[p - 1];
            }
            pphi = lane_below(phi, pphi);
            pmm = lane_below(mm, pmm);
            if (live && p > 0) acc += short_hypot(mm - pmm, phi - pphi);
            if (a.sorted && live) {
                rec_t r;
                r.x = phi;
                r.y = mm;
                a.sorted[sorted_at + p] = r;
            }
            if (live && p == 0) {
                a.ssum[item * 4 + 0] = phi;
                a.ssum[item * 4 + 1] = mm;
            }
            if (live && p == n_s - 1) {
                a.ssum[item * 4 + 2] = phi;
                a.ssum[item * 4 + 3] = mm;
            }
        }
        acc = wave_sum_fixed(acc);
        if (lane == 0) red[par][wave] = acc;
        pend = item;
        pend_par = par;
        par ^= 1;
    }
    __syncthreads();
    flush();
}

// One cycle of the period covers all samples and t is non-decreasing: phase order = sample order (equal phases -
// equal times - stay in index order, as the stable sort leaves them).  kDirectW workgroups per such period sum the
// segments of one contiguous share of the samples each, every thread the samples i = its id mod 1024 of the share,
// the threads' sums in a fixed order; sl_link_kernel adds the shares in order and the closing segment.
__global__ __launch_bounds__(kBB) void sl_direct_kernel(StreamArgs a) {
    __shared__ double red[kBB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = (int)(blockIdx.x / kDirectW), g = (int)(blockIdx.x % kDirectW);
    if (a.flag[q] != kFlagDirect) return;                 // (workgroup-uniform)
    const double period = a.periods[a.p0 + q];
    const double y = 1.0 / period;
    const bool safe = period_is_safe(period, a.bad_t[0] == 0u);
    const int64_t share = (a.n + kDirectW - 1) / kDirectW;
    const int64_t lo = g * share > 1 ? g * share : 1, hi = (g + 1) * share < a.n ? (g + 1) * share : a.n;
    double acc = 0.0;
    for (int64_t i = lo + tid; i < hi; i += kBB) {        // segment (i - 1, i)
        const double p1 = fast_phase(a.t[i], period, y, safe), p0 = fast_phase(a.t[i - 1], period, y, safe);
        acc += short_hypot(a.m[i] - a.m[i - 1], p1 - p0);
    }
    acc = wave_sum_fixed(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double total = 0.0;
        for (int x = 0; x < kBB / 64; ++x) total += red[x];
        a.dpart[q * kDirectW + g] = total;
    }
}

// one wave per period of the batch: 64 bins' summaries are fetched side by side, then folded in bin order
__global__ __launch_bounds__(256) void sl_link_kernel(StreamArgs a) {
    const int lane = threadIdx.x & 63;
    const int q = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (q >= a.batch) return;
    const int64_t p = a.p0 + q;
    if (a.flag[q] == kFlagDirect) {                       // (sl_direct_kernel summed the segments share by share)
        if (lane == 0) {
            const double period = a.periods[p];
            const double y = 1.0 / period;
            const bool safe = period_is_safe(period, a.bad_t[0] == 0u);
            double total = 0.0;
            for (int g = 0; g < kDirectW; ++g) total += a.dpart[q * kDirectW + g];
            // closing segment of np.roll(-1): first minus last, no phase wrap (phase.py:50)
            total += hypot(a.m[0] - a.m[a.n - 1],
                           fast_phase(a.t[0], period, y, safe) - fast_phase(a.t[a.n - 1], period, y, safe));
            a.ell[p] = total;
            a.todo[p] = 0;
        }
        return;
    }
    if (a.flag[q] != 0u) {
        if (lane == 0) {
            a.todo[p] = (unsigned char)(a.flag[q] | 0x80u);   // (non-zero; the low bits say which kernel gave up - debug print)
            atomicAdd(a.todo_count, 1u);
        }
        return;
    }
    double total = 0.0, f_phi = 0.0, f_m = 0.0, l_phi = 0.0, l_m = 0.0;
    bool any = false;
    for (int s0 = 0; s0 < a.s1; s0 += 64) {
        const int s = s0 + lane;
        const int64_t item = (int64_t)q * a.s1 + (s < a.s1 ? s : a.s1 - 1);
        const unsigned cnt = s < a.s1 ? a.bcnt[item] : 0u;
        double len = 0.0, p0 = 0.0, m0 = 0.0, p1 = 0.0, m1 = 0.0;
        if (cnt) {
            len = a.slen[item];
            p0 = a.ssum[item * 4 + 0];
            m0 = a.ssum[item * 4 + 1];
            p1 = a.ssum[item * 4 + 2];
            m1 = a.ssum[item * 4 + 3];
        }
        unsigned long long live = __ballot(cnt != 0u);
        while (live) {                                   // (wave-uniform) non-empty bins in order
            const int l = __builtin_ctzll(live);
            live &= live - 1;
            const double bl = read_lane(len, l), bp0 = read_lane(p0, l), bm0 = read_lane(m0, l);
            total += bl;
            if (any) {
                total += hypot(bm0 - l_m, bp0 - l_phi);
            } else {
                f_phi = bp0;
                f_m = bm0;
                any = true;
            }
            l_phi = read_lane(p1, l);
            l_m = read_lane(m1, l);
        }
    }
    // closing segment of np.roll(-1): first minus last, no phase wrap (phase.py:50)
    if (any) total += hypot(f_m - l_m, f_phi - l_phi);
    if (lane == 0) {
        a.ell[p] = total;
        a.todo[p] = 0;
    }
}

}  // namespace stream

#include "timesort.inc"
#include "supersmoother.inc"

int64_t pad_pow2(int64_t n) {
    int64_t p = 2;
    while (p < n) p <<= 1;
    return p;
}

// (resident workgroups of the one-workgroup-per-period kernels - each owns 12 bytes a padded sample of scratch, 17 GB for
// 1024 of them at N = 1e6 -, fewer under a workspace budget: pdc_internal.h, WorkScale)
int64_t grid_for(int64_t n_periods) {
    int64_t cap = (int64_t)((double)kMaxGrid * work_scale());
    cap = cap < 1 ? 1 : cap;
    return n_periods < cap ? n_periods : cap;
}

// ranges: one per window of every slice, plus one per slice / oversized bucket
int64_t range_slots(int64_t n) { return (n / kWin + n / 8192 + 64 + 3) & ~(int64_t)3; }   // (multiple of 4: keeps every array 16-byte aligned)

// (the smaller of the two slice capacities: above it the kernel may need the per-period partition)
bool may_need_partition(int64_t n) { return n > Lds<unsigned>::capacity; }

int64_t scratch_bytes(int64_t n, int64_t n_periods, int64_t partition) {
    return grid_for(n_periods > 0 ? n_periods : 1) * (pad_pow2(n) * 12 + range_slots(n) * 44 + partition) + 512;
}

// AoS (t, m) records + flags of the fast path, placed behind the general scratch
int64_t fast_table_bytes(int64_t n) { return ((n * 16 + 255) & ~(int64_t)255) + 256; }

// ---- periods of ONE cycle, whatever kernels take the rest ---------------------------------------------------
// A period that outlasts the samples (p > baseline: the last ten of the reference's grid, phase.py:67-68 with dphi =
// 0.1) folds them into [0, baseline / p): with t non-decreasing that IS the sorted order, and the kernels that give
// a period to one workgroup are at their worst there - the phases pile up in a tenth of the coarse buckets, the
// ranges overflow and go through the deferred bitonic sort: 0.88 ms for such a period at N = 29 000 against 0.06 ms
// for any other, i.e. most of a call with the reference's default 1000 periods.  A pre-pass marks these periods
// (skip[]) and sums their segments as the samples stand; the kernels behind it pass them over.
namespace onecycle {
using namespace fast;
struct OneArgs {
    const double *t, *m, *periods;
    int64_t n, n_periods;
    const unsigned *bad;        // [0] some |t| outside {0} u [1e-150, 1e150]; [1] t not non-decreasing (sl_tame_kernel)
    unsigned char *skip;        // [n_periods] 1 = summed here
    unsigned *list, *count;     // the marked periods, in no particular order
    double *ell;
};

__global__ __launch_bounds__(256) void sl_onecycle_mark_kernel(OneArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n_periods) return;
    bool one = false;
    if (a.bad[1] == 0u && a.n >= 2) {
        const double period = a.periods[p];
        const double y = 1.0 / period;
        const bool safe = period_is_safe(period, a.bad[0] == 0u);
        const double q0 = exact_quotient(a.t[0], period, y, safe), q1 = exact_quotient(a.t[a.n - 1], period, y, safe);
        const double c0 = __builtin_floor(q0), c1 = __builtin_floor(q1);
        // (the samples span less than one cycle: all in one, or the later ones in the next with phases BELOW the first
        // sample's - sorted order = those, then the earlier ones, each as they stand: the same segments, the pair across
        // the cycle boundary being the closing one.  NaN: no)
        one = period > 0.0 && (c1 - c0 == 0.0 || (c1 - c0 == 1.0 && q1 - c1 < q0 - c0));
    }
    a.skip[p] = one ? 1 : 0;
    if (one) a.list[atomicAdd(a.count, 1u)] = (unsigned)p;
}

__global__ __launch_bounds__(kBlock) void sl_onecycle_kernel(OneArgs a) {
    __shared__ double red[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned count = *a.count;
    for (unsigned k = blockIdx.x; k < count; k += gridDim.x) {   // (workgroup-uniform)
        const int64_t p = a.list[k];
        const double period = a.periods[p];
        const double y = 1.0 / period;
        const bool safe = period_is_safe(period, a.bad[0] == 0u);
        double acc = 0.0;
        for (int64_t i = tid + 1; i < a.n; i += kBlock) {        // segment (i - 1, i)
            const double p1 = fast_phase(a.t[i], period, y, safe), p0 = fast_phase(a.t[i - 1], period, y, safe);
            acc += short_hypot(a.m[i] - a.m[i - 1], p1 - p0);
        }
        acc = wave_sum_fixed(acc);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            double total = 0.0;
            for (int x = 0; x < kBlock / 64; ++x) total += red[x];
            // closing segment of np.roll(-1): first minus last, no phase wrap (phase.py:50)
            total += hypot(a.m[0] - a.m[a.n - 1],
                           fast_phase(a.t[0], period, y, safe) - fast_phase(a.t[a.n - 1], period, y, safe));
            a.ell[p] = total;
        }
        __syncthreads();
    }
}
}  // namespace onecycle

// skip[n_periods], list[n_periods], the flags of sl_tame_kernel + the count
int64_t onecycle_bytes(int64_t n_periods) { return ((n_periods + 255) & ~(int64_t)255) + ((n_periods * 4 + 255) & ~(int64_t)255) + 512; }

// workspace of the two-workgroups-per-CU kernel, behind the (t, m) table: the marks for the one-workgroup
// kernel and the ticket counter
int64_t duo_bytes(int64_t n_periods) { return ((n_periods + 255) & ~(int64_t)255) + 256; }


// ---- streamed path: batch geometry and workspace ---------------------------------------------------------
// PDC_SL_STREAM_MIN: smallest N that takes the streamed kernels.  Default 262 144 = where the several-slice kernel's
// bit-plane indices end (same box, reference grid, ms streamed / ms several slices: N = 250 000 x 1000 3.0 / 3.1, x 4096
// 12.9 / 9.7; N = 262 000 x 2048 6.6 / 5.8, x 8192 29.0 / 20.9; N = 270 000 x 2048 6.8 / 7.4; N = 300 000 x 1000 3.6 / 5.0,
// x 4096 15.3 / 17.2; N = 383 000 x 2048 9.2 / 15.6).  (Before the one-cycle pre-pass - namespace onecycle - the
// kernels that give a period to one workgroup lost 0.9 ms and more per call to the ten longest periods of the
// reference's grid, and the streamed kernels looked better for few periods at any N: a rule by period count was
// measured, built and withdrawn the same day.)  PDC_SL_STREAM=0 switches the streamed kernels off (A/B, tests).
int64_t stream_min_n() {
    static const int64_t v = [] {
        const char *on = getenv("PDC_SL_STREAM");
        if (on && on[0] == '0') return (int64_t)1 << 62;
        const char *e = getenv("PDC_SL_STREAM_MIN");
        return e ? (int64_t)atoll(e) : (int64_t)262144;
    }();
    return v;
}
constexpr int64_t kStreamMaxN = (int64_t)(stream::kS1Max - 2) * stream::kMinFill;
bool stream_takes(int64_t n, int64_t) { return n >= stream_min_n() && n >= 4096 && n <= kStreamMaxN; }

struct StreamShape {
    int s1, batch, groups, tiles_w;
    int64_t o_bad, o_flag, o_nbins, o_ncyc, o_cyc0, o_hist, o_lut, o_clo, o_sub, o_bcnt, o_bstart, o_ssum, o_slen, o_dpart, o_bnd, o_ix, o_pm, o_tm, o_todo, o_tcount,
            o_ts, o_rhist, o_rtot, total;
    int ts_tiles;
};
StreamShape stream_shape(int64_t n, int64_t n_periods, bool lists = true) {
    auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    StreamShape h;
    h.s1 = (int)(n / stream::kMinFill + 2);
    h.s1 = h.s1 > stream::kS1Max ? stream::kS1Max : h.s1;
    const int64_t tiles = (n + stream::kTA - 1) / stream::kTA;
    // workgroups per period in the histogram / partition / boundary kernels: a power of two up to 4, each with >= ~16
    // tiles, and batches of up to 384 periods as far as 12 GB of lists allow (measured, N = 1e6 / 4e5 / 2.5e5 x 2048 /
    // 2048 / 4096 periods, slices mode: 8 groups x 96 periods 27.0 / 11.5 / 14.4 ms, 4 x 192 25.7 / 10.5 / 14.2, 2 x 384
    // 25.4 / 10.2 / 13.7, 4 x 384 25.2 / 10.1 / 13.7, 4 x 96 29.1 / 12.1 / 16.2: the kernels before the sort want >= 768
    // workgroups, the sort kernel few and long launches; lists mode, 16 / 8 / 4 groups: 37.3 / 35.3 / 35.2 ms at N = 1e6)
    int groups = 1;
    while (groups < 4 && tiles / (2 * groups) >= 16) groups *= 2;
    static const int env_groups = [] { const char *e = getenv("PDC_SL_STREAM_GROUPS"); return e ? atoi(e) : 0; }();
    if (env_groups == 1 || env_groups == 2 || env_groups == 4 || env_groups == 8 || env_groups == 16) groups = env_groups;
    while ((int64_t)h.s1 * groups * 4 > 128 * 1024 && groups > 1) groups /= 2;   // (the bin table kernel's LDS)
    h.groups = groups;
    h.tiles_w = (int)((tiles + groups - 1) / groups);
    static const int64_t env_batch = [] { const char *e = getenv("PDC_SL_STREAM_BATCH"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
    const int64_t list_bytes = (int64_t)h.s1 * stream::kCap * 20;                 // one period's lists
    const double scale = work_scale();                                            // (< 1 under a workspace budget)
    int64_t batch = (int64_t)((double)((int64_t)12 << 30) * scale) / list_bytes / 32 * 32;
    batch = batch > 384 ? 384 : (batch < 32 ? (scale < 1.0 ? (int64_t)((double)((int64_t)12 << 30) * scale) / list_bytes : 32) : batch);
    if (!lists) batch = (int64_t)(384.0 * scale);
    batch = batch < 1 ? 1 : batch;
    if (env_batch > 0) batch = env_batch;
    batch = batch > n_periods ? n_periods : batch;
    batch = batch < 1 ? 1 : (batch > stream::kBatchMax ? stream::kBatchMax : batch);   // (the sort kernel keeps a prefix over the batch's periods in LDS)
    h.batch = (int)batch;
    const int64_t items = batch * h.s1;
    h.o_bad = 0;
    h.o_flag = 256;
    h.o_nbins = h.o_flag + up(batch * 4);
    h.o_ncyc = h.o_nbins + up(batch * 4);
    h.o_cyc0 = h.o_ncyc + up(batch * 4);
    h.o_hist = h.o_cyc0 + up(batch * 8);
    h.o_lut = h.o_hist + up(batch * groups * stream::kNC * 4);
    h.o_clo = h.o_lut + up(batch * stream::kNC * 2);
    h.o_sub = h.o_clo + up(batch * (h.s1 + 1) * 2);
    h.o_bcnt = h.o_sub + up(items * groups * 4);
    h.o_bstart = h.o_bcnt + up(items * 4);
    h.o_ssum = h.o_bstart + up(items * 4);
    h.o_slen = h.o_ssum + up(items * 32);
    h.o_dpart = h.o_slen + up(items * 8);
    h.o_bnd = h.o_dpart + up(batch * stream::kDirectW * 8);
    h.o_ix = h.o_bnd + up(items * stream::kCycS * 4);
    h.o_pm = h.o_ix + (lists ? up(items * stream::kCap * 4) : 256);
    h.o_tm = h.o_pm + (lists ? up(items * stream::kCap * 16) : 256);
    h.o_todo = h.o_tm + up(n * 16);
    h.o_tcount = h.o_todo + up(n_periods);
    // the time sort of samples that come in any order (timesort.inc): (t, m) in order + the other side of the ping-pong,
    // the tiles' digit counts
    h.ts_tiles = (int)((n + tsort::kTile - 1) / tsort::kTile);
    h.o_ts = h.o_tcount + 256;
    h.o_rhist = h.o_ts + 4 * up(n * 8);
    h.o_rtot = h.o_rhist + up((int64_t)256 * h.ts_tiles * 4);
    h.total = h.o_rtot + 1024;
    return h;
}
int64_t stream_bytes(int64_t n, int64_t n_periods, bool lists = true) {
    return stream_takes(n, n_periods) && n_periods > 0 ? stream_shape(n, n_periods, lists).total : 0;
}

template <int KMAX, int BLK = duo::kB, int NBL = fast::kNB>
int launch_duo(const duo::DuoArgs &a, int64_t grid, hipStream_t st) {
    const size_t fixed = (size_t)(BLK / 64) * duo::kWaveB + 2 * (duo::kRangesD + 8) * 2 + 64;
    const size_t lds = fixed + (size_t)((a.n + 64 + 7) & ~7) * 2;
    PDC_TRY(allow_dynamic_lds((const void *)duo::sl_duo_kernel<KMAX, BLK, NBL>, duo::kLdsWg - duo::kStaticD));
    hipLaunchKernelGGL((duo::sl_duo_kernel<KMAX, BLK, NBL>), dim3((unsigned)grid), dim3(BLK), lds, st, a);
    return PDC_OK;
}

int cu_count(int device) {
    static int cached[64] = {};
    if (device < 0 || device >= 64) return 256;
    if (cached[device] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = 256;
        cached[device] = v;
    }
    return cached[device];
}


stream::StreamArgs stream_args(const StreamShape &h, char *area, const double *d_t, const double *d_m,
                               const double *d_periods, int64_t n, double *d_ell) {
    stream::StreamArgs sa;
    sa.t = d_t;
    sa.m = d_m;
    sa.periods = d_periods;
    sa.n = n;
    sa.s1 = h.s1;
    sa.groups = h.groups;
    sa.tiles_w = h.tiles_w;
    sa.bad_t = reinterpret_cast<unsigned *>(area + h.o_bad);
    sa.flag = reinterpret_cast<unsigned *>(area + h.o_flag);
    sa.hist = reinterpret_cast<unsigned *>(area + h.o_hist);
    sa.lut = reinterpret_cast<unsigned short *>(area + h.o_lut);
    sa.clo = reinterpret_cast<unsigned short *>(area + h.o_clo);
    sa.boff = reinterpret_cast<unsigned *>(area + h.o_sub);
    sa.bcnt = reinterpret_cast<unsigned *>(area + h.o_bcnt);
    sa.bstart = reinterpret_cast<unsigned *>(area + h.o_bstart);
    sa.nbins = reinterpret_cast<unsigned *>(area + h.o_nbins);
    sa.ncyc = reinterpret_cast<int *>(area + h.o_ncyc);
    sa.cyc0 = reinterpret_cast<double *>(area + h.o_cyc0);
    sa.bnd = reinterpret_cast<unsigned *>(area + h.o_bnd);
    static const bool slices = [] { const char *e = getenv("PDC_SL_SLICES"); return !(e && e[0] == '0'); }();
    sa.slices = slices ? 1 : 0;
    sa.direct = sa.slices;
    sa.no_lists = 0;
    sa.sorted = nullptr;
    sa.ssum = reinterpret_cast<double *>(area + h.o_ssum);
    sa.slen = reinterpret_cast<double *>(area + h.o_slen);
    sa.dpart = reinterpret_cast<double *>(area + h.o_dpart);
    sa.ix = reinterpret_cast<unsigned *>(area + h.o_ix);
    sa.pm = reinterpret_cast<fast::rec_t *>(area + h.o_pm);
    sa.todo = reinterpret_cast<unsigned char *>(area + h.o_todo);
    sa.tm = reinterpret_cast<const fast::rec_t *>(area + h.o_tm);
    sa.todo_count = reinterpret_cast<unsigned *>(area + h.o_tcount);
    sa.ell = d_ell;
    return sa;
}

// What the host knows about a call (the entry points that hold the host arrays look; the _dev entries do not know):
constexpr int kHintLists = 1;     // the workspace holds the streamed kernels' bin lists
constexpr int kHintOrdered = 2;   // t is non-decreasing: no time sort to launch

// PDC_SL_TIMESORT=0: samples in any order take the lists mode as up to round 4 (A/B, tests)
bool timesort_on() {
    static const bool on = [] {
        const char *e = getenv("PDC_SL_TIMESORT"), *s = getenv("PDC_SL_SLICES");
        return !(e && e[0] == '0') && !(s && s[0] == '0');
    }();
    return on;
}

// The streamed kernels' intake: sl_tame_kernel looks at t (tame? in order?) and writes the (t, m) records; samples that
// may be in any order are copied, ordered by time on the device if the kernel found them out of order (timesort.inc -
// every launch of it returns at once otherwise) and the kernels downstream read the copies (sa.t / sa.m).
int stream_intake(hipStream_t st, const StreamShape &h, char *area, const double *d_t, const double *d_m, int64_t n, int hints,
                  stream::StreamArgs &sa) {
    unsigned *bad = const_cast<unsigned *>(sa.bad_t);
    fast::rec_t *tm = const_cast<fast::rec_t *>(sa.tm);
    PDC_HIP(hipMemsetAsync(area + h.o_bad, 0, 256, st));
    if ((hints & kHintOrdered) != 0 || !timesort_on()) {
        hipLaunchKernelGGL(stream::sl_tame_kernel, dim3(512), dim3(256), 0, st, d_t, d_m, tm, static_cast<double *>(nullptr),
                           static_cast<double *>(nullptr), n, bad);
        PDC_HIP(hipGetLastError());
        return PDC_OK;
    }
    const int64_t plane = (n * 8 + 255) & ~(int64_t)255;
    double *k0 = reinterpret_cast<double *>(area + h.o_ts), *v0 = reinterpret_cast<double *>(area + h.o_ts + plane);
    double *k1 = reinterpret_cast<double *>(area + h.o_ts + 2 * plane), *v1 = reinterpret_cast<double *>(area + h.o_ts + 3 * plane);
    hipLaunchKernelGGL(stream::sl_tame_kernel, dim3(512), dim3(256), 0, st, d_t, d_m, static_cast<fast::rec_t *>(nullptr), k0, v0, n, bad);
    tsort::Args ta;
    ta.bad = bad;
    ta.hist = reinterpret_cast<unsigned *>(area + h.o_rhist);
    ta.tot = reinterpret_cast<unsigned *>(area + h.o_rtot);
    ta.n = n;
    ta.tiles = h.ts_tiles;
    for (int pass = 0; pass < 8; ++pass) {
        ta.kin = (pass & 1) ? k1 : k0;
        ta.vin = (pass & 1) ? v1 : v0;
        ta.kout = (pass & 1) ? k0 : k1;
        ta.vout = (pass & 1) ? v0 : v1;
        ta.shift = 8 * pass;
        hipLaunchKernelGGL(tsort::ts_hist_kernel, dim3((unsigned)h.ts_tiles), dim3(tsort::kB), 0, st, ta);
        hipLaunchKernelGGL(tsort::ts_scan_kernel, dim3(256), dim3(tsort::kB), 0, st, ta);
        hipLaunchKernelGGL(tsort::ts_scatter_kernel, dim3((unsigned)h.ts_tiles), dim3(tsort::kB), 0, st, ta);
    }
    hipLaunchKernelGGL(tsort::ts_finish_kernel, dim3(512), dim3(256), 0, st, k0, v0, tm, n, bad);
    PDC_HIP(hipGetLastError());
    sa.t = k0;
    sa.m = v0;
    return PDC_OK;
}

// the four launches that counting-sort one batch of periods by phase bin and sort every bin (no link kernel)
int stream_sort_batch(int device, hipStream_t st, const StreamShape &h, stream::StreamArgs &sa, int64_t p0, int64_t bc) {
    sa.p0 = p0;
    sa.batch = (int)bc;
    PDC_REQUIRE(bc * h.s1 < ((int64_t)1 << 31), "stringlength: grid too large");
    const int s1p = h.s1 > 1024 ? 2048 : (h.s1 > 512 ? 1024 : 512);
    const size_t lds_a = stream::lds_part(s1p);
    // (period_and_group(): eight workgroups - one per XCD - per 8 / groups periods)
    const dim3 wg((unsigned)(h.groups > 8 ? bc * h.groups : 8 * ((bc + 8 / h.groups - 1) / (8 / h.groups))));
    hipLaunchKernelGGL(stream::sl_hist_kernel, wg, dim3(stream::kBA), 0, st, sa);
    hipLaunchKernelGGL(stream::sl_lut_kernel, dim3((unsigned)bc), dim3(256), (size_t)h.s1 * h.groups * 4, st, sa);
    if (sa.direct && !sa.sorted) hipLaunchKernelGGL(stream::sl_direct_kernel, dim3((unsigned)(bc * stream::kDirectW)), dim3(stream::kBB), 0, st, sa);
    // (every other period takes one of the two: a workgroup of the other kernel returns at once)
    if (s1p == 2048) hipLaunchKernelGGL(stream::sl_part_kernel<2048>, wg, dim3(stream::kBA), lds_a, st, sa);
    else if (s1p == 1024) hipLaunchKernelGGL(stream::sl_part_kernel<1024>, wg, dim3(stream::kBA), lds_a, st, sa);
    else hipLaunchKernelGGL(stream::sl_part_kernel<512>, wg, dim3(stream::kBA), lds_a, st, sa);
    if (sa.slices) hipLaunchKernelGGL(stream::sl_bound_kernel, wg, dim3(stream::kBA), 0, st, sa);
    const int64_t sort_slots = (int64_t)cu_count(device) * (stream::kLdsB + 1024 <= 80 * 1024 ? 2 : 1);
    const int64_t sort_grid = bc * h.s1 < sort_slots ? bc * h.s1 : sort_slots;
    hipLaunchKernelGGL(stream::sl_sort_kernel, dim3((unsigned)sort_grid), dim3(stream::kBB), stream::kLdsB, st, sa);
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}

int stream_allow_lds(const StreamShape &h) {
    PDC_TRY(allow_dynamic_lds((const void *)stream::sl_sort_kernel, (int)stream::kLdsB));
    PDC_REQUIRE((size_t)h.s1 * h.groups * 4 <= 128 * 1024, "stringlength: %d bins x %d groups do not fit the bin table kernel's LDS", h.s1, h.groups);
    PDC_TRY(allow_dynamic_lds((const void *)stream::sl_lut_kernel, 128 * 1024));
    if (h.s1 > 1024) PDC_TRY(allow_dynamic_lds((const void *)stream::sl_part_kernel<2048>, (int)stream::lds_part(2048)));
    else if (h.s1 > 512) PDC_TRY(allow_dynamic_lds((const void *)stream::sl_part_kernel<1024>, (int)stream::lds_part(1024)));
    else PDC_TRY(allow_dynamic_lds((const void *)stream::sl_part_kernel<512>, (int)stream::lds_part(512)));
    return PDC_OK;
}

// ---- Supersmoother: workspace -----------------------------------------------------------------------------
template <int KMAX, typename IdxT = unsigned short, int NB = fast::kNB, bool MULTI = false, int PL = 0, bool EMIT = false>
int launch_fast(const fast::FastArgs &a, int64_t grid, hipStream_t st) {
    constexpr bool P17 = PL > 0;
    const int64_t slice = MULTI ? a.slice_cap : a.n;
    const int64_t entries = (slice + 64 + 7) & ~(int64_t)7;
    const size_t lds = (size_t)fast::FL<IdxT>::fixed +
                       (P17 ? (size_t)entries * 2 + (size_t)PL * ((slice + 64 + 31) / 32) * 4 + 16 : (size_t)entries * sizeof(IdxT));
    PDC_REQUIRE(lds <= (size_t)fast::kLdsTotalDyn, "stringlength: slice of %lld samples does not fit LDS", (long long)slice);
    PDC_TRY(allow_dynamic_lds((const void *)fast::sl_fast_kernel<KMAX, IdxT, NB, MULTI, PL, EMIT>, fast::kLdsTotalDyn));
    hipLaunchKernelGGL((fast::sl_fast_kernel<KMAX, IdxT, NB, MULTI, PL, EMIT>), dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    return PDC_OK;
}

// ---- the one-workgroup-per-period kernels as a SORT (EMIT instances of sl_fast_kernel): the Supersmoother's sorted
// curves below the streamed kernels' range.  Workspace per batch: the kernels' scratch for grid_for(batch)
// workgroups, the (t, y) table, the one-cycle marks, a dummy length per period.
bool fast_sort_takes(int64_t n) {
    static const int max_slices = [] { const char *e = getenv("PDC_SL_FAST_SLICES"); return e ? atoi(e) : 16; }();
    static const bool on = [] { const char *e = getenv("PDC_SS_FASTSORT"); return !(e && e[0] == '0'); }();
    return on && n >= 64 && n < stream_min_n() && n <= max_slices * (int64_t)fast::FL<unsigned>::capacity;
}
int64_t fast_sort_bytes(int64_t n, int64_t batch) {
    const int64_t partition = may_need_partition(n) ? pad_pow2(n) * 4 + kBucketsLarge * 4 : 0;
    return scratch_bytes(n, batch, partition) + fast_table_bytes(n) + onecycle_bytes(batch) + ((batch * 8 + 255) & ~(int64_t)255) + 256;
}
// once per call: the AoS (t, y) table + its flags
void fast_sort_prepare(hipStream_t st, const double *d_t, const double *d_y, int64_t n, int64_t batch, void *work) {
    const int64_t partition = may_need_partition(n) ? pad_pow2(n) * 4 + kBucketsLarge * 4 : 0;
    char *table = static_cast<char *>(work) + scratch_bytes(n, batch, partition);
    hipLaunchKernelGGL(fast::sl_prep_kernel, dim3(1), dim3(kBlock), 0, st, d_t, d_y, (int)n, reinterpret_cast<fast::rec_t *>(table),
                       reinterpret_cast<unsigned *>(table + ((n * 16 + 255) & ~(int64_t)255)));
}
// one batch: periods d_periods[0 .. bc) -> sorted[q][n]; skip_out = the one-cycle marks (those rows are NOT written here)
int fast_sort_batch(hipStream_t st, const double *d_t, const double *d_y, int64_t n, const double *d_periods, int64_t bc,
                    int64_t batch, const unsigned *bad, fast::rec_t *sorted, void *work, const unsigned char **skip_out) {
    const int64_t grid = grid_for(bc);
    const int64_t n_pad = pad_pow2(n), nr_pad = range_slots(n);
    const int64_t gridc = grid_for(batch);            // (the layout is that of a full batch)
    const int64_t partition = may_need_partition(n) ? n_pad * 4 + kBucketsLarge * 4 : 0;
    fast::FastArgs f;
    f.t = d_t;
    f.m = d_y;
    f.periods = d_periods;
    f.n = n;
    f.n_periods = bc;
    f.n_pad = n_pad;
    f.nr_pad = nr_pad;
    f.gkeys = reinterpret_cast<unsigned long long *>(work);
    f.rsum = reinterpret_cast<double *>(f.gkeys + gridc * n_pad);
    f.gidx = reinterpret_cast<unsigned *>(f.rsum + gridc * nr_pad * 4);
    f.rcnt = reinterpret_cast<int *>(f.gidx + gridc * n_pad);
    f.rlen = reinterpret_cast<double *>(f.rcnt + gridc * nr_pad);
    unsigned *gorder = reinterpret_cast<unsigned *>(f.rlen + gridc * nr_pad);
    f.ghist = gorder + (may_need_partition(n) ? gridc * n_pad : 0);
    f.gbucket = reinterpret_cast<unsigned short *>(gorder);
    char *table = static_cast<char *>(work) + scratch_bytes(n, batch, partition);
    f.rec = reinterpret_cast<const fast::rec_t *>(table);
    f.flags = reinterpret_cast<const unsigned *>(table + ((n * 16 + 255) & ~(int64_t)255));
    char *oc = table + fast_table_bytes(n);
    onecycle::OneArgs o;
    o.t = d_t;
    o.m = d_y;
    o.periods = d_periods;
    o.n = n;
    o.n_periods = bc;
    o.bad = bad;
    o.skip = reinterpret_cast<unsigned char *>(oc);
    o.list = reinterpret_cast<unsigned *>(oc + ((batch + 255) & ~(int64_t)255));
    o.count = reinterpret_cast<unsigned *>(oc + ((batch + 255) & ~(int64_t)255) + ((batch * 4 + 255) & ~(int64_t)255));
    o.ell = nullptr;
    PDC_HIP(hipMemsetAsync(o.count, 0, 256, st));
    hipLaunchKernelGGL(onecycle::sl_onecycle_mark_kernel, dim3((unsigned)((bc + 255) / 256)), dim3(256), 0, st, o);
    f.ell = reinterpret_cast<double *>(oc + onecycle_bytes(batch));
    f.todo = nullptr;
    f.todo_count = nullptr;
    f.skip = o.skip;
    f.sorted = sorted;
    f.slice_cap = fast::FL<unsigned>::capacity;
    *skip_out = o.skip;
    const int k = (int)((n + kBlock - 1) / kBlock);
    if (n > fast::kCapacity && n < 65536) {
        f.slice_cap = fast::FL<unsigned short>::capacity;
        PDC_TRY((launch_fast<4, unsigned short, fast::kNBLarge, true, 0, true>(f, grid, st)));
    } else if (n > fast::kCapacity && n < 131072) {
        f.slice_cap = fast::kCapacity17;
        PDC_TRY((launch_fast<4, unsigned, fast::kNBLarge, true, 1, true>(f, grid, st)));
    } else if (n > fast::kCapacity && n < 262144) {
        f.slice_cap = fast::kCapacity18;
        PDC_TRY((launch_fast<4, unsigned, fast::kNBLarge, true, 2, true>(f, grid, st)));
    } else if (n > fast::kCapacity) {
        PDC_TRY((launch_fast<4, unsigned, fast::kNBLarge, true, 0, true>(f, grid, st)));
    } else if (k <= 8) {
        PDC_TRY((launch_fast<8, unsigned short, fast::kNB, false, 0, true>(f, grid, st)));
    } else if (k <= 20) {
        PDC_TRY((launch_fast<20, unsigned short, fast::kNB, false, 0, true>(f, grid, st)));
    } else if (k <= 36) {
        PDC_TRY((launch_fast<36, unsigned short, fast::kNB, false, 0, true>(f, grid, st)));
    } else {
        PDC_TRY((launch_fast<fast::kKMax, unsigned short, fast::kNB, false, 0, true>(f, grid, st)));
    }
    return PDC_OK;
}

struct SsShape {
    StreamShape h;
    bool streamed, tiled, fastsort;
    int batch, grid_ss, grid_fb, sb, seg, seg_len, seg34, seg_len34;
    int64_t stride, n_pad, o_sorted, o_scratch, o_gk, o_gi, o_bad, o_sm, o_arec, o_srec, o_flag, o_part, o_fast, total;
};
SsShape ss_shape(int64_t n, int64_t n_periods, bool lists = true) {
    auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    SsShape z;
    z.fastsort = fast_sort_takes(n);
    z.o_fast = 0;
    z.streamed = !z.fastsort && n >= 4096 && n <= kStreamMaxN;
    z.tiled = n >= ss2::kMinN && n <= ss2::kMaxN;
    // the sorted batch - 16 bytes a point and period - stays within 2 GB (batches of >= 8 periods)
    const double scale = work_scale();                                     // (< 1 under a workspace budget)
    int64_t cap = (int64_t)((double)((int64_t)2 << 30) * scale) / (16 * (n > 0 ? n : 1));
    cap = cap < 8 ? 8 : cap;
    // (the bin lists of the streamed sort take ~49 bytes per point and period: ~350 MB of them - 136 periods at
    // n = 5e4 -, but at least one full sub-batch of the smoother, and the streamed kernels' own 384 from n = 18 000 down)
    static const int64_t env_cap = [] { const char *e = getenv("PDC_SS_BATCH"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
    int64_t sub = (int64_t)((double)((int64_t)3 << 29) * scale) / (72 * (n > 0 ? n : 1)) / 8 * 8;     // the smoother's sub-batch (below)
    static const int64_t env_submax = [] { const char *e = getenv("PDC_SS_SUBMAX"); return e ? (int64_t)atoll(e) : (int64_t)0; }();
    const int64_t sub_max = env_submax >= 8 ? env_submax : (n >= 40000 ? ss2::kSubBatch : (n >= 10000 ? 128 : 384));
    sub = sub < 8 ? 8 : (sub > sub_max ? sub_max : sub);
    int64_t list_cap = (int64_t)(7000000.0 * scale) / (n > 0 ? n : 1);
    list_cap = list_cap < sub ? sub : (list_cap > 384 ? 384 : list_cap / sub * sub);
    if (z.fastsort) {
        // (no lists: one workgroup per period - 256 periods a launch fill the CUs - within ~600 MB of sorted curves
        // + the kernels' per-workgroup scratch, and at least one sub-batch of the smoother)
        const int64_t partition = may_need_partition(n) ? pad_pow2(n) * 4 + kBucketsLarge * 4 : 0;
        const int64_t per = 16 * n + pad_pow2(n) * 12 + range_slots(n) * 44 + partition;
        int64_t b = (int64_t)((double)((int64_t)600 << 20) * scale) / per;
        b = b > 256 ? 256 : b;
        list_cap = b / sub * sub > 0 ? b / sub * sub : sub;
    }
    if (env_cap > 0) list_cap = env_cap;
    cap = cap < list_cap ? cap : list_cap;
    int64_t at = 0;
    if (z.streamed) {
        z.h = stream_shape(n, n_periods < cap ? n_periods : cap, lists);
        z.batch = z.h.batch;
        at = up(z.h.total);
    } else {
        z.h = StreamShape{};
        int64_t b = n_periods < 512 ? n_periods : 512;
        b = b < cap ? b : cap;
        z.batch = (int)(b < 1 ? 1 : b);
        if (z.fastsort) {
            z.o_fast = 0;
            at = (fast_sort_bytes(n, z.batch) + 255) & ~(int64_t)255;
        }
    }
    z.stride = (n + n / 2 + 24 + 7) & ~(int64_t)7;   // (prefix arrays run over the curve extended by a quarter on either side)
    // The generic smoother (ss_smooth_kernel: fifteen arrays of 1.5 n doubles per workgroup) is the whole path below
    // ss2::kMinN samples and, above, takes only the periods the tiled kernels hand back: 1 GB of arrays for it.
    int64_t g = (int64_t)((double)((int64_t)1 << 30) * scale) / (ss::kArrays * z.stride * 8);
    g = g < 1 ? 1 : (g > (z.tiled ? 16 : 512) ? (z.tiled ? 16 : 512) : g);
    z.grid_ss = (int)(g < z.batch ? g : z.batch);
    z.grid_fb = z.batch < (z.streamed ? 64 : 256) ? z.batch : (z.streamed ? 64 : 256);
    if (z.fastsort) z.grid_fb = 1;   // (the whole-period bitonic fallback is not used)
    // Periods whose phases cluster are sorted by one workgroup each in global scratch (12 bytes a padded point): the
    // pool of such workgroups stays within 1 GB
    const int64_t fb_cap = (int64_t)((double)((int64_t)1 << 30) * scale) / (pad_pow2(n) * 12);
    z.grid_fb = (int)(z.grid_fb < fb_cap ? z.grid_fb : (fb_cap < 1 ? 1 : fb_cap));
    z.n_pad = pad_pow2(n);
    // tiled smoother: sub-batches of <= 64 periods from n = 4e4 on (short curves take more per launch - a period of
    // 4096 points is three tiles), 72 bytes of intermediates per point and period within 1.5 GB
    int64_t sb = sub < z.batch ? sub : (z.batch + 7) / 8 * 8;
    static const int env_sb = [] { const char *e = getenv("PDC_SS_SB"); return e ? atoi(e) : 0; }();
    static const int env_seg = [] { const char *e = getenv("PDC_SS_SEG"); return e ? atoi(e) : 0; }();
    if (env_sb >= 8 && env_sb % 8 == 0 && env_sb < sb) sb = env_sb;
    z.sb = (int)sb;
    const int64_t tiles = (n + ss2::kSegUnit - 1) / ss2::kSegUnit;
    // one workgroup per CU in the first two sweeps: 256 / sub-batch segments, at least 4 (measured at n = 5e4, 64
    // periods a sub-batch: 2 / 4 / 8 / 16 segments 37.4 / 26.0 / 28.8 / 31.7 ms - every segment sums its first window
    // directly, ~15 us of a workgroup's ~60), 16 at n = 1e6 where a sub-batch is 16 periods
    int seg_rule = (int)((256 + sb - 1) / sb);
    seg_rule = seg_rule < 4 ? 4 : (seg_rule > ss2::kSegMax ? ss2::kSegMax : seg_rule);
    const int seg_max = env_seg >= 1 && env_seg <= ss2::kSegMax ? env_seg : seg_rule;
    z.seg = (int)(tiles < seg_max ? (tiles < 1 ? 1 : tiles) : seg_max);
    z.seg_len = (int)((tiles + z.seg - 1) / z.seg * ss2::kSegUnit);
    static const int env_seg34 = [] { const char *e = getenv("PDC_SS_SEG34"); return e ? atoi(e) : 0; }();
    const int seg34_max = env_seg34 >= 1 && env_seg34 <= ss2::kSegMax ? env_seg34 : z.seg;
    z.seg34 = (int)(tiles < seg34_max ? (tiles < 1 ? 1 : tiles) : seg34_max);
    z.seg_len34 = (int)((tiles + z.seg34 - 1) / z.seg34 * ss2::kSegUnit);
    z.o_sorted = at;
    z.o_scratch = z.o_sorted + up((int64_t)z.batch * n * 16);
    z.o_gk = z.o_scratch + up((int64_t)z.grid_ss * ss::kArrays * z.stride * 8);
    z.o_gi = z.o_gk + up((int64_t)z.grid_fb * z.n_pad * 8);
    z.o_bad = z.o_gi + up((int64_t)z.grid_fb * z.n_pad * 4);
    z.o_sm = z.o_bad + 256;
    z.o_arec = z.o_sm + (z.tiled ? up(3 * sb * n * 8) : 0);
    z.o_srec = z.o_arec + (z.tiled ? up(sb * n * 32) : 0);
    z.o_flag = z.o_srec + (z.tiled ? up(sb * n * 16) : 0);
    z.o_part = z.o_flag + up((int64_t)z.batch * 4);
    z.total = z.o_part + up((int64_t)z.batch * ss2::kSegMax * 8);
    return z;
}


}  // namespace

extern "C" {

}  // extern "C"

namespace {
int64_t stringlength_work_bytes(int64_t n, int64_t n_periods, bool lists) {
    if (n < 0 || n_periods < 0) return -1;
    const int64_t partition = may_need_partition(n) ? pad_pow2(n) * 4 + kBucketsLarge * 4 : 0;
    return scratch_bytes(n, n_periods, partition) + fast_table_bytes(n) + duo_bytes(n_periods) + onecycle_bytes(n_periods) +
           stream_bytes(n, n_periods, lists);
}

// Host-side test for pdc_stringlength_scan (host arrays in hand): will EVERY period take the slices or the one-cycle
// mode of the streamed kernels?  Then the workspace needs no lists - at N = 1e6 12 GB of them, whose allocation alone
// made a process's first call take 1.2 s.  The same conditions as sl_lut_kernel's with the bins reserved (s1) in place
// of the bins used: stricter, never laxer.
// Samples out of order (the C ABI allows it; a TSeries never is): the streamed kernels order them by time on the device
// first (timesort.inc), so the same test applies with the smallest / largest time stamp in place of the first / last -
// unless some time stamp is not tame (NaN, infinite, beyond 1e+-150: no time sort, lists mode as it stands).
bool host_all_slices(const double *t, int64_t n, const double *periods, int64_t n_periods, int64_t min_n, bool *ordered) {
    static const bool on = [] { const char *e = getenv("PDC_SL_SLICES"); return !(e && e[0] == '0'); }();
    *ordered = false;
    if (!on || n < min_n || n < 4096 || n > kStreamMaxN || n_periods < 1) return false;
    bool in_order = true, tame = true;
    double lo = t[0], hi = t[0];
    for (int64_t i = 0; i < n; ++i) {
        const double v = t[i], av = std::fabs(v);
        tame = tame && (av == 0.0 || (av >= 1e-150 && av <= 1e150));
        in_order = in_order && (i == 0 || t[i - 1] <= v);
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    *ordered = in_order;
    if (!in_order && !(tame && timesort_on() && n >= stream_min_n())) return false;
    const double s1 = (double)stream_shape(n, n_periods, false).s1;
    for (int64_t p = 0; p < n_periods; ++p) {
        const double period = periods[p];
        if (!(period > 0.0)) return false;
        const double cycles = std::floor(hi / period) - std::floor(lo / period) + 1.0;
        if (!(cycles >= 1.0 && cycles < (double)stream::kCycS && cycles * s1 * (double)stream::kMinSlice <= (double)n)) return false;
    }
    return true;
}
int host_hints(const double *t, int64_t n, const double *periods, int64_t n_periods, int64_t min_n) {
    bool ordered = false;
    const bool all_slices = host_all_slices(t, n, periods, n_periods, min_n, &ordered);
    return (all_slices ? 0 : kHintLists) | (ordered ? kHintOrdered : 0);
}

int stringlength_scan_impl(int device, void *stream, const double *d_t, const double *d_m,
                           int64_t n, const double *d_periods, int64_t n_periods, double *d_ell,
                           void *work, int64_t work_bytes, int hints);
}  // namespace

extern "C" {

int64_t pdc_stringlength_work_bytes(int64_t n, int64_t n_periods) {
    WorkScale ws(work_budget(), [&] { return stringlength_work_bytes(n, n_periods, true); });   // (PDC_WORK_BUDGET_GB)
    return ws.need;
}

int pdc_stringlength_scan_dev(int device, void *stream, const double *d_t, const double *d_m,
                              int64_t n, const double *d_periods, int64_t n_periods, double *d_ell,
                              void *work, int64_t work_bytes) {
    return stringlength_scan_impl(device, stream, d_t, d_m, n, d_periods, n_periods, d_ell, work, work_bytes, kHintLists);
}

}  // extern "C"

namespace {
int stringlength_scan_impl(int device, void *stream, const double *d_t, const double *d_m,
                           int64_t n, const double *d_periods, int64_t n_periods, double *d_ell,
                           void *work, int64_t work_bytes, int hints) {
    const bool lists = (hints & kHintLists) != 0;
    PDC_REQUIRE(d_t && d_m && (d_periods || n_periods == 0) && (d_ell || n_periods == 0),
                "stringlength: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "stringlength: negative size");
    PDC_REQUIRE(n < ((int64_t)1 << 31), "stringlength: at most 2^31-1 samples");
    PDC_REQUIRE(n_periods < ((int64_t)1 << 32), "stringlength: at most 2^32-1 trial periods per call");
    WorkScale ws(work_budget(), [&] { return stringlength_work_bytes(n, n_periods, lists); });   // (a host entry's scope, if any, stays)
    PDC_REQUIRE_FITS(ws, "stringlength");
    PDC_REQUIRE(work && work_bytes >= ws.need, "stringlength: workspace too small (%lld < %lld bytes)", (long long)work_bytes,
                (long long)ws.need);
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
    a.nr_pad = range_slots(n);
    a.gkeys = reinterpret_cast<unsigned long long *>(work);
    a.rsum = reinterpret_cast<double *>(a.gkeys + grid * a.n_pad);
    a.gidx = reinterpret_cast<unsigned *>(a.rsum + grid * a.nr_pad * 4);
    a.rcnt = reinterpret_cast<int *>(a.gidx + grid * a.n_pad);
    double *rlen = reinterpret_cast<double *>(a.rcnt + grid * a.nr_pad);   // [grid][nr_pad] (fast kernels)
    a.gorder = reinterpret_cast<unsigned *>(rlen + grid * a.nr_pad);
    a.ghist = a.gorder + (may_need_partition(n) ? grid * a.n_pad : 0);
    hipStream_t st = (hipStream_t)stream;
    static const bool general_only = [] { const char *e = getenv("PDC_SL_GENERAL"); return e && e[0] == '1'; }();
    if (!general_only && stream_takes(n, n_periods)) {
        // ---- streamed path: histogram -> bin table -> partition -> sort every bin in LDS -> link, batch by batch ----
        const StreamShape h = stream_shape(n, n_periods, lists);
        const int64_t partition = may_need_partition(n) ? pad_pow2(n) * 4 + kBucketsLarge * 4 : 0;
        char *area = static_cast<char *>(work) + scratch_bytes(n, n_periods, partition) + fast_table_bytes(n) +
                     duo_bytes(n_periods) + onecycle_bytes(n_periods);
        stream::StreamArgs sa = stream_args(h, area, d_t, d_m, d_periods, n, d_ell);
        sa.no_lists = lists ? 0 : 1;
        PDC_HIP(hipMemsetAsync(sa.todo_count, 0, 256, st));
        PDC_TRY(stream_intake(st, h, area, d_t, d_m, n, hints, sa));
        a.t = sa.t;     // (the general kernel's few periods see the samples the streamed kernels saw: one tie order)
        a.m = sa.m;
        PDC_TRY(stream_allow_lds(h));
        for (int64_t p0 = 0; p0 < n_periods; p0 += h.batch) {
            const int64_t bc = n_periods - p0 < h.batch ? n_periods - p0 : h.batch;
            PDC_TRY(stream_sort_batch(device, st, h, sa, p0, bc));
            hipLaunchKernelGGL(stream::sl_link_kernel, dim3((unsigned)((bc + 3) / 4)), dim3(256), 0, st, sa);
            PDC_HIP(hipGetLastError());
        }
        static const bool dbg = [] { const char *e = getenv("PDC_SL_STREAM_DEBUG"); return e && e[0] == '1'; }();
        if (dbg) {
            unsigned left = 0;
            PDC_HIP(hipStreamSynchronize(st));
            PDC_HIP(hipMemcpy(&left, sa.todo_count, 4, hipMemcpyDeviceToHost));
            fprintf(stderr, "sl stream: n=%lld periods=%lld s1=%d groups=%d tiles_w=%d batch=%d -> %u periods left to the general kernel\n",
                    (long long)n, (long long)n_periods, h.s1, h.groups, h.tiles_w, h.batch, left);
            if (left) {
                std::vector<unsigned char> td((size_t)n_periods);
                std::vector<double> pp((size_t)n_periods);
                PDC_HIP(hipMemcpy(td.data(), sa.todo, (size_t)n_periods, hipMemcpyDeviceToHost));
                PDC_HIP(hipMemcpy(pp.data(), d_periods, (size_t)n_periods * 8, hipMemcpyDeviceToHost));
                for (int64_t p = 0, shown = 0; p < n_periods && shown < 16; ++p)
                    if (td[p]) {
                        fprintf(stderr, "   period %lld = %.17g: flag bits 0x%x\n", (long long)p, pp[p], td[p] & 0x7f);
                        ++shown;
                    }
            }
        }
        // the periods a coarse bucket was too heavy in (clustered phases): the general kernel, which sorts what LDS
        // cannot hold
        a.todo = sa.todo;
        a.todo_count = sa.todo_count;
        if (n < 65536) {
            using L = Lds<unsigned short>;
            const int64_t slice = n < L::capacity ? n : L::capacity;
            const size_t lds = (size_t)L::fixed + (size_t)((slice + 7) & ~(int64_t)7) * 2;
            PDC_TRY(allow_dynamic_lds((const void *)sl_scan_kernel<unsigned short, kBuckets>, (int)lds));
            hipLaunchKernelGGL((sl_scan_kernel<unsigned short, kBuckets>), dim3((unsigned)grid), dim3(kBlock), lds, st, a);
        } else {
            using L = Lds<unsigned>;
            const size_t lds = (size_t)L::fixed + (size_t)L::capacity * 4;
            PDC_TRY(allow_dynamic_lds((const void *)sl_scan_kernel<unsigned, kBucketsLarge>, (int)lds));
            hipLaunchKernelGGL((sl_scan_kernel<unsigned, kBucketsLarge>), dim3((unsigned)grid), dim3(kBlock), lds, st, a);
        }
        PDC_HIP(hipGetLastError());
        return PDC_OK;
    }
    // ---- periods of one cycle: summed as the samples stand, passed over by the kernels below (namespace onecycle) ----
    static const bool onecycle_on = [] { const char *e = getenv("PDC_SL_SLICES"); return !(e && e[0] == '0'); }();
    if (onecycle_on && n >= 2) {
        const int64_t partition = may_need_partition(n) ? pad_pow2(n) * 4 + kBucketsLarge * 4 : 0;
        char *oc = static_cast<char *>(work) + scratch_bytes(n, n_periods, partition) + fast_table_bytes(n) + duo_bytes(n_periods);
        onecycle::OneArgs o;
        o.t = d_t;
        o.m = d_m;
        o.periods = d_periods;
        o.n = n;
        o.n_periods = n_periods;
        o.skip = reinterpret_cast<unsigned char *>(oc);
        o.list = reinterpret_cast<unsigned *>(oc + ((n_periods + 255) & ~(int64_t)255));
        unsigned *flags = reinterpret_cast<unsigned *>(oc + ((n_periods + 255) & ~(int64_t)255) + ((n_periods * 4 + 255) & ~(int64_t)255));
        o.bad = flags;
        o.count = flags + 64;
        o.ell = d_ell;
        PDC_HIP(hipMemsetAsync(flags, 0, 512, st));
        hipLaunchKernelGGL(stream::sl_tame_kernel, dim3(n < 65536 ? 64 : 512), dim3(256), 0, st, d_t, d_m,
                           static_cast<fast::rec_t *>(nullptr), static_cast<double *>(nullptr), static_cast<double *>(nullptr), n, flags);
        hipLaunchKernelGGL(onecycle::sl_onecycle_mark_kernel, dim3((unsigned)((n_periods + 255) / 256)), dim3(256), 0, st, o);
        hipLaunchKernelGGL(onecycle::sl_onecycle_kernel, dim3(64), dim3(kBlock), 0, st, o);
        a.skip = o.skip;
    }
    // beyond ~16 slices the general kernel (one-off grouping of the indices in global scratch) is ahead: the
    // fast kernel's per-slice passes over the bucket ids and its 16-byte gathers (table > L2) grow with N
    // (x 2048 periods: N = 3.3e5 20.5 against 23.2 ms, N = 4.5e5 39.3 against 35.3 ms); PDC_SL_FAST_SLICES moves it
    static const int max_slices = [] { const char *e = getenv("PDC_SL_FAST_SLICES"); return e ? atoi(e) : 16; }();
    if (n >= 1 && !general_only && n <= max_slices * (int64_t)fast::FL<unsigned>::capacity) {
        fast::FastArgs f;
        f.t = d_t;
        f.m = d_m;
        f.periods = d_periods;
        f.n = n;
        f.n_periods = n_periods;
        f.ell = d_ell;
        f.gkeys = a.gkeys;
        f.gidx = a.gidx;
        f.rsum = a.rsum;
        f.rcnt = a.rcnt;
        f.rlen = rlen;
        f.n_pad = a.n_pad;
        f.nr_pad = a.nr_pad;
        f.ghist = a.ghist;
        f.gbucket = reinterpret_cast<unsigned short *>(a.gorder);   // (the general kernel's grouping area: [grid][n_pad] words)
        const int64_t partition = may_need_partition(n) ? pad_pow2(n) * 4 + kBucketsLarge * 4 : 0;
        char *table = static_cast<char *>(work) + scratch_bytes(n, n_periods, partition);
        f.rec = reinterpret_cast<const fast::rec_t *>(table);
        f.flags = reinterpret_cast<const unsigned *>(table + ((n * 16 + 255) & ~(int64_t)255));
        hipLaunchKernelGGL(fast::sl_prep_kernel, dim3(1), dim3(kBlock), 0, st, d_t, d_m, (int)n,
                           reinterpret_cast<fast::rec_t *>(table), const_cast<unsigned *>(f.flags));
        const int k = (int)((n + kBlock - 1) / kBlock);
        f.slice_cap = fast::FL<unsigned>::capacity;
        f.todo = nullptr;
        f.todo_count = nullptr;
        f.skip = a.skip;
        f.sorted = nullptr;
        // Two workgroups per CU (sl_duo_kernel) whenever a period's permutation fits half of LDS;
        // PDC_SL_DUO=0 keeps the one-workgroup kernel (A/B, tests)
        static const bool duo_on = [] { const char *e = getenv("PDC_SL_DUO"); return !(e && e[0] == '0'); }();
        if (duo_on && n <= duo::kCapD) {
            duo::DuoArgs d;
            d.t = d_t;
            d.m = d_m;
            d.periods = d_periods;
            d.rec = f.rec;
            d.flags = f.flags;
            d.n = (int)n;
            d.n_periods = n_periods;
            d.ell = d_ell;
            char *area = table + fast_table_bytes(n);
            unsigned char *todo = reinterpret_cast<unsigned char *>(area);
            d.todo = todo;
            d.skip = a.skip;
            d.ticket = reinterpret_cast<unsigned *>(area + ((n_periods + 255) & ~(int64_t)255));
            d.rsum = a.rsum;
            d.rcnt = a.rcnt;
            d.rlen = rlen;
            d.nr_pad = a.nr_pad;
            static const bool quad_on = [] { const char *e = getenv("PDC_SL_QUAD"); return !(e && e[0] == '0'); }();
            const bool quad = quad_on && n <= duo::kCapQ;   // four 256-thread workgroups per CU
            int64_t dgrid = (quad ? 4 : 2) * (int64_t)cu_count(device);
            dgrid = dgrid < grid ? dgrid : grid;       // (the range scratch is laid out for `grid` workgroups)
            PDC_HIP(hipMemsetAsync(d.ticket, 0, 8, st));
            const int kd = (int)((n + duo::kB - 1) / duo::kB);
            if (quad) PDC_TRY((launch_duo<16, 256, 512>(d, dgrid, st)));
            else if (kd <= 16) PDC_TRY(launch_duo<16>(d, dgrid, st));
            else if (kd <= 36) PDC_TRY(launch_duo<36>(d, dgrid, st));
            else PDC_TRY(launch_duo<52>(d, dgrid, st));
            f.todo = todo;   // the periods the duo kernel marked (clustered phases) go through the one-slice kernel
            f.todo_count = d.ticket + 1;
        }
        if (n > fast::kCapacity && n < 65536) {
            // (sample indices still fit 16 bits: slices of 52 112 instead of 23 976 - two slices, not three)
            f.slice_cap = fast::FL<unsigned short>::capacity;
            PDC_TRY((launch_fast<4, unsigned short, fast::kNBLarge, true>(f, grid, st)));
        } else if (n > fast::kCapacity && n < 262144) {
            // (17- / 18-bit sample indices: 16 bits per entry + one / two bit planes - slices of ~45 000 / ~42 700)
            static const bool p17_on = [] { const char *e = getenv("PDC_SL_P17"); return !(e && e[0] == '0'); }();
            if (p17_on && n < 131072) {
                f.slice_cap = fast::kCapacity17;
                PDC_TRY((launch_fast<4, unsigned, fast::kNBLarge, true, 1>(f, grid, st)));
            } else if (p17_on) {
                f.slice_cap = fast::kCapacity18;
                PDC_TRY((launch_fast<4, unsigned, fast::kNBLarge, true, 2>(f, grid, st)));
            } else {
                PDC_TRY((launch_fast<4, unsigned, fast::kNBLarge, true>(f, grid, st)));
            }
        } else if (n > fast::kCapacity) PDC_TRY((launch_fast<4, unsigned, fast::kNBLarge, true>(f, grid, st)));
        else if (k <= 8) PDC_TRY(launch_fast<8>(f, grid, st));
        else if (k <= 20) PDC_TRY(launch_fast<20>(f, grid, st));
        else if (k <= 36) PDC_TRY(launch_fast<36>(f, grid, st));
        else PDC_TRY(launch_fast<fast::kKMax>(f, grid, st));
    } else if (n < 65536) {
        using L = Lds<unsigned short>;
        const int64_t slice = n < L::capacity ? n : L::capacity;
        const size_t lds = (size_t)L::fixed + (size_t)((slice + 7) & ~(int64_t)7) * 2;
        PDC_TRY(allow_dynamic_lds((const void *)sl_scan_kernel<unsigned short, kBuckets>, (int)lds));
        hipLaunchKernelGGL((sl_scan_kernel<unsigned short, kBuckets>), dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    } else {
        using L = Lds<unsigned>;
        const size_t lds = (size_t)L::fixed + (size_t)L::capacity * 4;
        PDC_TRY(allow_dynamic_lds((const void *)sl_scan_kernel<unsigned, kBucketsLarge>, (int)lds));
        hipLaunchKernelGGL((sl_scan_kernel<unsigned, kBucketsLarge>), dim3((unsigned)grid), dim3(kBlock), lds, st, a);
    }
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}
}  // namespace

extern "C" {

int pdc_stringlength_scan(const double *t, const double *m, int64_t n, const double *periods,
                          int64_t n_periods, double *ell_out, int device) {
    PDC_REQUIRE(t && m && (periods || n_periods == 0) && (ell_out || n_periods == 0),
                "stringlength: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "stringlength: negative size");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    // (time-ordered samples and periods that all take the slices / one-cycle modes: no lists in the workspace)
    const int hints = host_hints(t, n, periods, n_periods, stream_min_n());
    // (the workspace fitted to PDC_WORK_BUDGET_GB and to what the device has free right now)
    WorkScale ws(host_work_budget(device), [&] { return stringlength_work_bytes(n, n_periods, (hints & kHintLists) != 0); });
    PDC_REQUIRE_FITS(ws, "stringlength");
    const int64_t wb = ws.need;
    void *d_t, *d_m, *d_p, *d_e, *d_w;
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_m));
    PDC_TRY(cached(device, SLOT_IN2, n_periods * 8, &d_p));
    PDC_TRY(cached(device, SLOT_OUT0, n_periods * 8, &d_e));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_w));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_m, m, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_p, periods, n_periods * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(stringlength_scan_impl(device, st, (double *)d_t, (double *)d_m, n, (double *)d_p,
                                   n_periods, (double *)d_e, d_w, wb, hints));
    PDC_HIP(hipMemcpyAsync(ell_out, d_e, n_periods * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

// ---- Supersmoother period search (spectral.py:8, a TODO upstream; Friedman 1984 + Reimann 1994) -----------------
int64_t pdc_supersmoother_work_bytes(int64_t n, int64_t n_periods) {
    if (n < 0 || n_periods < 0) return -1;
    WorkScale ws(work_budget(), [&] { return ss_shape(n, n_periods).total; });   // (PDC_WORK_BUDGET_GB)
    return ws.need;
}

}  // extern "C"

namespace {
int supersmoother_scan_impl(int device, void *stream, const double *d_t, const double *d_y, int64_t n,
                            const double *d_periods, int64_t n_periods, double alpha, double *d_stat, void *work,
                            int64_t work_bytes, int hints);
}

extern "C" {

int pdc_supersmoother_scan_dev(int device, void *stream, const double *d_t, const double *d_y, int64_t n,
                               const double *d_periods, int64_t n_periods, double alpha, double *d_stat, void *work,
                               int64_t work_bytes) {
    return supersmoother_scan_impl(device, stream, d_t, d_y, n, d_periods, n_periods, alpha, d_stat, work, work_bytes, kHintLists);
}

}  // extern "C"

namespace {
int supersmoother_scan_impl(int device, void *stream, const double *d_t, const double *d_y, int64_t n,
                            const double *d_periods, int64_t n_periods, double alpha, double *d_stat, void *work,
                            int64_t work_bytes, int hints) {
    const bool lists = (hints & kHintLists) != 0;
    PDC_REQUIRE(d_t && d_y && (d_periods || n_periods == 0) && (d_stat || n_periods == 0), "supersmoother: NULL argument");
    PDC_REQUIRE(n >= 5 && n < ((int64_t)1 << 30), "supersmoother: between 5 and 2^30 samples (the woofer window spans "
                                                  "half the curve)");
    PDC_REQUIRE(n_periods >= 0, "supersmoother: negative size");
    PDC_REQUIRE(alpha >= 0.0 && alpha <= 10.0, "supersmoother: the bass control alpha lies in [0, 10] (0 = off)");
    WorkScale ws(work_budget(), [&] { return ss_shape(n, n_periods, lists).total; });   // (a host entry's scope, if any, stays)
    PDC_REQUIRE_FITS(ws, "supersmoother");
    const SsShape z = ss_shape(n, n_periods, lists);
    PDC_REQUIRE(work && work_bytes >= z.total, "supersmoother: workspace too small (%lld < %lld bytes)",
                (long long)work_bytes, (long long)z.total);
    if (n_periods == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    hipStream_t st = (hipStream_t)stream;
    char *base = static_cast<char *>(work);
    fast::rec_t *sorted = reinterpret_cast<fast::rec_t *>(base + z.o_sorted);
    unsigned *bad = reinterpret_cast<unsigned *>(base + (z.streamed ? z.h.o_bad : z.o_bad));
    stream::StreamArgs sa;
    if (z.streamed) {
        sa = stream_args(z.h, base, d_t, d_y, d_periods, n, nullptr);
        sa.sorted = sorted;
        sa.direct = sa.slices;             // (one-cycle periods are marked by the bin table kernel: ss_direct_kernel writes them)
        sa.no_lists = lists ? 0 : 1;
        PDC_TRY(stream_allow_lds(z.h));
        PDC_TRY(stream_intake(st, z.h, base, d_t, d_y, n, hints, sa));   // (samples in any order: ordered by time first)
    } else {
        PDC_HIP(hipMemsetAsync(bad, 0, 256, st));
        hipLaunchKernelGGL(stream::sl_tame_kernel, dim3(512), dim3(256), 0, st, d_t, d_y, static_cast<fast::rec_t *>(nullptr),
                           static_cast<double *>(nullptr), static_cast<double *>(nullptr), n, bad);
    }
    ss::SsSortArgs fa;
    fa.t = z.streamed ? sa.t : d_t;
    fa.y = z.streamed ? sa.m : d_y;
    fa.periods = d_periods;
    fa.bad_t = bad;
    fa.n = n;
    fa.n_pad = z.n_pad;
    fa.flag = z.streamed ? sa.flag : nullptr;
    fa.skip8 = nullptr;
    if (z.fastsort) fast_sort_prepare(st, d_t, d_y, n, z.batch, base + z.o_fast);
    fa.gkeys = reinterpret_cast<unsigned long long *>(base + z.o_gk);
    fa.gidx = reinterpret_cast<unsigned *>(base + z.o_gi);
    fa.sorted = sorted;
    ss::SsArgs ka;
    ka.sorted = sorted;
    ka.n = n;
    ka.stride = z.stride;
    ka.alpha = alpha;
    ka.scratch = reinterpret_cast<double *>(base + z.o_scratch);
    ka.stat = d_stat;
    ka.only = z.tiled ? reinterpret_cast<const unsigned *>(base + z.o_flag) : nullptr;
    ss2::Args ta;
    ta.sorted = sorted;
    ta.n = n;
    ta.seg = z.seg;
    ta.seg_len = z.seg_len;
    ta.alpha = alpha;
    ta.sm = reinterpret_cast<double *>(base + z.o_sm);
    ta.arec = reinterpret_cast<ss2::rec4_t *>(base + z.o_arec);
    ta.srec = reinterpret_cast<fast::rec_t *>(base + z.o_srec);
    ta.mrec = reinterpret_cast<fast::rec_t *>(base + z.o_arec);    // (over arec: dead after the second sweep)
    ta.flag = reinterpret_cast<unsigned *>(base + z.o_flag);
    ta.part = reinterpret_cast<double *>(base + z.o_part);
    ta.plane = (int64_t)z.sb * n;
    ta.stat = d_stat;
    for (int64_t p0 = 0; p0 < n_periods; p0 += z.batch) {
        const int64_t bc = n_periods - p0 < z.batch ? n_periods - p0 : z.batch;
        fa.p0 = p0;
        fa.batch = (int)bc;
        if (z.fastsort) {
            // one workgroup per period sorts it in LDS (the StringLength kernels' EMIT instances) and writes the sorted
            // curve; the periods that outlast the samples are marked and written as they stand
            PDC_TRY(fast_sort_batch(st, d_t, d_y, n, d_periods + p0, bc, z.batch, bad, sorted, base + z.o_fast, &fa.skip8));
            hipLaunchKernelGGL(ss::ss_direct_kernel, dim3((unsigned)(bc * 8)), dim3(kBlock), 0, st, fa);
        } else {
            if (z.streamed) PDC_TRY(stream_sort_batch(device, st, z.h, sa, p0, bc));
            hipLaunchKernelGGL(ss::ss_sort_fallback_kernel, dim3((unsigned)(bc < z.grid_fb ? bc : z.grid_fb)), dim3(kBlock), 0,
                               st, fa);
            if (z.streamed && sa.direct) hipLaunchKernelGGL(ss::ss_direct_kernel, dim3((unsigned)(bc * 8)), dim3(kBlock), 0, st, fa);
        }
        if (z.tiled) {
            PDC_HIP(hipMemsetAsync(ta.flag, 0, (size_t)bc * 4, st));
            ta.p0 = p0;
            for (int q0 = 0; q0 < (int)bc; q0 += z.sb) {
                ta.q0 = q0;
                ta.nq = (int)bc - q0 < z.sb ? (int)bc - q0 : z.sb;
                const unsigned grid = (unsigned)((ta.nq + 7) / 8 * z.seg * 8);
                hipLaunchKernelGGL(ss2::ss2_stage_kernel<1>, dim3(grid), dim3(ss2::kTh), 0, st, ta);
                hipLaunchKernelGGL(ss2::ss2_stage_kernel<2>, dim3(grid), dim3(ss2::kTh), 0, st, ta);
                // (the last two sweeps keep up to four workgroups per CU: they may take more segments per period)
                ss2::Args tb = ta;
                tb.seg = z.seg34;
                tb.seg_len = z.seg_len34;
                const unsigned grid34 = (unsigned)((ta.nq + 7) / 8 * z.seg34 * 8);
                hipLaunchKernelGGL(ss2::ss2_stage_kernel<3>, dim3(grid34), dim3(ss2::kTh), 0, st, tb);
                hipLaunchKernelGGL(ss2::ss2_stage_kernel<4>, dim3(grid34), dim3(ss2::kTh), 0, st, tb);
                hipLaunchKernelGGL(ss2::ss2_finish_kernel, dim3((unsigned)((ta.nq + 63) / 64)), dim3(64), 0, st, tb);
            }
        }
#ifdef PDC_SS_DBG   // (developer builds: s_memrealtime stamps of one tile of sweep PDC_SS_DBG, 10 ns ticks)
        if (z.tiled && p0 == 0) {
            std::vector<double> h((size_t)bc * ss2::kSegMax);
            PDC_HIP(hipStreamSynchronize(st));
            PDC_HIP(hipMemcpy(h.data(), ta.part, h.size() * 8, hipMemcpyDeviceToHost));
            for (int q : {0, 9, 63, 64, 100}) {
                if (q >= bc) continue;
                const double *r = &h[(size_t)q * ss2::kSegMax];
                fprintf(stderr, "dbg sweep %d period %d seg 1 tile 2: loads landed %.0f scans %.0f barrier %.0f fits %.0f stores %.0f\n", PDC_SS_DBG, q,
                        r[21] - r[20], r[22] - r[21], r[23] - r[22], r[24] - r[23], r[25] - r[24]);
            }
        }
#endif
        ka.p0 = p0;
        ka.batch = (int)bc;
        hipLaunchKernelGGL(ss::ss_smooth_kernel, dim3((unsigned)(bc < z.grid_ss ? bc : z.grid_ss)), dim3(ss::kB), 0, st, ka);
        PDC_HIP(hipGetLastError());
    }
    return PDC_OK;
}
}  // namespace

extern "C" {

int pdc_supersmoother_scan(const double *t, const double *y, int64_t n, const double *periods, int64_t n_periods,
                           double alpha, double *stat_out, int device) {
    PDC_REQUIRE(t && y && (periods || n_periods == 0) && (stat_out || n_periods == 0), "supersmoother: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "supersmoother: negative size");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    // (time-ordered samples and periods that all take the slices mode: no lists in the workspace, as pdc_stringlength_scan)
    const int hints = host_hints(t, n, periods, n_periods, 4096);
    // (the workspace fitted to PDC_WORK_BUDGET_GB and to what the device has free right now)
    WorkScale ws(host_work_budget(device), [&] { return ss_shape(n, n_periods, (hints & kHintLists) != 0).total; });
    PDC_REQUIRE_FITS(ws, "supersmoother");
    const int64_t wb = ws.need;
    void *d_t, *d_y, *d_p, *d_s, *d_w;
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_y));
    PDC_TRY(cached(device, SLOT_IN2, n_periods * 8, &d_p));
    PDC_TRY(cached(device, SLOT_OUT0, n_periods * 8, &d_s));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_w));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_y, y, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_p, periods, n_periods * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(supersmoother_scan_impl(device, st, (double *)d_t, (double *)d_y, n, (double *)d_p, n_periods, alpha,
                                    (double *)d_s, d_w, wb, hints));
    PDC_HIP(hipMemcpyAsync(stat_out, d_s, n_periods * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

}  // extern "C"

// The two scans that sort every period by phase, for callers that hold the HOST arrays too (the phase plan of
// multi.hip): when the host sees that every period will take the slices / one-cycle modes of the streamed kernels the
// workspace needs no bin lists (~12 GB at a million samples) - what pdc_stringlength_scan / pdc_supersmoother_scan
// do for themselves.  kind 3 = StringLength, 5 = Supersmoother (alpha).
namespace pdc {
int sorted_scan_hints(int kind, const double *t, int64_t n, const double *periods, int64_t n_periods) {
    return host_hints(t, n, periods, n_periods, kind == 5 ? (int64_t)4096 : stream_min_n());
}
int64_t sorted_scan_work_bytes(int kind, int64_t n, int64_t n_periods, int hints) {
    const bool lists = (hints & kHintLists) != 0;
    if (n < 0 || n_periods < 0) return -1;
    WorkScale ws(work_budget(), [&] { return kind == 5 ? ss_shape(n, n_periods, lists).total : stringlength_work_bytes(n, n_periods, lists); });
    return ws.need;   // (may still exceed the budget: the scan itself then says so - PDC_REQUIRE_FITS)
}
int sorted_scan_dev(int kind, int device, void *stream, const double *d_t, const double *d_v, int64_t n, const double *d_periods,
                    int64_t n_periods, double alpha, double *d_out, void *work, int64_t work_bytes, int hints) {
    if (kind == 5) return supersmoother_scan_impl(device, stream, d_t, d_v, n, d_periods, n_periods, alpha, d_out, work, work_bytes, hints);
    return stringlength_scan_impl(device, stream, d_t, d_v, n, d_periods, n_periods, d_out, work, work_bytes, hints);
}
}  // namespace pdc

