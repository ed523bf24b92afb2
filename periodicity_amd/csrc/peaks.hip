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

// `segs` workgroups per row, each the maxima whose rising edge lies in its share of the bins (a flat top is read on
// across the share's end, so every maximum belongs to exactly one share); segs > 1 leaves (value, bin) per share in
// part_v / part_i for highest_peak_merge_kernel.  (A single workgroup reads a row at 5 GB/s: 16 ms for the 1e7 bins of
// one C4 spectrum.)
__global__ __launch_bounds__(kBlock) void highest_peak_kernel(const double *power, int64_t nf, int segs,
                                                              int64_t *idx_out, double *val_out,
                                                              double *part_v, long long *part_i) {
    __shared__ double red_v[kBlock / 64];
    __shared__ long long red_i[kBlock / 64];
    const int64_t row = blockIdx.x / (unsigned)segs, seg = blockIdx.x % (unsigned)segs;
    const double *x = power + row * nf;
    const int64_t share = (nf + segs - 1) / segs;
    const int64_t lo = seg * share > 1 ? seg * share : 1, hi = (seg + 1) * share < nf - 1 ? (seg + 1) * share : nf - 1;
    double best = 0.0;
    long long best_i = -1;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {  // ascending per thread
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
        if (segs > 1) {
            part_v[blockIdx.x] = best;
            part_i[blockIdx.x] = best_i;
        } else {
            if (idx_out) idx_out[row] = best_i;
            if (val_out) val_out[row] = best_i >= 0 ? best : __builtin_nan("");
        }
    }
}

// one wave per row: the shares' winners, the higher value first, the lower bin on equal values
__global__ __launch_bounds__(64) void highest_peak_merge_kernel(const double *part_v, const long long *part_i, int segs,
                                                                int64_t *idx_out, double *val_out) {
    const int64_t row = blockIdx.x;
    double best = 0.0;
    long long best_i = -1;
    for (int s = threadIdx.x; s < segs; s += 64) {
        const double ov = part_v[row * segs + s];
        const long long oi = part_i[row * segs + s];
        if (oi >= 0 && (best_i < 0 || ov > best || (ov == best && oi < best_i))) {
            best = ov;
            best_i = oi;
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
    if (threadIdx.x == 0) {
        if (idx_out) idx_out[row] = best_i;
        if (val_out) val_out[row] = best_i >= 0 ? best : __builtin_nan("");
    }
}

// ---- top-k peaks with prominences and half-maximum crossings ------------------------------------------
// For every spectrum of a batch resident in HBM: all find_peaks() maxima (core.py:283-317) with their
// scipy prominences (scipy.signal.peak_prominences, wlen=None: walk left and right from the peak while
// x[i] <= x[peak], remember the lowest sample on each side, prominence = x[peak] - max of the two),
// ranked by height (psort_by_peak, core.py:944-946) or by prominence (psort_by_prominence :948-950,
// period_at_highest_prominence :957-961); only the first k <= 16 come back, together with the two
// sign changes of x - (x[peak] - height/2) that periods_at_half_max (:963-978) looks up: the last
// one left of the peak and the first one from the peak rightwards (np.diff(np.signbit(..)), core.py:362).
// Equal heights / prominences rank the lower bin first - a documented deviation: upstream's
// argsort()[::-1] (core.py:944-950, 969) is an unstable sort read backwards, which for short arrays puts the
// HIGHER bin first; the host-side FSeries methods of this package run numpy as upstream does.
//
// One workgroup per spectrum.  Only k peaks are wanted, and walking every maximum (6300 per 5e4-bin C3
// row; what the first version did, 7.9 ms for the C3 batch, 85 % of it in wave-iterations as long as
// their longest walk) is not needed to rank them:
//   * prominence <= height - (lowest sample of the row): a maximum with  height - lowest < tau  cannot
//     be among the k most prominent once k prominences >= tau are known;
//   * ranked by height, only the k winners need a prominence at all.
// So the row is swept once through LDS tiles (maxima found and counted; a candidate list in LDS keeps the
// M highest seen so far: M = k by height, 32 by prominence), the candidates are walked, and - by
// prominence - a second sweep collects the few maxima with  height - lowest >= tau  (tiles whose block
// maxima rule that out are not even staged).  Per-block minima and maxima in LDS let a walk hop over
// every block that cannot stop it (a block holding a NaN never hops: a NaN stops a walk, as
// x[i] <= x[peak] is false).  Ranking is by successive "best entry strictly after the previous winner" in
// the total order (key descending, bin ascending), so a maximum collected twice is reported once.
constexpr int kPkBlock = 256;
#ifndef PDC_PK_WAVES
#define PDC_PK_WAVES 5
#endif
constexpr int kPkMaxBlocks = 4096;   // LDS: two doubles per block
constexpr int kPkMaxK = 128;         // ranked peaks per launch (round 6: 128 - half the launches, and sweeps, of a k > 64 call; round 5: 64)
constexpr int kPkMaxKTotal = 1024;
#ifndef PDC_PK_SORT_FROM
#define PDC_PK_SORT_FROM 16
#endif
constexpr int kPkSortFrom = PDC_PK_SORT_FROM;      // rankings of more than this many entries sort the candidate list instead of m reduction rounds   // ... per call: chunks of 64, each launch ranking what comes AFTER the chunk before
constexpr int kPkPre = 132;          // by prominence: the first walks go to the k + 4 highest maxima (first chunk), to 2 k + 4
                                     // seeds in the later chunks of a k > 64 call (round 6: a tighter threshold for the second sweep)
#ifndef PDC_PK_WALK_LOADS
#define PDC_PK_WALK_LOADS 8
#endif
constexpr int kPkWalkLoads = PDC_PK_WALK_LOADS;   // loads a lane of a walk has in flight: 16 lanes x 8 = 128 bins per memory latency
                                                  // (round 5, with 32-bit bin numbers inside a walk: 4 / 8 / 16 loads = 70 / 77 / 95
                                                  // registers; with 64-bit ones 8 cost the sixth workgroup per CU)
#ifndef PDC_PK_AHEAD
#define PDC_PK_AHEAD 2               // chunks in flight ahead of the one examined (first sweep).  Round 5: 2 - with the walks'
                                     // 32-bit bin numbers it fits the 80 registers of six workgroups per CU (round 3: it cost the
                                     // sixth and measured the same); same box k = 1 / 4 / 8 by height 0.362 / 0.438 / 0.473 ->
                                     // 0.340 / 0.428 / 0.455 ms, by prominence 0.532 / 0.592 / 0.703 -> 0.50 / 0.59 / 0.693
#endif
constexpr int kPkChunk = 1024;       // bins per sweep step
constexpr int kPkFusedShift = 8;     // 256-bin blocks = one wave's stretch of a chunk: extrema come with the sweep
static_assert(kPkChunk / kPkBlock * 64 == 1 << kPkFusedShift, "a wave's stretch is one block");
constexpr int kPkCap = 1024;         // candidate slots in LDS (24 B each): a chunk adds at most kPkChunk / 2

struct PeakArgs {
    const double *power;
    int64_t nf;
    int k, by_prominence, blk_shift;
    int k_total, k_off;   // the outputs are [rows][k_total]; this launch fills columns k_off .. k_off + k - 1 and ranks
                          // only what comes after column k_off - 1 in the total order (key descending, bin ascending)
    int64_t nblk, tile;
    long long *count, *idx, *half_lo, *half_hi;
    double *height, *prom;
};

// minimum over the wave, valid in lane 63: row shifts and row broadcasts on both halves of the double (DPP),
// no LDS traffic (lanes without a source keep their own value)
__device__ __forceinline__ double wave_min_to_lane63(double v) {
#define PDC_PK_STEP(ctrl)                                                                                    \
    {                                                                                                        \
        const long long b = __double_as_longlong(v);                                                         \
        const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, ctrl, 0xf, 0xf, false);                   \
        const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), ctrl, 0xf, 0xf, false);   \
        const double o = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);                         \
        v = o < v ? o : v;                                                                                   \
    }
    PDC_PK_STEP(0x111) PDC_PK_STEP(0x112) PDC_PK_STEP(0x114) PDC_PK_STEP(0x118) PDC_PK_STEP(0x142) PDC_PK_STEP(0x143)
#undef PDC_PK_STEP
    return v;
}

__device__ __forceinline__ int wave_min_int_to_lane63(int v) {
#define PDC_PK_STEP(ctrl)                                                            \
    {                                                                                \
        const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false);      \
        v = o < v ? o : v;                                                           \
    }
    PDC_PK_STEP(0x111) PDC_PK_STEP(0x112) PDC_PK_STEP(0x114) PDC_PK_STEP(0x118) PDC_PK_STEP(0x142) PDC_PK_STEP(0x143)
#undef PDC_PK_STEP
    return v;
}

__device__ __forceinline__ bool cand_before(double ka, long long ia, double kb, long long ib) {
    return ib < 0 || (ia >= 0 && (ka > kb || (ka == kb && ia < ib)));
}

// PDC_PK_DBG (developer builds, tools/peaks_stamps.py): s_memrealtime stamps (10 ns ticks) of every row's phases
#ifdef PDC_PK_DBG
__device__ unsigned long long pk_dbg[8192 * 16];
#define PK_STAMP(i)                                                                                   \
    if (tid == 0 && blockIdx.x < 8192) pk_dbg[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define PK_STAMP(i)
#endif

// SORT: the instance for launches of more than kPkSortFrom ranks (its rankings sort the candidate list; four more
// registers, five workgroups per CU instead of six - which cost the few-ranks launches 10 %, so they keep their own)
template <bool SORT>
__global__ __launch_bounds__(kPkBlock, SORT ? PDC_PK_WAVES : PDC_PK_WAVES + 1) void peaks_topk_kernel(PeakArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *bmin = reinterpret_cast<double *>(lds_raw);   // [nblk]
    double *bmax = bmin + a.nblk;                         // [nblk]
    double *ch = bmax + a.nblk;                           // [kPkCap] candidates: height,
    double *cp = ch + kPkCap;                             //          prominence (NaN: not walked yet),
    int *ci = reinterpret_cast<int *>(cp + kPkCap);               //  bin (rows have < 2^31 bins: launch_topk)
    __shared__ int s_ncand, s_first;
    __shared__ double s_thr;
    __shared__ double red_k[kPkBlock / 64];
    __shared__ int red_e[kPkBlock / 64];
    __shared__ long long s_count[kPkBlock / 64];
    constexpr int kWin = SORT ? kPkPre : kPkSortFrom + 4;     // winners a ranking of this instance can be asked for
    __shared__ double win_key[kWin], win_h[kWin], win_p[kWin];
    __shared__ long long win_idx[kWin];
    constexpr int kTop = kPkSortFrom + 4;                     // few-ranks rankings (m <= kPkSortFrom): every wave's winners
    __shared__ int wtop[kPkBlock / 64 * kTop];
    __shared__ double s_low[2][8][2];                         // the walks of a pass: lowest sample met leftwards / rightwards               // few ranks: every wave's winners (slots of the candidate list)
    __shared__ unsigned s_need[32];   // second sweep: chunks that can hold a candidate at the initial tau
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *x = a.power + (int64_t)blockIdx.x * a.nf;
    const int64_t nf = a.nf;
    const int sh = a.blk_shift;
    const int64_t blk = (int64_t)1 << sh;
    const double inf = __builtin_inf();

    // ---- A: per-block minimum / maximum (a wave per block, coalesced); with 256-bin blocks (rows up to
    // 1M bins) the first sweep below does this on its way
    for (int64_t b = wave; b < a.nblk && sh != kPkFusedShift; b += kPkBlock / 64) {
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
    if (tid == 0) {
        s_ncand = 0;
        s_thr = -inf;
    }
    __syncthreads();
    // Lowest sample met walking from bin `from` in direction dir (+1 / -1) while x[i] <= h (scipy's loop).
    // Sixteen lanes walk together (arguments uniform over the group): 64 bins of the first block per step
    // (four loads per lane in flight), then 16 blocks per step over the block extrema in LDS - whole
    // blocks that cannot stop the walk are hopped over -, then the block that does stop it.  A step
    // costs one memory latency instead of one per bin: the highest peak of a 5e4-bin row (900 single
    // steps) takes ~12.
    const int gl = lane & 15, gbase = lane & 48;
    auto group_min = [&](double m) __attribute__((always_inline)) -> double {
        for (int o = 8; o > 0; o >>= 1) {
            const double om = __shfl_xor(m, o, 64);
            m = om < m ? om : m;
        }
        return m;
    };
    auto group_ballot = [&](bool p) __attribute__((always_inline)) -> unsigned {
        return (unsigned)((__ballot(p) >> gbase) & 0xffffull);
    };
    // (32-bit bin numbers inside a walk - rows have < 2^31 bins -: the row pointer stays a scalar base and a load takes
    // one address register)
    auto walk = [&](int from, int dir, double h) __attribute__((always_inline)) -> double {
        double low = h;
        const int nfi = (int)nf, nblki = (int)a.nblk;
        // `count` bins from `start` on in direction dir; true when the walk ended among them
        auto run = [&](int start, int count) __attribute__((always_inline)) -> bool {
            for (int o = 0; o < count; o += 16 * kPkWalkLoads) {
                double v[kPkWalkLoads];
#pragma unroll
                for (int u = 0; u < kPkWalkLoads; ++u) {
                    const int k = o + 16 * u + gl;
                    v[u] = k < count ? x[(unsigned)(start + dir * k)] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < kPkWalkLoads; ++u) {
                    const bool valid = o + 16 * u + gl < count;
                    const unsigned sb = group_ballot(valid && !(v[u] <= h));
                    const int first = sb ? __builtin_ctz(sb) : 16;
                    const double m = group_min(valid && gl < first ? v[u] : inf);
                    low = m < low ? m : low;
                    if (sb) return true;
                }
            }
            return false;
        };
        const int b0 = from >> sh;
        const int b0_end = ((b0 + 1) << sh) < nfi ? ((b0 + 1) << sh) : nfi;
        if (run(from, dir > 0 ? b0_end - from : from - (b0 << sh) + 1)) return low;
        int b = b0 + dir;
        for (;;) {
            const int bb = b + dir * gl;
            const bool pass = bb >= 0 && bb < nblki && bmax[bb] <= h;
            const unsigned fb = group_ballot(!pass);
            const int first = fb ? __builtin_ctz(fb) : 16;
            const double m = group_min(gl < first ? bmin[bb] : inf);   // (the lanes below `first` pass: in range)
            low = m < low ? m : low;
            if (fb) {
                b += dir * first;
                break;
            }
            b += 16 * dir;
        }
        if (b < 0 || b >= nblki) return low;   // reached the border of the signal
        const int b_end = ((b + 1) << sh) < nfi ? ((b + 1) << sh) : nfi;
        run(dir > 0 ? b << sh : b_end - 1, b_end - (b << sh));   // this block holds a sample > h (or a NaN)
        return low;
    };
    // prominences of the candidates that have none yet, eight candidates at a time: the groups of waves 0 and 1 walk
    // left, those of waves 2 and 3 right (round 5: one group walked left, then right - the walks are chains of dependent
    // loads, and half of the sixteen groups had nothing to do for the eight candidates of a k = 4 call).  The direction
    // is the wave's, so both walks keep a compile-time stride (a stride per group cost 14 registers = the sixth
    // workgroup per CU).
    auto walk_candidates = [&]() __attribute__((always_inline)) {
        const int n = s_ncand;
        // the entries without a prominence start at t0 (a ranking leaves its winners - walked - in front, a sweep appends)
        if (tid == 0) s_first = n;
        __syncthreads();
        for (int e = tid; e < n; e += kPkBlock)
            if (!(cp[e] == cp[e])) atomicMin(&s_first, e);
        __syncthreads();
        const int t0 = s_first;
        if (n - t0 > 16) {
            // many (a later launch of a k > 64 call, a second sweep that found a lot): one group per candidate, left then
            // right, no barrier between candidates
            for (int e = t0 + (tid >> 4); e < n; e += kPkBlock / 16) {
                if (cp[e] == cp[e]) continue;
                const double v = ch[e];
                const int at = ci[e];
                // (round 6: both walks in lockstep - the loads of the left and the right step requested together - measured
                // 6.61 against 6.68 ms for k = 256 by prominence: the walks are bound by how many are in flight, not by a
                // candidate's own chain of latencies; withdrawn)
                const double lo = walk(at, -1, v), hi = walk(at, +1, v);
                if (gl == 0) cp[e] = v - (lo > hi ? lo : hi);
            }
            __syncthreads();
            return;
        }
        const int slot = (tid >> 4) & 7;
        const bool right = __builtin_amdgcn_readfirstlane(wave) >= 2;
        int par = 0;
        for (int base = t0; base < n; base += 8, par ^= 1) {
            const int e = base + slot;
            if (e < n && !(cp[e] == cp[e])) {
                const double v = ch[e];
                const int at = ci[e];
                const double low = right ? walk(at, +1, v) : walk(at, -1, v);
                if (gl == 0) s_low[par][slot][right ? 1 : 0] = low;
            }
            __syncthreads();
            if (tid < 8 && base + tid < n && !(cp[base + tid] == cp[base + tid])) {
                const double lo = s_low[par][tid][0], hi = s_low[par][tid][1];
                cp[base + tid] = ch[base + tid] - (lo > hi ? lo : hi);
            }
            // (the next pass writes the other half of s_low and other candidates; its barrier orders these reads before
            // the pass after it)
        }
        __syncthreads();
    };
    const bool excl = SORT && a.k_off > 0;   // (later chunks of a k > 64 call always run the SORT instance: launch_topk)
    double prev_key = inf;
    long long prev_idx = -1;
    double h_cut = inf;   // (later chunks by prominence) lowest height among the winners already output
    // The m best candidates (by height or by prominence; ties: lower bin first; the same bin only once)
    // into win_*[0 .. m), by m rounds of "best entry strictly after the previous winner"; the list is then
    // cut down to those.  Returns how many there are.
    auto rank_candidates = [&](int m, bool by_prom) __attribute__((always_inline)) -> int {
        const int n = s_ncand;
        double pk = inf;          // previous winner (key, bin): everything is "after" (+inf, -1)
        long long pidx = -1;
        // a later chunk of a call with k > 64: rankings by the call's own key start after the chunk before's last winner
        bool ex = excl && by_prom == (a.by_prominence != 0);
        if (ex) {
            pk = prev_key;
            pidx = prev_idx;
        } else if (excl && a.by_prominence) {
            // ... and (round 6) the SEEDS of a later chunk by prominence - the rankings by height that pick the maxima
            // walked first, whose K-th prominence becomes the threshold tau of the second sweep - are the highest
            // maxima BELOW the lowest height any earlier chunk has output: none of them is an earlier winner, so all
            // of them count towards tau.  (Up to round 5 the seeds were the `pre` highest maxima of the row - mostly the
            // winners already output, dropped again by the exclusion above - so fewer than K were left, tau stayed
            // -inf and the second sweep collected and walked EVERY maximum of the row: 5.5 ms per 64 ranks on the C3
            // batch against 0.5 for the first 64.)  Any set of seeds gives a valid tau; this one gives a useful one.
            ex = true;
            pk = h_cut;
            pidx = 0x7fffffff;   // (strictly below h_cut)
        }
        int found = 0;
        if (SORT && m > kPkSortFrom) {
            // Many ranks (a launch of a k > 16 call): the rounds below cost a block reduction each - 68 of them ~0.3 ms per
            // ranking.  Instead the whole list is sorted in LDS (bitonic, by key descending then bin ascending; unused
            // slots last), duplicates of a bin and entries not after the previous chunk's last winner are dropped,
            // and the first m that remain are the winners: ~55 compare-exchange steps for a full list.
            int P = 2;
            while (P < n) P <<= 1;
            for (int e = n + tid; e < P; e += kPkBlock) {
                ci[e] = -1;
                ch[e] = -inf;
                cp[e] = -inf;
            }
            __syncthreads();
            for (int kk = 2; kk <= P; kk <<= 1) {
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    for (int c = tid; c < (P >> 1); c += kPkBlock) {
                        const int i = ((c & ~(j - 1)) << 1) | (c & (j - 1)), l = i | j;
                        const double ka = by_prom ? cp[i] : ch[i], kb = by_prom ? cp[l] : ch[l];
                        const int ia = ci[i], ib = ci[l];
                        // (a NaN key - none is expected here - sorts with the unused slots' -inf)
                        const bool b_first = cand_before(kb == kb ? kb : -inf, ib, ka == ka ? ka : -inf, ia);
                        const bool up = (i & kk) == 0;
                        if (b_first == up && !(ia == ib && ka == kb)) {
                            const double ha = ch[i], pa = cp[i];
                            ch[i] = ch[l];
                            cp[i] = cp[l];
                            ci[i] = ib;
                            ch[l] = ha;
                            cp[l] = pa;
                            ci[l] = ia;
                        }
                    }
                    __syncthreads();
                }
            }
            int taken = 0;                                       // winners so far (workgroup-uniform)
            for (int base = 0; base < n && taken < m; base += kPkBlock) {
                const int e = base + tid;
                bool ok = false;
                double key = 0.0;
                long long bin = -1;
                if (e < n) {
                    key = by_prom ? cp[e] : ch[e];
                    bin = ci[e];
                    ok = bin >= 0 && (e == 0 || ci[e - 1] != (int)bin) && (!ex || key < pk || (key == pk && bin > pidx));
                }
                const unsigned long long mask = __ballot(ok);
                if (lane == 0) red_e[wave] = __builtin_popcountll(mask);
                __syncthreads();
                int rank = taken + __builtin_popcountll(mask & ((1ull << lane) - 1ull)), total = 0;
                for (int w = 0; w < kPkBlock / 64; ++w) {
                    if (w < wave) rank += red_e[w];
                    total += red_e[w];
                }
                if (ok && rank < m) {
                    win_key[rank] = key;
                    win_idx[rank] = bin;
                    win_h[rank] = ch[e];
                    win_p[rank] = cp[e];
                }
                taken += total;
                __syncthreads();
            }
            found = taken < m ? taken : m;
        } else {
            // Few ranks.  Round 5: two phases without a barrier inside - (1) every wave ranks the m best of ITS share of
            // the list (entry e belongs to wave (e / 64) mod 4), m rounds of "best entry strictly after the wave's previous
            // winner", the winners' slots to wtop[wave][]; (2) after one barrier every wave ranks the <= 4 m slots of
            // wtop the same way - all four get the same answer, so none waits for another.  A round is a maximum of the
            // keys over the wave by DPP, the lowest bin among the lanes that hold it, and a ballot: no LDS crossbar, no
            // barrier (round 4: 3 block reductions with 2 barriers per round - 11.7 us for the 8 rounds of a k = 4 call
            // by prominence, three rankings per row).
            // best (key descending, bin ascending) over the wave of every lane's (key, bin >= 0 or -1: none, slot)
            auto wave_best = [&](double key, int bin, int slot, double &bk, int &bi, int &be) __attribute__((always_inline)) -> bool {
                const double kk = bin >= 0 ? key : -inf;
                const long long tb = __double_as_longlong(wave_min_to_lane63(-kk));
                const double top = -__longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(tb >> 32), 63) << 32) |
                                                         (unsigned)__builtin_amdgcn_readlane((int)tb, 63));
                const bool tied = bin >= 0 && kk == top;
                const int lowest = __builtin_amdgcn_readlane(wave_min_int_to_lane63(tied ? bin : 0x7fffffff), 63);
                const unsigned long long who = __ballot(tied && bin == lowest);
                if (who == 0ull) return false;
                const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(who));
                // (the winner's own bits: a tie of -0.0 and +0.0 keys is a tie, but the half-maximum level downstream takes
                // the sign along - tools/fuzz_peaks.py case 23)
                const long long kb = __double_as_longlong(key);
                bk = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(kb >> 32), src) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)kb, src));
                bi = lowest;
                be = __builtin_amdgcn_readlane(slot, src);
                return true;
            };
            {   // phase 1
                double wpk = pk;
                int wpi = (int)pidx;
                bool any_prev = ex;
                int r = 0;
                for (; r < m; ++r) {
                    double lk = 0.0;
                    int li = -1, le = -1;
                    for (int e = tid; e < n; e += kPkBlock) {
                        const double key = by_prom ? cp[e] : ch[e];
                        const int bin = ci[e];
                        const bool after = !any_prev || key < wpk || (key == wpk && bin > wpi);
                        if (after && cand_before(key, bin, lk, li)) {
                            lk = key;
                            li = bin;
                            le = e;
                        }
                    }
                    double bk;
                    int bi, be;
                    if (!wave_best(lk, li, le, bk, bi, be)) break;    // (wave-uniform)
                    if (lane == 0) wtop[wave * kTop + r] = be;
                    wpk = bk;
                    wpi = bi;
                    any_prev = true;
                }
                for (int q = r + lane; q < kTop; q += 64) wtop[wave * kTop + q] = -1;
            }
            __syncthreads();
            bool any_prev = ex;
            for (int round = 0; round < m; ++round) {
                double lk = 0.0;
                int li = -1, le = -1;
                for (int q = lane; q < (kPkBlock / 64) * kTop; q += 64) {
                    const int e = wtop[q];
                    if (e < 0) continue;
                    const double key = by_prom ? cp[e] : ch[e];
                    const int bin = ci[e];
                    const bool after = !any_prev || key < pk || (key == pk && bin > pidx);
                    if (after && cand_before(key, bin, lk, li)) {
                        lk = key;
                        li = bin;
                        le = e;
                    }
                }
                double bk;
                int bi, be;
                if (!wave_best(lk, li, le, bk, bi, be)) break;        // (the same in every wave)
                if (tid == 0) {
                    win_key[round] = bk;
                    win_idx[round] = bi;
                    win_h[round] = ch[be];
                    win_p[round] = cp[be];
                }
                pk = bk;
                pidx = bi;
                any_prev = true;
                ++found;
            }
        }
        __syncthreads();
        for (int e = tid; e < found; e += kPkBlock) {
            ci[e] = (int)win_idx[e];
            ch[e] = win_h[e];
            cp[e] = win_p[e];
        }
        if (tid == 0) s_ncand = found;
        __syncthreads();
        return found;
    };
    // One sweep over the row in chunks of 1024 bins, lanes = consecutive bins, straight from global memory
    // (neighbours by lane shuffles).  FIRST: every maximum is counted; those with height >= s_thr are
    // collected, and whenever the list is half full it is cut down to the m highest (s_thr = the m-th).
    // Otherwise (by prominence, second sweep): the maxima with height - row_min >= s_thr are collected
    // (s_thr = tau), walked and ranked when the list fills up (tau rises to the k-th prominence); chunks
    // whose block maxima rule out a candidate are not read.
    long long mine = 0;
    double row_min = inf;
    auto sweep = [&](auto first_tag, int m) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const double sub = FIRST ? 0.0 : row_min;
        const double nan = __builtin_nan("");
        // a wave reads a stretch of 256 consecutive bins, 64 at a time, + the bin before and after:
        // every neighbour is a lane shuffle away, and all loads of a chunk are in flight together - in the
        // first sweep one chunk ahead of the one being examined
        constexpr int kPer = kPkChunk / kPkBlock;
        // lane L owns the kPer CONSECUTIVE bins w0 + kPer L .. + kPer - 1 (two 16-byte loads on the fast path):
        // three of four neighbours are the lane's own registers, the other two one DPP wave shift away
        // (the first version gave lane L the bins w0 + 64 j + L and fetched every neighbour with
        // ds_bpermute: 55 LDS instructions per wave and chunk)
        static_assert(kPer == 4, "the fast path loads two pairs of doubles");
        typedef double pair_t __attribute__((ext_vector_type(2), aligned(8)));
        auto load = [&](int64_t c0, double (&v)[kPer], double &halo_lo, double &halo_hi) __attribute__((always_inline)) {
            const int64_t w0 = c0 + (int64_t)wave * (64 * kPer);
            const int64_t i0 = w0 + (int64_t)lane * kPer;
            if (w0 + 64 * kPer <= nf) {   // (wave-uniform) the whole stretch lies inside the row
                const pair_t a01 = *reinterpret_cast<const pair_t *>(x + i0);
                const pair_t a23 = *reinterpret_cast<const pair_t *>(x + i0 + 2);
                v[0] = a01.x;
                v[1] = a01.y;
                v[2] = a23.x;
                v[3] = a23.y;
            } else {
#pragma unroll
                for (int j = 0; j < kPer; ++j) v[j] = i0 + j < nf ? x[i0 + j] : nan;
            }
            halo_lo = (w0 >= 1 && w0 <= nf) ? x[w0 - 1] : nan;
            halo_hi = w0 + 64 * kPer < nf ? x[w0 + 64 * kPer] : nan;
        };
        // x of the lane below / above (lane 0 / 63: the halo), by DPP wave shifts on both halves of the double
        auto from_below = [&](double q, double edge) __attribute__((always_inline)) -> double {
            const long long qb = __double_as_longlong(q), eb = __double_as_longlong(edge);
            const int lo = __builtin_amdgcn_update_dpp((int)eb, (int)qb, 0x138, 0xf, 0xf, false);           // wave_shr:1
            const int hi = __builtin_amdgcn_update_dpp((int)(eb >> 32), (int)(qb >> 32), 0x138, 0xf, 0xf, false);
            return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        };
        auto from_above = [&](double q, double edge) __attribute__((always_inline)) -> double {
            const long long qb = __double_as_longlong(q), eb = __double_as_longlong(edge);
            const int lo = __builtin_amdgcn_update_dpp((int)eb, (int)qb, 0x130, 0xf, 0xf, false);           // wave_shl:1
            const int hi = __builtin_amdgcn_update_dpp((int)(eb >> 32), (int)(qb >> 32), 0x130, 0xf, 0xf, false);
            return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        };
        // (first sweep: one chunk ahead of the one being examined; two ahead measured the same and cost the
        // registers that decide between five and six workgroups per CU)
        double v[kPer], halo_lo, halo_hi, vn[kPer], next_lo = nan, next_hi = nan;
#if PDC_PK_AHEAD == 2
        double vnn[kPer], nn_lo = nan, nn_hi = nan;
#endif
        if (FIRST) load(0, v, halo_lo, halo_hi);
#if PDC_PK_AHEAD == 2
        if (FIRST) load(kPkChunk, vn, next_lo, next_hi);
#endif
        // Second sweep: tau only rises, so a chunk whose block maxima rule a candidate out at the INITIAL tau
        // never needs a look - found for all chunks at once, and those chunks cost neither a barrier nor an
        // LDS scan below (of 49 chunks of a C3 row a handful remain)
        const int64_t nchunks = (nf + kPkChunk - 1) / kPkChunk;
        const bool masked = !FIRST && nchunks <= 32 * 32;
        if (masked) {
            if (tid < 32) s_need[tid] = 0u;
            __syncthreads();
            const double thr0 = s_thr;
            for (int64_t c = tid; c < nchunks; c += kPkBlock) {
                const int64_t e0 = c * kPkChunk, e1 = e0 + kPkChunk < nf ? e0 + kPkChunk : nf;
                double tmx = -inf;
                for (int64_t b = e0 >> sh; b < ((e1 + blk - 1) >> sh); ++b) tmx = bmax[b] > tmx ? bmax[b] : tmx;
                if (tmx - sub >= thr0) atomicOr(&s_need[c >> 5], 1u << (c & 31));
            }
            __syncthreads();
        }
        for (int64_t c0 = 0; c0 < nf; c0 += kPkChunk) {
            const int64_t c1 = c0 + kPkChunk < nf ? c0 + kPkChunk : nf;
            if (masked && !((s_need[(c0 / kPkChunk) >> 5] >> ((c0 / kPkChunk) & 31)) & 1u)) continue;   // (workgroup-uniform)
#if PDC_PK_AHEAD == 2
            if (FIRST) load(c0 + 2 * kPkChunk, vnn, nn_lo, nn_hi);
#else
            if (FIRST) load(c0 + kPkChunk, vn, next_lo, next_hi);
#endif
            __syncthreads();   // everyone is done with the previous chunk (and sees s_thr / s_ncand)
            if (s_ncand > kPkCap / 2) {   // (workgroup-uniform) a chunk adds at most kPkChunk / 2 maxima
                if (!FIRST) walk_candidates();
                const int got = rank_candidates(m, !FIRST);
                if (tid == 0 && got == m) s_thr = win_key[m - 1];
                __syncthreads();
            }
            const double thr = s_thr;
            if (!FIRST) {   // (workgroup-uniform) no block of this chunk reaches the threshold
                double tmx = -inf;
                for (int64_t b = c0 >> sh; b < ((c1 + blk - 1) >> sh); ++b) tmx = bmax[b] > tmx ? bmax[b] : tmx;
                if (!(tmx - sub >= thr)) continue;
            }
            if (!FIRST) load(c0, v, halo_lo, halo_hi);
            const int64_t w0 = c0 + (int64_t)wave * (64 * kPer);
            if (FIRST && sh == kPkFusedShift && w0 < nf) {   // (wave-uniform) this wave's stretch is one block
                double mn = inf, mx = -inf;
                bool isnan = false;
#pragma unroll
                for (int j = 0; j < kPer; ++j) {
                    const double q = v[j];
                    isnan = isnan || (w0 + lane * kPer + j < nf && q != q);
                    mn = q < mn ? q : mn;
                    mx = q > mx ? q : mx;
                }
                mn = wave_min_to_lane63(mn);
                mx = -wave_min_to_lane63(-mx);
                if (__any(isnan)) mx = inf;
                if (lane == 63) {
                    bmin[w0 >> kPkFusedShift] = mn;
                    bmax[w0 >> kPkFusedShift] = mx;
                }
            }
            const double below = from_below(v[kPer - 1], halo_lo), above = from_above(v[0], halo_hi);
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int64_t i = w0 + (int64_t)lane * kPer + j;
                const double prev = j == 0 ? below : v[j > 0 ? j - 1 : 0];
                const double next = j == kPer - 1 ? above : v[j < kPer - 1 ? j + 1 : j];
                // a maximum: strict rise, flat tops -> midpoint, edges and NaN never peaks
                const double vv = v[j];
                if (!(i >= 1 && i < nf - 1 && prev < vv)) continue;
                if (!FIRST && !(vv - sub >= thr)) continue;
                int64_t ahead = i + 1;
                if (!(next < vv)) {
                    if (!(next == vv)) continue;
                    while (ahead < nf - 1 && x[ahead] == vv) ++ahead;   // a flat top: global reads from here on
                    if (!(x[ahead] < vv)) continue;
                }
                if (FIRST) ++mine;
                if (FIRST && !(vv >= thr)) continue;
                const int slot = atomicAdd(&s_ncand, 1);
                const int64_t mid = (i + ahead - 1) / 2;
                ci[slot] = (int)mid;
                ch[slot] = mid == i ? vv : x[mid];   // (a flat top of zeros may mix +0.0 and -0.0: the reported
                                                     // height is the midpoint's own bits, as scipy's x[peaks])
                cp[slot] = nan;
            }
            if (FIRST) {
#pragma unroll
                for (int j = 0; j < kPer; ++j) v[j] = vn[j];
                halo_lo = next_lo;
                halo_hi = next_hi;
#if PDC_PK_AHEAD == 2
#pragma unroll
                for (int j = 0; j < kPer; ++j) vn[j] = vnn[j];
                next_lo = nn_lo;
                next_hi = nn_hi;
#endif
            }
        }
        __syncthreads();
    };

    const int K = a.k < kPkMaxK ? a.k : kPkMaxK;
    // by prominence: the first walks go to the `pre` highest maxima - in a later chunk (seeds below h_cut, see
    // rank_candidates) to twice as many: tau is then the K-th of 2 K + 4 prominences instead of the K-th of K + 4
    const int pre = SORT && a.k_off > 0 && 2 * K + 4 <= kPkPre ? 2 * K + 4 : K + 4;
    const int64_t ob = (int64_t)blockIdx.x * a.k_total + a.k_off;
    if (excl) {
        prev_key = (a.by_prominence ? a.prom : a.height)[ob - 1];
        prev_idx = a.idx[ob - 1];
        if (a.by_prominence) {   // lowest height among the winners of the chunks before (their bins are in idx[])
            double lowest = inf;
            for (int j = tid; j < a.k_off; j += kPkBlock) {
                const long long b = a.idx[ob - a.k_off + j];
                if (b >= 0) {
                    const double hb = x[b];
                    lowest = hb < lowest ? hb : lowest;
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double om = __shfl_xor(lowest, o, 64);
                lowest = om < lowest ? om : lowest;
            }
            if (lane == 0) red_k[wave] = lowest;
            __syncthreads();
            for (int w = 0; w < kPkBlock / 64; ++w) lowest = red_k[w] < lowest ? red_k[w] : lowest;
            __syncthreads();
            h_cut = lowest;
        }
        if (prev_idx < 0) {   // (workgroup-uniform) the chunk before already ran out of peaks
            if (tid < a.k) {
                if (a.idx) a.idx[ob + tid] = -1;
                if (a.height) a.height[ob + tid] = __builtin_nan("");
                if (a.prom) a.prom[ob + tid] = __builtin_nan("");
                if (a.half_lo) a.half_lo[ob + tid] = -1;
                if (a.half_hi) a.half_hi[ob + tid] = -1;
            }
            return;
        }
    }
    static_assert(kPkMaxK + 4 <= kPkPre && kPkPre <= kPkCap / 4, "win_* and the candidate list hold the first walks");
    PK_STAMP(0)
    sweep(std::true_type{}, a.by_prominence ? pre : K);
    PK_STAMP(1)
    // lowest sample of the row (NaN aside): prominence <= height - row_min
    for (int64_t b = tid; b < a.nblk; b += kPkBlock) row_min = bmin[b] < row_min ? bmin[b] : row_min;
    for (int o = 32; o > 0; o >>= 1) {
        const double om = __shfl_xor(row_min, o, 64);
        row_min = om < row_min ? om : row_min;
    }
    if (lane == 0) red_k[wave] = row_min;
    __syncthreads();
    for (int w = 0; w < kPkBlock / 64; ++w) row_min = red_k[w] < row_min ? red_k[w] : row_min;
    __syncthreads();

    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if (lane == 0) s_count[wave] = mine;
    int nwin;
    if (!a.by_prominence) {
        nwin = rank_candidates(K, false);
        PK_STAMP(2)
        walk_candidates();                       // the winners' prominences
        PK_STAMP(3)
        if (tid < nwin) win_p[tid] = cp[tid];
    } else {
        rank_candidates(pre, false);
        PK_STAMP(2)
        walk_candidates();                       // the highest maxima
        PK_STAMP(3)
        nwin = rank_candidates(K, true);
        PK_STAMP(4)
        if (tid == 0) s_thr = nwin == K ? win_key[K - 1] : -inf;   // tau
        sweep(std::false_type{}, K);
        PK_STAMP(5)
        walk_candidates();
        PK_STAMP(6)
        nwin = rank_candidates(K, true);
        PK_STAMP(7)
    }
    __syncthreads();
    if (tid == 0) {
        long long total = 0;
        for (int w = 0; w < kPkBlock / 64; ++w) total += s_count[w];
        if (a.count) a.count[blockIdx.x] = total;
    }
    if (tid < a.k) {
        const bool ok = tid < nwin;
        if (a.idx) a.idx[ob + tid] = ok ? win_idx[tid] : -1;
        if (a.height) a.height[ob + tid] = ok ? win_h[tid] : __builtin_nan("");
        if (a.prom) a.prom[ob + tid] = ok ? win_p[tid] : __builtin_nan("");
    }

    // ---- D: half-maximum crossings of every ranked peak (periods_at_half_max) -------------------
    // One wave per ranked peak (the k searches run side by side, no workgroup barriers): 256 bins per step
    // outwards from the peak, four loads per lane in flight, the nearest sign change by ballot.
    PK_STAMP(8)
    if (!a.half_lo && !a.half_hi) return;
    if (SORT && a.k > 16) {
        // Many ranked peaks (round 6): a QUARTER wave per peak, four peaks per wave side by side, 64 bins per step.  The
        // crossings of a periodogram's peaks lie a few bins from the maximum, and one wave per peak spent a 256-bin step
        // each way on every one of them, 32 peaks per wave one after the other: 136 us of a row's 835 in a 128-rank
        // launch (profiles/r06_peaks_stamps.txt).  Same searches, same answers.
        const int sub = lane & 15, grp = lane >> 4;
        for (int r0 = wave * 4; r0 < a.k; r0 += (kPkBlock / 64) * 4) {
            const int r = r0 + grp;
            const long long idmax = (r < a.k && r < nwin) ? win_idx[r] : -1;
            long long lo_abs = -1, hi_abs = -1;
            const double half = idmax >= 0 ? x[idmax] - win_key[r] / 2 : 0.0;     // core.py:972 (height or prominence)
            auto flips = [&](int64_t i) {                       // signbit(x[i]-half) != signbit(x[i+1]-half)
                return (__double_as_longlong(x[i] - half) < 0) != (__double_as_longlong(x[i + 1] - half) < 0);
            };
            // last sign change inside x[:idmax], nearest to the peak first
            for (int64_t top = idmax - 2;; top -= 64) {
                const bool go = idmax >= 0 && hi_abs < 0 && top >= 0;
                if (!__any(go)) break;                          // (wave-uniform)
                bool f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = top - (16 * u + sub);
                    f[u] = go && i >= 0 && flips(i);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned mine = (unsigned)(__ballot(f[u]) >> (16 * grp)) & 0xFFFFu;
                    if (mine && hi_abs < 0) hi_abs = top - (16 * u + __builtin_ctz(mine));
                }
            }
            // first sign change from the peak rightwards
            for (int64_t base = idmax;; base += 64) {
                const bool go = idmax >= 0 && lo_abs < 0 && base < nf - 1;
                if (!__any(go)) break;                          // (wave-uniform)
                bool f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = base + 16 * u + sub;
                    f[u] = go && i < nf - 1 && flips(i);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned mine = (unsigned)(__ballot(f[u]) >> (16 * grp)) & 0xFFFFu;
                    if (mine && lo_abs < 0) lo_abs = base + 16 * u + __builtin_ctz(mine);
                }
            }
            if (sub == 0 && r < a.k) {
                if (a.half_lo) a.half_lo[ob + r] = lo_abs;
                if (a.half_hi) a.half_hi[ob + r] = hi_abs;
            }
        }
        PK_STAMP(9)
        return;
    }
    for (int r = wave; r < a.k; r += kPkBlock / 64) {
        const long long idmax = r < nwin ? win_idx[r] : -1;
        long long lo_abs = -1, hi_abs = -1;
        if (idmax >= 0) {
            const double half = x[idmax] - win_key[r] / 2;     // core.py:972 (height or prominence)
            auto flips = [&](int64_t i) {                       // signbit(x[i]-half) != signbit(x[i+1]-half)
                return (__double_as_longlong(x[i] - half) < 0) != (__double_as_longlong(x[i + 1] - half) < 0);
            };
            // last sign change inside x[:idmax]: pairs (i, i+1), i+1 <= idmax-1; nearest to the peak first
            for (int64_t top = idmax - 2; top >= 0 && hi_abs < 0; top -= 256) {
                bool f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = top - (64 * u + lane);
                    f[u] = i >= 0 && flips(i);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long mask = __ballot(f[u]);
                    if (mask && hi_abs < 0) hi_abs = top - (64 * u + __builtin_ctzll(mask));
                }
            }
            // first sign change from the peak rightwards: pairs (idmax+i, idmax+i+1)
            for (int64_t base = idmax; base < nf - 1 && lo_abs < 0; base += 256) {
                bool f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = base + 64 * u + lane;
                    f[u] = i < nf - 1 && flips(i);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long mask = __ballot(f[u]);
                    if (mask && lo_abs < 0) lo_abs = base + 64 * u + __builtin_ctzll(mask);
                }
            }
        }
        if (lane == 0) {
            if (a.half_lo) a.half_lo[ob + r] = lo_abs;
            if (a.half_hi) a.half_hi[ob + r] = hi_abs;
        }
    }
    PK_STAMP(9)
}

}  // namespace

#ifdef PDC_PK_DBG
extern "C" int pdc_debug_peaks_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pk_dbg), sizeof(pk_dbg)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" {

int pdc_highest_peak_dev(int device, void *stream, const double *d_power, int64_t n_curves,
                         int64_t nf, int64_t *d_idx, double *d_val) {
    PDC_REQUIRE(d_power || n_curves * nf == 0, "highest_peak: power is NULL");
    PDC_REQUIRE(d_idx || d_val, "highest_peak: no output requested");
    PDC_REQUIRE(n_curves >= 0 && nf >= 0 && n_curves < ((int64_t)1 << 31), "highest_peak: bad size");
    if (n_curves == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    hipStream_t st = (hipStream_t)stream;
    // few long rows: several workgroups per row (shares of >= 16384 bins, ~2048 workgroups in all) + a merge
    int64_t segs = 2048 / n_curves;
    segs = segs < nf / 16384 ? segs : nf / 16384;
    if (segs > 1) {
        void *sp = nullptr;
        PDC_TRY(stream_scratch(device, st, n_curves * segs * 16, &sp));
        ScratchPin pin;
        pin.device = device;
        pin.stream = st;
        pin.held = true;
        double *part_v = static_cast<double *>(sp);
        long long *part_i = reinterpret_cast<long long *>(part_v + n_curves * segs);
        hipLaunchKernelGGL(highest_peak_kernel, dim3((unsigned)(n_curves * segs)), dim3(kBlock), 0, st, d_power, nf, (int)segs,
                           d_idx, d_val, part_v, part_i);
        hipLaunchKernelGGL(highest_peak_merge_kernel, dim3((unsigned)n_curves), dim3(64), 0, st, part_v, part_i, (int)segs, d_idx, d_val);
    } else {
        hipLaunchKernelGGL(highest_peak_kernel, dim3((unsigned)n_curves), dim3(kBlock), 0, st, d_power, nf, 1, d_idx, d_val,
                           static_cast<double *>(nullptr), static_cast<long long *>(nullptr));
    }
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
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_p, power, n_curves * nf * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(pdc_highest_peak_dev(device, st, (double *)d_p, n_curves, nf, (int64_t *)d_i, (double *)d_v));
    if (idx_out) PDC_HIP(hipMemcpyAsync(idx_out, d_i, n_curves * 8, hipMemcpyDeviceToHost, st));
    if (val_out) PDC_HIP(hipMemcpyAsync(val_out, d_v, n_curves * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

namespace {

int launch_topk(hipStream_t st, PeakArgs a, int64_t n_curves) {
    PDC_REQUIRE(a.nf < ((int64_t)1 << 31), "peaks_topk: at most 2^31-1 bins per spectrum");
    int sh = kPkFusedShift;
    while (((a.nf + ((int64_t)1 << sh) - 1) >> sh) > kPkMaxBlocks) ++sh;
    a.blk_shift = sh;
    a.nblk = (a.nf + ((int64_t)1 << sh) - 1) >> sh;
    // LDS: 24 KB of candidate slots + 12.5 KB of block extrema at 5e4 bins - four workgroups per CU
    a.tile = 0;
    const size_t lds = (size_t)(a.nblk > 0 ? a.nblk : 1) * 16 + (size_t)kPkCap * 20 + 16;
    PDC_REQUIRE(lds <= 150 * 1024, "peaks_topk: %lld bins per spectrum need %zu bytes of LDS", (long long)a.nf, lds);
    if (a.k > kPkSortFrom || a.k_off > 0) {   // (the exclusion logic of later chunks lives in the SORT instance only: the few-ranks
                                              // instance keeps the 80 registers of six workgroups per CU)
        PDC_TRY(allow_dynamic_lds((const void *)peaks_topk_kernel<true>, 150 * 1024));
        hipLaunchKernelGGL(peaks_topk_kernel<true>, dim3((unsigned)n_curves), dim3(kPkBlock), lds, st, a);
    } else {
        PDC_TRY(allow_dynamic_lds((const void *)peaks_topk_kernel<false>, 150 * 1024));
        hipLaunchKernelGGL(peaks_topk_kernel<false>, dim3((unsigned)n_curves), dim3(kPkBlock), lds, st, a);
    }
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}

}  // namespace

int pdc_peaks_topk_dev(int device, void *stream, const double *d_power, int64_t n_curves, int64_t nf,
                       int k, int by_prominence, int64_t *d_count, int64_t *d_idx, double *d_height,
                       double *d_prominence, int64_t *d_half_lo, int64_t *d_half_hi) {
    PDC_REQUIRE(d_power || n_curves * nf == 0, "peaks_topk: power is NULL");
    PDC_REQUIRE(k >= 1 && k <= kPkMaxKTotal, "peaks_topk: k must be 1..%d", kPkMaxKTotal);
    PDC_REQUIRE(k <= kPkMaxK || (d_idx && (by_prominence ? d_prominence != nullptr : d_height != nullptr)),
                "peaks_topk: k > %d is ranked in chunks of %d, each after the chunk before: the indices and the ranking key's "
                "output (heights, or prominences) must be requested", kPkMaxK, kPkMaxK);
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
    a.k_total = k;
    // k > 128: one launch per 128 ranks; launch c ranks what comes after column 128 c - 1 of the outputs (every launch
    // sweeps the spectra again: ~1.2 ms per 64 ranks by height for the 1.64 GB of a C3 batch, 2-3 ms by prominence)
    for (int off = 0; off < k; off += kPkMaxK) {
        a.k = k - off < kPkMaxK ? k - off : kPkMaxK;
        a.k_off = off;
        a.count = off == 0 ? (long long *)d_count : nullptr;
        PDC_TRY(launch_topk((hipStream_t)stream, a, n_curves));
    }
    return PDC_OK;
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
    PDC_REQUIRE(k >= 1 && k <= kPkMaxKTotal, "peaks_topk: k must be 1..%d", kPkMaxKTotal);
    PDC_REQUIRE(n_curves >= 0 && nf >= 0, "peaks_topk: negative size");
    PDC_REQUIRE(count_out || idx_out || height_out || prominence_out || half_lo_out || half_hi_out,
                "peaks_topk: no output requested");
    if (n_curves == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    void *d_p;
    PDC_TRY(cached(device, SLOT_OUT0, n_curves * nf * 8, &d_p));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
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
    PDC_REQUIRE(k >= 1 && k <= kPkMaxKTotal, "gls_batch_peaks: k must be 1..%d", kPkMaxKTotal);
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
    PDC_TRY(host_stream(device, &st));
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
    PDC_TRY(host_stream(device, &st));
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
