// Phase Dispersion Minimization sweep on gfx950.
//
// Replaces pool.map(PDM._pdm, periods) (/root/reference/src/periodicity/phase.py:128-149,
// 185-187).  The reference's argsort (:132-134) only permutes the inputs of order-independent
// masks, so no sort is done: theta is a function of a (m0+1)-bin histogram of (count, sum x)
// over the fine phase bins [k/m0, (k+1)/m0), m0 = nb*nc; cover k of :137-140 is the union of
// fine bins k .. k+nc-1 (wrapping), and
//     sum_k (n_k - 1) var_k  =  sum_k ( Q_k - S_k^2 / n_k ),   sum_k Q_k = nc * Q_total
// (every sample lies in exactly nc covers; the one exception, phi == 1.0 exactly, is tracked in
// an overflow bin).  x is shifted by its mean first, which removes the cancellation.
//
// Mapping: one thread per trial period; the workgroup streams (t, x - mean) through LDS and every
// lane reads each sample as a broadcast.  Each thread keeps a PRIVATE histogram in LDS laid out
// [bin][thread] (bank = lane: conflict-free for any mix of bins) and updates it with one
// ds_add_f64 + one ds_add_u32 per pair.
//
// With few trial periods and many samples (the reference's default grid has 1000 periods) the
// samples are split over blockIdx.y as well: statistics once (pdm_stats_kernel, two passes), one
// partial histogram per slice of the samples, pdm_finish_kernel adds them in slice order.
//
// Bin membership must be bit-identical to numpy's: phi = (t/period) % 1 with an IEEE division
// and Python modulo, compared against the doubles k/m0.  The fast path uses t * (1/period) and
// accepts its bin only when phi is provably farther from every edge than the worst-case error of
// that shortcut; otherwise (about 1e-9 of the pairs) the lane redoes the sample with the exact
// division and explicit edge comparisons.
#include "pdc_internal.h"

#include <cstdlib>
#include <type_traits>

using namespace pdc;

namespace {

// samples staged per barrier (256 = the staging area a 256-thread workgroup owns anyway; 128 measured 2 %,
// 64 10 % slower at C5)
#ifndef PDC_PDM_CHUNK
#define PDC_PDM_CHUNK 256
#endif
constexpr int kChunk = PDC_PDM_CHUNK;

constexpr int kStatParts = 512;  // partial sums of the sample statistics (split mode)
constexpr int kCUs = 256;        // MI355X
constexpr int kLdsPerCU = 163840;

struct PdmArgs {
    const double *t, *x, *periods;
    int64_t n, n_periods;
    int nb, nc;
    double sigma;
    double *theta;
    // split mode (few trial periods, many samples): workgroup (g, z) bins the samples
    // [z * z_len, (z + 1) * z_len) of period group g and leaves its histograms in `psum`/`pcnt`
    // ([z][bin][p_pad]) and `pq` ([z][2][p_pad] = sum x'^2 of the overflow bin / of NaN phases);
    // pdm_finish_kernel adds them up in z order.  `stat` = [3][kStatParts] partial sum x,
    // max |t|, sum (x - mean)^2 from pdm_stats_kernel.
    double *psum = nullptr;
    unsigned *pcnt = nullptr;
    double *pq = nullptr;
    double *stat = nullptr;
    int64_t z_len = 0, p_pad = 0;
    int n_z = 1, n_stat = 0;
    // which statistic the epilogue evaluates from the same histograms: 0 = PDM theta (phase.py:128-149);
    // the scans the reference lists as TODO at phase.py:11-15: 1 = Analysis of Variance
    // (Schwarzenberg-Czerny 1989; nb phase bins), 2 = conditional entropy (Graham et al. 2013; nb
    // phase bins x nc magnitude bins, x = magnitude bin of every sample)
    int kind = 0;
};

template <int BLOCK>
__device__ __forceinline__ double block_reduce(double v, double *red, bool take_max) {
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_down(v, o, 64);
        v = take_max ? (u > v ? u : v) : v + u;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < BLOCK / 64; ++w) r = take_max ? (red[w] > r ? red[w] : r) : r + red[w];
    return r;
}

// Fixed-order fold of `count` partial values by a whole workgroup (identical in every workgroup).
template <int BLOCK>
__device__ __forceinline__ double fold_parts(const double *p, int count, double *red, bool take_max) {
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += BLOCK) v = take_max ? (p[i] > v ? p[i] : v) : v + p[i];
    return block_reduce<BLOCK>(v, red, take_max);
}

// Sample statistics for the split mode, grid-wide: pass 0 leaves partial sum x and max |t|, pass 1
// (after re-reducing pass 0's partials to the mean) partial sum (x - mean)^2.
__global__ __launch_bounds__(256) void pdm_stats_kernel(PdmArgs a, int pass) {
    __shared__ double red[4];
    double mean = 0.0;
    if (pass == 1) mean = fold_parts<256>(a.stat, a.n_stat, red, false) / (double)a.n;
    double acc = 0.0, tmax = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        if (pass == 0) {
            acc += a.x ? a.x[i] : 0.0;
            const double at = __builtin_fabs(a.t[i]);
            tmax = at > tmax ? at : tmax;
        } else {
            const double d = (a.x ? a.x[i] : 0.0) - mean;
            acc += d * d;
        }
    }
    acc = block_reduce<256>(acc, red, false);
    if (pass == 0) tmax = block_reduce<256>(tmax, red, true);
    if (threadIdx.x == 0) {
        if (pass == 0) {
            a.stat[blockIdx.x] = acc;
            a.stat[kStatParts + blockIdx.x] = tmax;
        } else {
            a.stat[2 * kStatParts + blockIdx.x] = acc;
        }
    }
}

// theta from one period's fine-bin histogram (phase.py:137-148); sum_at / cnt_at read bin b.
template <typename SumAt, typename CntAt>
__device__ __forceinline__ double theta_from_bins(SumAt sum_at, CntAt cnt_at, int m0, int nc, double q_total,
                                                  double q_nan, double q_over, double sigma) {
    double num = (double)nc * (q_total - q_nan) - q_over;
    long long n_sum = 0;
    int good = 0;
    for (int k = 0; k < m0; ++k) {
        double s = 0.0;
        long long c = 0;
        for (int j = 0; j < nc; ++j) {
            int b = k + j;
            if (b >= m0) {
                if (b == m0) {  // [1.0, (m0+1)/m0): only phi == 1.0 can live here
                    s += sum_at(m0);
                    c += cnt_at(m0);
                }
                b -= m0;
            }
            s += sum_at(b);
            c += cnt_at(b);
        }
        if (c > 1) {
            num -= s * s / (double)c;
            n_sum += c;
            ++good;
        } else if (c == 1) {
            num -= s * s;  // a singleton contributes x^2 - x^2 = 0 and is not a "good" bin
        }
    }
    // no cover with two or more members: the reference divides an empty sum by zero -> NaN
    // (phase.py:147); here `num` would only hold the rounding residue of the singletons
    return good == 0 ? __builtin_nan("") : (num / (double)(n_sum - good)) / sigma;
}

// Analysis of Variance (Schwarzenberg-Czerny 1989, MNRAS 241, 153, eq. 1-3) from the same histogram:
// r = m0 phase bins [k/r, (k+1)/r) (phi == 1.0 joins the last one), n valid samples,
//     s1^2 = sum_i n_i (xbar_i - xbar)^2 / (r - 1),   s2^2 = sum_i sum_j (x_ij - xbar_i)^2 / (n - r),
// Theta_AoV = s1^2 / s2^2.  With S_i = sum of the (mean-shifted) samples of bin i and Q their total
// square: between = sum S_i^2 / n_i - (sum S_i)^2 / n,  within = Q - sum S_i^2 / n_i.
template <typename SumAt, typename CntAt>
__device__ __forceinline__ double aov_from_bins(SumAt sum_at, CntAt cnt_at, int m0, double q_valid) {
    double per_bin = 0.0, s_all = 0.0;
    long long n = 0;
    for (int k = 0; k < m0; ++k) {
        double s = sum_at(k);
        long long c = cnt_at(k);
        if (k == m0 - 1) {
            s += sum_at(m0);
            c += cnt_at(m0);
        }
        if (c > 0) per_bin += s * s / (double)c;
        s_all += s;
        n += c;
    }
    if (n <= m0 || m0 < 2) return __builtin_nan("");
    const double between = per_bin - s_all * s_all / (double)n;
    const double within = q_valid - per_bin;
    return ((double)(n - m0) * between) / ((double)(m0 - 1) * within);
}

// Conditional entropy (Graham et al. 2013, MNRAS 434, 2629, eq. 1): H_c = sum_ij p(m_j, phi_i)
// ln(p(phi_i) / p(m_j, phi_i)) over the occupied cells of an m0 x mag (phase x magnitude) partition;
// cnt_at(i * mag + j) reads cell (i, j), row m0 (phi == 1.0) joins row m0 - 1.
template <typename CntAt>
__device__ __forceinline__ double ce_from_bins(CntAt cnt_at, int m0, int mag) {
    long long n = 0;
    for (int c = 0; c < (m0 + 1) * mag; ++c) n += cnt_at(c);
    if (n == 0) return __builtin_nan("");
    double h = 0.0;
    for (int i = 0; i < m0; ++i) {
        long long row = 0;
        for (int j = 0; j < mag; ++j) row += cnt_at(i * mag + j) + (i == m0 - 1 ? cnt_at(m0 * mag + j) : 0);
        for (int j = 0; j < mag; ++j) {
            const long long c = cnt_at(i * mag + j) + (i == m0 - 1 ? cnt_at(m0 * mag + j) : 0);
            if (c > 0) h += ((double)c / (double)n) * log((double)row / (double)c);
        }
    }
    return h;
}

// Gregory & Loredo (1992, ApJ 398, 146): arrival times t_i, model M_m = a periodic rate that is constant in
// each of m phase bins.  For trial frequency w and phase offset phi the likelihood depends on the data only
// through the multiplicity W_m(w, phi) = N! / (n_1! ... n_m!) of the bin counts (their eq. 5.13-5.14), and
// the marginal over the offset,
//     S_m(w) = (1 / 2 pi) Int dphi  m^N / W_m(w, phi),
// is what the odds ratio O_m1 (eq. 5.28) integrates over dw / w.  Here: ln S_m(w) with the offset integral as
// the mean over `offsets` equally spaced shifts of the bin boundaries by 1 / (m offsets) of a cycle - the
// counts of every shift are sums of `offsets` consecutive FINE bins of a histogram over F = m offsets bins
// [f / F, (f + 1) / F) (phi == 1.0 joins the last), cnt_at(f).  Log-sum-exp over the shifts, lgamma for the
// factorials.
template <typename CntAt>
__device__ __forceinline__ double gl_from_bins(CntAt cnt_at, int F, int m) {
    const int offsets = F / m;
    long long n = 0;
    for (int f = 0; f <= F; ++f) n += cnt_at(f);
    if (n == 0 || offsets < 1) return __builtin_nan("");
    const double base = (double)n * log((double)m) - lgamma((double)n + 1.0);
    double top = 0.0, sum = 0.0;
    for (int k = 0; k < offsets; ++k) {
        double lw = base;
        int f = k;
        for (int j = 0; j < m; ++j) {
            long long c = 0;
            for (int i = 0; i < offsets; ++i) {
                c += cnt_at(f) + (f == F - 1 ? cnt_at(F) : 0);
                f = f + 1 == F ? 0 : f + 1;
            }
            lw += lgamma((double)c + 1.0);
        }
        if (k == 0 || lw > top) {   // running log-sum-exp
            sum = k == 0 ? 1.0 : sum * exp(top - lw) + 1.0;
            top = lw;
        } else {
            sum += exp(lw - top);
        }
    }
    return top + log(sum / (double)offsets);
}

// SPLIT waves of a workgroup share one group of 64 trial periods (lane = period) and split every
// staged chunk of samples between them, so that a grid of ~1e5 periods still puts several waves on
// every SIMD (the loop is a chain of LDS read -> ALU -> LDS atomic: it needs wave-level parallelism
// to hide latency).  Each thread keeps its own private histogram; the SPLIT partial histograms of
// a period are summed in a fixed order at the end.
// ZS: split mode (the samples are split over blockIdx.y as well, see PdmArgs).
// KIND: which statistic (PdmArgs::kind; a template parameter so that the three show up as three kernels
// in a profile): 0 PDM theta, 1 AoV, 2 conditional entropy - cells (phase bin x magnitude bin, counts
// only) instead of phase bins.
template <int BLOCK, int SPLIT, bool ZS = false, int KIND = 0>
__global__ __launch_bounds__(BLOCK) void pdm_scan_kernel(PdmArgs a) {
    constexpr bool CE = KIND == 2 || KIND == 4;   // counts only
    constexpr bool GL = KIND == 4;                // ... of arrival times alone: x is not read
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int mag = KIND == 2 ? a.nc : 1;          // magnitude bins per phase bin
    const int m0 = CE ? a.nb : a.nb * a.nc;        // phase bins (Gregory-Loredo: the fine bins, nb = m * offsets)
    const int nbins = (m0 + 1) * mag;              // + overflow row for phi == 1.0
    double2 *stage = reinterpret_cast<double2 *>(lds_raw);                  // [kChunk] (t, x')
    // the staging area doubles as the q_over/q_nan exchange ([2][BLOCK] doubles) at the end
    constexpr int kStage = kChunk > BLOCK ? kChunk : BLOCK;
    double *hsum = reinterpret_cast<double *>(stage + kStage);              // [nbins][BLOCK] (not with CE: counts only)
    // counts: one word per bin - or, for the counts-only kinds, two 16-bit cells per word (a thread's cells
    // count at most the samples of its workgroup's slice, kept <= 65 280 by the launcher: the histogram of 55
    // cells x 256 threads then takes 28 KB instead of 56, five workgroups per CU instead of two)
    const int ncw = CE ? (nbins + 1) / 2 : nbins;                                  // count words per thread
    unsigned *hcnt = reinterpret_cast<unsigned *>(hsum + (CE ? 0 : (size_t)nbins * BLOCK));  // [ncw][BLOCK]
    double *edge = reinterpret_cast<double *>(hcnt + (size_t)ncw * BLOCK);         // [m0 + 2] (BLOCK words: 8-byte aligned)
    double *red = edge + m0 + 2;                                                  // [BLOCK/64]
    const int tid = threadIdx.x;

    for (int k = tid; k < m0 + 2; k += BLOCK) edge[k] = (double)k / (double)m0;  // Python's k / m0
    for (int k = 0; k < nbins; ++k)
        if (!CE) hsum[k * BLOCK + tid] = 0.0;
    for (int k = 0; k < ncw; ++k) hcnt[k * BLOCK + tid] = 0u;

    // mean of x, total sum of squares about it, and max |t| (identical in every workgroup)
    double mean, tmax, q_total;
    if (ZS) {
        mean = fold_parts<BLOCK>(a.stat, a.n_stat, red, false) / (double)a.n;
        tmax = fold_parts<BLOCK>(a.stat + kStatParts, a.n_stat, red, true);
        q_total = fold_parts<BLOCK>(a.stat + 2 * kStatParts, a.n_stat, red, false);
    } else {
        double acc = 0.0;
        tmax = 0.0;
        for (int64_t i = tid; i < a.n; i += BLOCK) {
            acc += GL ? 0.0 : a.x[i];
            const double at = __builtin_fabs(a.t[i]);
            tmax = at > tmax ? at : tmax;
        }
        mean = block_reduce<BLOCK>(acc, red, false) / (double)a.n;
        tmax = block_reduce<BLOCK>(tmax, red, true);
        acc = 0.0;
        for (int64_t i = tid; i < a.n && !CE; i += BLOCK) {
            const double d = a.x[i] - mean;
            acc += d * d;
        }
        q_total = block_reduce<BLOCK>(acc, red, false);
    }

    constexpr int PERIODS = BLOCK / SPLIT;       // trial periods per workgroup
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), part = wave % SPLIT;  // scalar
    const int slot = (wave / SPLIT) * 64 + (tid & 63);  // period slot inside the workgroup
    const int64_t pidx = (int64_t)blockIdx.x * PERIODS + slot;
    const double period = pidx < a.n_periods ? a.periods[pidx] : 1.0;
    const double rp = 1.0 / period;
    const double dm0 = (double)m0;
    // worst-case error of u = frac(t*rp)*m0 against the exact ((t/period) % 1) vs edge k/m0, in
    // units of u: quotient error <= 1.5 ulp(q) <= 3.4e-16 |q|, product/edge roundings 2.3e-16 m0;
    // doubled for safety.
    const double eps = dm0 * (8.9e-16 * tmax * __builtin_fabs(rp) + 8.9e-16);
    const double thr = 0.5 - eps;  // accept when |g - 0.5| < thr
    double q_over = 0.0;           // sum x'^2 of samples in the overflow bin
    double q_nan = 0.0;            // sum x'^2 of samples whose phase is NaN (they are in no cover)

    // this workgroup's samples: all of them, or slice blockIdx.y of them in split mode
    const int64_t s_begin = ZS ? (int64_t)blockIdx.y * a.z_len : 0;
    const int64_t s_end = ZS ? (s_begin + a.z_len < a.n ? s_begin + a.z_len : a.n) : a.n;
    for (int64_t base = s_begin; base < s_end; base += kChunk) {
        __syncthreads();
        for (int i = tid; i < kChunk; i += BLOCK) {
            const int64_t g = base + i;
            double2 v = g < s_end ? make_double2(a.t[g], GL ? 0.0 : (CE ? a.x[g] : a.x[g] - mean)) : make_double2(0.0, 0.0);
            if (KIND == 2 && !(v.y >= 0.0 && v.y < (double)mag)) {
                // the magnitude bin is the caller's double and indexes the cell histogram: anything outside
                // 0 .. mag-1 (NaN included) is staged with a NaN time - its phase is NaN, so it takes the exact
                // path and counts nowhere, exactly as if the sample were absent (the host entries reject such
                // input; `_dev` callers get this).  Checked once per staged sample, not once per pair.
                v = make_double2(__builtin_nan(""), 0.0);
            }
            stage[i] = v;
        }
        __syncthreads();
        const int cnt = (int)((s_end - base) < kChunk ? (s_end - base) : kChunk);
        const int i_end = cnt < (part + 1) * (kChunk / SPLIT) ? cnt : (part + 1) * (kChunk / SPLIT);
        // two samples per trip: two independent read -> bin -> atomic chains in flight per wave.
        // Fast path: 7 VALU ops (v_fract_f64 twice); the histogram update is unconditional.
        auto fast_bin = [&](const double t, int &k) -> bool {
            const double u = __builtin_amdgcn_fract(t * rp) * dm0;
            k = (int)u;
            return __builtin_fabs(__builtin_amdgcn_fract(u) - 0.5) < thr;
        };
        // exact path: numpy's float remainder of the IEEE quotient, explicit edges; a NaN phase
        // belongs to no bin (adds zero to bin 0)
        auto exact_bin = [&](const double2 tx, int &k, double &val, unsigned &inc) {
            const double qe = tx.x / period;
            const double phi = qe - __builtin_floor(qe);  // == fmod-based Python % for divisor 1
            if (phi != phi) {
                q_nan += tx.y * tx.y;
                k = 0;
                val = 0.0;
                inc = 0u;
                return;
            }
            k = (int)(phi * dm0);
            k = k < 0 ? 0 : (k > m0 ? m0 : k);
            while (k > 0 && phi < edge[k]) --k;
            while (k < m0 && phi >= edge[k + 1]) ++k;
            if (k == m0) q_over += tx.y * tx.y;
        };
        auto add = [&](const int k, const double val, const unsigned inc) {
            if (CE) {   // val = the sample's magnitude bin (range-checked when staged); a NaN phase (inc == 0) counts nowhere
                const int cell = k * mag + (int)val;
                atomicAdd(&hcnt[(cell >> 1) * BLOCK + tid], inc << ((cell & 1) * 16));
            } else {
                atomicAdd(&hsum[k * BLOCK + tid], val);
                atomicAdd(&hcnt[k * BLOCK + tid], inc);
            }
        };
        auto update = [&](const double2 tx) {
            int k;
            double val = tx.y;
            unsigned inc = 1u;
            if (!fast_bin(tx.x, k)) exact_bin(tx, k, val, inc);
            add(k, val, inc);
        };
        int i = part * (kChunk / SPLIT);
        // the samples of the next trip are read before this trip's histogram atomics go out (the compiler
        // cannot move an LDS read above a possibly aliasing LDS atomic by itself); the staging area
        // is followed by the histograms, so reading a few entries past the chunk is harmless
        {   // four samples per trip: four independent read -> bin -> atomic chains in flight per wave (two per
            // trip measured 3.5-8 % slower on one box); the next four are read before this trip's atomics go out
            double2 n0 = stage[i], n1 = stage[i + 1], n2 = stage[i + 2], n3 = stage[i + 3];
            for (; i + 3 < i_end; i += 4) {
                const double2 t0 = n0, t1 = n1, t2 = n2, t3 = n3;
                n0 = stage[i + 4];
                n1 = stage[i + 5];
                n2 = stage[i + 6];
                n3 = stage[i + 7];
                int k0, k1, k2, k3;
                double v0 = t0.y, v1 = t1.y, v2 = t2.y, v3 = t3.y;
                unsigned i0 = 1u, i1 = 1u, i2 = 1u, i3 = 1u;
                const bool f0 = fast_bin(t0.x, k0), f1 = fast_bin(t1.x, k1), f2 = fast_bin(t2.x, k2), f3 = fast_bin(t3.x, k3);
                if (!f0) exact_bin(t0, k0, v0, i0);
                add(k0, v0, i0);
                if (!f1) exact_bin(t1, k1, v1, i1);
                add(k1, v1, i1);
                if (!f2) exact_bin(t2, k2, v2, i2);
                add(k2, v2, i2);
                if (!f3) exact_bin(t3, k3, v3, i3);
                add(k3, v3, i3);
            }
        }
        double2 na = stage[i], nb2 = stage[i + 1];
        for (; i + 1 < i_end; i += 2) {
            // both fast bins first: two independent dependency chains back to back
            const double2 ta = na, tb = nb2;
            na = stage[i + 2];
            nb2 = stage[i + 3];
            int ka, kb;
            double va = ta.y, vb = tb.y;
            unsigned ia = 1u, ib = 1u;
            const bool fa = fast_bin(ta.x, ka), fb = fast_bin(tb.x, kb);
            if (!fa) exact_bin(ta, ka, va, ia);
            add(ka, va, ia);
            if (!fb) exact_bin(tb, kb, vb, ib);
            add(kb, vb, ib);
        }
        for (; i < i_end; ++i) update(stage[i]);
    }

    if (SPLIT > 1) {
        // fold the partial histograms of parts 1..SPLIT-1 into part 0 (threads tid + 64*q)
        __syncthreads();
        double *qx = reinterpret_cast<double *>(stage);  // q_over / q_nan exchange, [2][BLOCK]
        qx[tid] = q_over;
        qx[BLOCK + tid] = q_nan;
        __syncthreads();
        if (part == 0) {
            for (int q = 1; q < SPLIT; ++q) {
                const int other = tid + 64 * q;
                for (int k = 0; k < nbins; ++k)
                    if (!CE) hsum[k * BLOCK + tid] += hsum[k * BLOCK + other];
                for (int k = 0; k < ncw; ++k) hcnt[k * BLOCK + tid] += hcnt[k * BLOCK + other];   // (fields cannot carry)
                q_over += qx[other];
                q_nan += qx[BLOCK + other];
            }
        }
    }
    if (part != 0 || pidx >= a.n_periods) return;
    if (ZS) {  // split mode: leave the histogram of this slice of the samples for pdm_finish_kernel
        const int64_t z = blockIdx.y;
        for (int k = 0; k < nbins; ++k)
            if (!CE) a.psum[(z * nbins + k) * a.p_pad + pidx] = hsum[k * BLOCK + tid];
        for (int k = 0; k < ncw; ++k) a.pcnt[(z * ncw + k) * a.p_pad + pidx] = hcnt[k * BLOCK + tid];
        if (!CE) {
            a.pq[(z * 2 + 0) * a.p_pad + pidx] = q_over;
            a.pq[(z * 2 + 1) * a.p_pad + pidx] = q_nan;
        }
        return;
    }
    auto sum_at = [&](int b) { return hsum[b * BLOCK + tid]; };
    auto cnt_at = [&](int b) {
        return CE ? (long long)((hcnt[(b >> 1) * BLOCK + tid] >> ((b & 1) * 16)) & 0xFFFFu) : (long long)hcnt[b * BLOCK + tid];
    };
    if (GL) a.theta[pidx] = gl_from_bins(cnt_at, m0, a.nc);
    else if (CE) a.theta[pidx] = ce_from_bins(cnt_at, m0, mag);
    else if (KIND == 1) a.theta[pidx] = aov_from_bins(sum_at, cnt_at, m0, q_total - q_nan);
    else a.theta[pidx] = theta_from_bins(sum_at, cnt_at, m0, a.nc, q_total, q_nan, q_over, a.sigma);
}

// Split mode, last step: one thread per trial period adds the slices' histograms in slice order
// (LDS laid out like the scan kernel's, [bin][thread]) and evaluates theta.
// (64 periods per workgroup - lane = period - times kFinZ waves that share the slices: few periods and many samples
// mean up to 1024 slices, and one thread adding 1024 x bins partial sums one after the other took 1.4 ms for 64
// periods at N = 1e6.  Wave zy adds the slices z = zy (mod kFinZ) in z order, the waves' sums are added in wave order:
// a fixed order again, whatever the slice count.)
constexpr int kFinZ = 16;
__global__ __launch_bounds__(64 * kFinZ) void pdm_finish_kernel(PdmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ double part_s[kFinZ][64];
    __shared__ long long part_a[kFinZ][64], part_b[kFinZ][64];
    __shared__ double s_q_total;
    const bool counts_only = a.kind == 2 || a.kind == 4;
    const int m0 = counts_only ? a.nb : a.nb * a.nc;
    const int nbins = a.kind == 2 ? (a.nb + 1) * a.nc : m0 + 1, tid = threadIdx.x & 63, zy = threadIdx.x >> 6;
    // sums [nbins][64] then counts [nbins][64]; the counts-only kinds keep no sums
    double *hsum = reinterpret_cast<double *>(lds_raw);
    long long *hcnt = reinterpret_cast<long long *>(hsum + (counts_only ? 0 : (size_t)nbins * 64));
    if (zy == 0) {   // (the fold of the statistics' partial sums as a 64-thread workgroup does it)
        double v = 0.0;
        for (int i = tid; i < a.n_stat; i += 64) v += a.stat[2 * kStatParts + i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (tid == 0) s_q_total = v;
    }
    const int64_t pidx = (int64_t)blockIdx.x * 64 + tid;
    const bool live = pidx < a.n_periods;
    const int64_t pc = live ? pidx : 0;
    // one cell (or pair of 16-bit cells) at a time: every wave its share of the slices, then wave 0 the waves
    auto over_waves = [&](auto per_slice, auto store) {
        double ps = 0.0;
        long long pa = 0, pb = 0;
        for (int64_t z = zy; z < a.n_z; z += kFinZ) per_slice(z, ps, pa, pb);
        part_s[zy][tid] = ps;
        part_a[zy][tid] = pa;
        part_b[zy][tid] = pb;
        __syncthreads();
        if (zy == 0) {
            double ts = part_s[0][tid];
            long long ta = part_a[0][tid], tb = part_b[0][tid];
            for (int w = 1; w < kFinZ; ++w) {
                ts += part_s[w][tid];
                ta += part_a[w][tid];
                tb += part_b[w][tid];
            }
            store(ts, ta, tb);
        }
        __syncthreads();
    };
    double q_over = 0.0, q_nan = 0.0;
    if (counts_only) {   // two 16-bit cells per word in a slice's partial histogram; their sum over the slices is not
        const int ncw = (nbins + 1) / 2;
        for (int k = 0; k < ncw; ++k)
            over_waves([&](int64_t z, double &, long long &lo, long long &hi) {
                           const unsigned w = a.pcnt[(z * ncw + k) * a.p_pad + pc];
                           lo += w & 0xFFFFu;
                           hi += w >> 16;
                       },
                       [&](double, long long lo, long long hi) {
                           hcnt[(2 * k) * 64 + tid] = lo;
                           if (2 * k + 1 < nbins) hcnt[(2 * k + 1) * 64 + tid] = hi;
                       });
    } else {
        for (int k = 0; k < nbins; ++k)
            over_waves([&](int64_t z, double &sum, long long &cnt, long long &) {
                           sum += a.psum[(z * nbins + k) * a.p_pad + pc];
                           cnt += a.pcnt[(z * nbins + k) * a.p_pad + pc];
                       },
                       [&](double sum, long long cnt, long long) {
                           hsum[k * 64 + tid] = sum;
                           hcnt[k * 64 + tid] = cnt;
                       });
        for (int which = 0; which < 2; ++which)
            over_waves([&](int64_t z, double &sum, long long &, long long &) { sum += a.pq[(z * 2 + which) * a.p_pad + pc]; },
                       [&](double sum, long long, long long) { (which == 0 ? q_over : q_nan) = sum; });
    }
    if (zy != 0 || !live) return;
    const double q_total = s_q_total;
    auto sum_at = [&](int b) { return hsum[b * 64 + tid]; };
    auto cnt_at = [&](int b) { return hcnt[b * 64 + tid]; };
    if (a.kind == 4) a.theta[pidx] = gl_from_bins(cnt_at, m0, a.nc);
    else if (a.kind == 2) a.theta[pidx] = ce_from_bins(cnt_at, m0, a.nc);
    else if (a.kind == 1) a.theta[pidx] = aov_from_bins(sum_at, cnt_at, m0, q_total - q_nan);
    else a.theta[pidx] = theta_from_bins(sum_at, cnt_at, m0, a.nc, q_total, q_nan, q_over, a.sigma);
}

// `last` = highest histogram bin; bytes_per_bin 12: sum + count, 4: counts only (two 16-bit cells per word)
size_t lds_bytes(int last, int block, int bytes_per_bin = 12) {
    const size_t stage = (size_t)(kChunk > block ? kChunk : block) * 16;
    const size_t nbins = (size_t)last + 1;
    const size_t hist = bytes_per_bin == 4 ? ((nbins + 1) / 2 + 1) * block * 4 : nbins * block * 12;
    return stage + hist + (size_t)(last + 2) * 8 + 64;
}

// dynamic-LDS limit of a kernel, raised once per device (not on every call)
template <typename Kernel>
int allow_lds(Kernel kernel) {
    return allow_dynamic_lds((const void *)kernel, 150 * 1024);
}

// Split mode (few trial periods, many samples; see phase_stat_dev): how the samples are cut and how
// much scratch the partial histograms take.  n_z == 1: not split.
struct SplitShape {
    bool use = false;   // take the split mode (possibly with ONE slice: statistics once per launch, not per workgroup)
    int64_t n_z = 1, z_len = 0, p_pad = 0, bytes = 0;
    int n_stat = 0;
    size_t stat_b = 0, psum_b = 0, pq_b = 0, pcnt_b = 0;
};

constexpr int64_t kCellSamples = 65280;   // samples a workgroup of a counts-only kind may bin: its cells are 16-bit

SplitShape split_shape(int kind, int64_t n, int64_t n_periods, int nb, int nc) {
    static const int env_split = [] { const char *e = getenv("PDC_PDM_SPLIT"); return e ? atoi(e) : -1; }();
    SplitShape sh;
    if (kind == 1) nc = 1;
    const int m0 = kind >= 2 ? nb : nb * nc;
    const int last = kind == 2 ? (nb + 1) * nc - 1 : m0;
    const int nbins = last + 1;
    const int64_t groups0 = (n_periods + 63) / 64;
    const bool counts_only = kind >= 2;
    // the counts-only kinds MUST cut more than kCellSamples samples into slices (16-bit cells); everything
    // else about the split is a choice, which PDC_PDM_SPLIT=0 turns off
    const int64_t must_z = counts_only ? (n + kCellSamples - 1) / kCellSamples : 1;
    if (n_periods == 0 || n == 0) return sh;
    const bool may = env_split != 0 && n >= 32 * kChunk && lds_bytes(last, 256, counts_only ? 4 : 12) <= 150 * 1024 &&
                     (size_t)nbins * 64 * (counts_only ? 8 : 16) <= 128 * 1024;   // (+ 24 KB of the finishing kernel's own)
    if (!may && must_z <= 1) return sh;
    const int64_t max_z = n / (8 * kChunk) > must_z ? n / (8 * kChunk) : must_z;
    int64_t n_z = 1;
    if (!may) {
        n_z = must_z;
    } else if (groups0 * 4 < 3072) {
        // (four workgroups of four waves fit a CU: below ~3000 waves the period grid alone leaves SIMD slots empty)
        n_z = (4096 + groups0 * 4 - 1) / (groups0 * 4);
    } else {
        // Enough waves - but workgroups come in rounds of kSlots (4 per CU, LDS-limited), and a last round
        // that is half empty costs as much as a full one: C5's 1563 workgroups take two rounds for 1.53
        // rounds' worth of work.  Cutting the samples in n_z slices multiplies the workgroups (each a slice
        // shorter): take the n_z <= 16 with the best (share of its rounds filled) - (cost of its slices), or
        // the unsplit grid if that scores higher.  PDC_PDM_NZ forces a value (experiments).
        static const int env_nz = [] { const char *e = getenv("PDC_PDM_NZ"); return e ? atoi(e) : 0; }();
        // resident workgroups of the scan kernel: LDS-limited (PDM at nb x nc = 10: 38 KB -> 4 per CU, 1024 on
        // the chip; the counts-only kinds with many cells hold fewer), at most 8 per CU (4 waves each)
        const int64_t per_cu = kLdsPerCU / (int64_t)(lds_bytes(last, 256, counts_only ? 4 : 12) + 512);
        const int64_t kSlots = kCUs * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
        auto rounds_filled = [&](int64_t w) { return (double)w / (double)((w + kSlots - 1) / kSlots * kSlots); };
        // the unsplit launch packs 64, 128 or 256 periods into a workgroup (phase_stat_dev below), and every
        // one of its workgroups reduces the sample statistics by itself (two passes over all samples: ~5 % of
        // a C5 workgroup's time) where the split mode has them from two short launches.  A slice costs its
        // partial histograms through L2 and a share of the finishing launch: ~1.2 % each (measured at C5:
        // 3 slices 2.74 ms, 5 2.66, 8 2.65, 13 2.71, 16 2.73; unsplit with 99 % of its rounds filled at N = 1e5 x
        // 1.3e5 periods 6.50 ms against 5.91 in 5 slices)
        const int64_t w_unsplit = groups0 >= 4096 ? (groups0 + 3) / 4 : (groups0 >= 2048 ? (groups0 + 1) / 2 : groups0);
        double best = rounds_filled(w_unsplit) - 0.05;
        n_z = 0;   // 0: unsplit
        // (slices of at most ~16k samples measured best wherever they were tried - N = 5e4: 5 slices, 1e5: 8 -
        // beyond what the filled rounds explain)
        int64_t z_min = (n + 16383) / 16384;
        z_min = z_min > 16 ? 16 : z_min;
        for (int64_t z = z_min; z <= 16 && z <= (max_z > 1 ? max_z : 1); ++z) {
            // (a slice's cost is per period, the work it is paid from grows with N: 1.2 % per slice at N = 5e4)
            const double score = rounds_filled(groups0 * z) - 0.012 * (double)z * (5e4 / (double)n);
            if (score > best) {
                best = score;
                n_z = z;
            }
        }
        if (env_nz > 0) n_z = env_nz;
        if (n_z == 0 && must_z <= 1) return sh;
        if (n_z == 0) n_z = must_z;
    }
    n_z = n_z < max_z ? n_z : max_z;
    n_z = n_z > must_z ? n_z : must_z;
    n_z = n_z < 1 ? 1 : n_z;
    int64_t z_len = ((n + n_z - 1) / n_z + kChunk - 1) / kChunk * kChunk;
    if (counts_only && z_len > kCellSamples) z_len = kCellSamples;   // (kCellSamples is a multiple of kChunk)
    n_z = (n + z_len - 1) / z_len;
    sh.use = true;
    sh.n_z = n_z;
    sh.z_len = z_len;
    sh.p_pad = groups0 * 64;
    sh.n_stat = (int)((n + 1023) / 1024 < kStatParts ? (n + 1023) / 1024 : kStatParts);
    sh.stat_b = (size_t)3 * kStatParts * 8;
    sh.psum_b = counts_only ? 0 : (size_t)n_z * nbins * sh.p_pad * 8;
    sh.pq_b = counts_only ? 0 : (size_t)n_z * 2 * sh.p_pad * 8;
    sh.pcnt_b = (size_t)n_z * (counts_only ? (nbins + 1) / 2 : nbins) * sh.p_pad * 4;
    sh.bytes = (int64_t)(sh.stat_b + sh.psum_b + sh.pq_b + sh.pcnt_b);
    return sh;
}

}  // namespace

int64_t pdc::phase_stat_work_bytes(int kind, int64_t n, int64_t n_periods, int nb, int nc) {
    if (kind < 0 || kind > 4 || kind == 3 || n < 0 || n_periods < 0 || nb < 1 || nc < 1) return -1;
    return split_shape(kind, n, n_periods, nb, nc).bytes;
}

// kind 0: PDM theta; 1: AoV over nb phase bins; 2: conditional entropy over nb x nc cells.
// `work`: scratch of at least phase_stat_work_bytes() bytes for the split mode's partial histograms;
// NULL = take it from the per-(device, stream) cache (the `_dev` entry points without a workspace).
int pdc::phase_stat_dev(int kind, int device, hipStream_t st, const double *d_t, const double *d_x, int64_t n,
                        const double *d_periods, int64_t n_periods, int nb, int nc, double sigma, double *d_out,
                        void *work, int64_t work_bytes) {
    PDC_REQUIRE(d_t && (d_x || kind == 4) && (d_periods || n_periods == 0) && (d_out || n_periods == 0),
                "phase scan: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "phase scan: negative size");
    PDC_REQUIRE(nb >= 1 && nc >= 1, "phase scan: bin counts must be positive");
    PDC_REQUIRE(kind != 4 || nb % nc == 0, "gregory_loredo: the fine bins (%d) must be a multiple of m (%d)", nb, nc);
    if (kind == 1) nc = 1;
    const int m0 = kind >= 2 ? nb : nb * nc;                    // phase bins
    const int last = kind == 2 ? (nb + 1) * nc - 1 : m0;        // highest histogram bin
    PDC_REQUIRE(last <= 190, "phase scan: %d histogram bins exceed the 191 that fit in LDS", last + 1);
    if (n_periods == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    PdmArgs a{d_t, d_x, d_periods, n, n_periods, nb, nc, sigma, d_out};
    a.kind = kind;
    // Few trial periods and many samples (the reference's default grid has 1000 periods): lanes are
    // periods, so the period grid alone would leave most of the chip idle.  Split the SAMPLES over
    // blockIdx.y as well: statistics once (two short grid-wide launches), partial histograms per
    // slice, one finishing launch that adds them in slice order.  PDC_PDM_SPLIT=0 disables it.
    const int64_t groups0 = (n_periods + 63) / 64;
    const int nbins = last + 1;
    const SplitShape sh = split_shape(kind, n, n_periods, nb, nc);
    if (sh.use) {
        a.p_pad = sh.p_pad;
        a.n_z = (int)sh.n_z;
        a.z_len = sh.z_len;
        a.n_stat = sh.n_stat;
        PDC_TRY(allow_dynamic_lds((const void *)pdm_finish_kernel, 128 * 1024));   // (+ 24 KB of its own)
        void *spv = work;
        ScratchPin pin;
        if (work) {
            PDC_REQUIRE(work_bytes >= sh.bytes, "phase scan: workspace too small (%lld < %lld bytes)",
                        (long long)work_bytes, (long long)sh.bytes);
        } else {   // cached per (device, stream): see pdc_internal.h on why not hipMallocAsync
            PDC_TRY(stream_scratch(device, st, sh.bytes, &spv));
            pin.device = device;
            pin.stream = st;
            pin.held = true;
        }
        char *const sp = static_cast<char *>(spv);
        // PDC_PDM_POISON=1 fills the scratch with a NaN pattern first (debugging aid: every word the kernels
        // read must have been written by them)
        static const bool poison = [] { const char *e = getenv("PDC_PDM_POISON"); return e && e[0] == '1'; }();
        if (poison) PDC_HIP(hipMemsetAsync(sp, 0x7f, (size_t)sh.bytes, st));
        a.stat = reinterpret_cast<double *>(sp);
        a.psum = reinterpret_cast<double *>(sp + sh.stat_b);
        a.pq = reinterpret_cast<double *>(sp + sh.stat_b + sh.psum_b);
        a.pcnt = reinterpret_cast<unsigned *>(sp + sh.stat_b + sh.psum_b + sh.pq_b);
        hipLaunchKernelGGL(pdm_stats_kernel, dim3((unsigned)a.n_stat), dim3(256), 0, st, a, 0);
        hipLaunchKernelGGL(pdm_stats_kernel, dim3((unsigned)a.n_stat), dim3(256), 0, st, a, 1);
        // (the statistic only matters to the finishing launch - and, for the counts-only kinds, to what the
        // histogram holds; it is a template argument so that a profile tells the launches apart)
        const size_t zlds = lds_bytes(last, 256, kind >= 2 ? 4 : 12);
        const dim3 zgrid((unsigned)groups0, (unsigned)sh.n_z);
        auto zlaunch = [&](auto kernel) -> int {
            PDC_TRY(allow_lds(kernel));
            hipLaunchKernelGGL(kernel, zgrid, dim3(256), zlds, st, a);
            return PDC_OK;
        };
        if (kind == 4) PDC_TRY(zlaunch(pdm_scan_kernel<256, 4, true, 4>));
        else if (kind == 2) PDC_TRY(zlaunch(pdm_scan_kernel<256, 4, true, 2>));
        else if (kind == 1) PDC_TRY(zlaunch(pdm_scan_kernel<256, 4, true, 1>));
        else PDC_TRY(zlaunch(pdm_scan_kernel<256, 4, true, 0>));
        hipLaunchKernelGGL(pdm_finish_kernel, dim3((unsigned)groups0), dim3(64 * kFinZ), (size_t)nbins * 64 * (kind >= 2 ? 8 : 16), st, a);
        PDC_HIP(hipGetLastError());
        return PDC_OK;
    }
    const int bpb = kind >= 2 ? 4 : 12;
    auto launch = [&](auto kernel, int block, int periods_per_block) -> int {
        PDC_TRY(allow_lds(kernel));
        hipLaunchKernelGGL(kernel, dim3((unsigned)((n_periods + periods_per_block - 1) / periods_per_block)),
                           dim3((unsigned)block), lds_bytes(last, block, bpb), st, a);
        return PDC_OK;
    };
    auto launch_kind = [&](auto kind_tag) -> int {
        constexpr int K = decltype(kind_tag)::value;
        if (lds_bytes(last, 256, bpb) > 150 * 1024) return launch(pdm_scan_kernel<64, 1, false, K>, 64, 64);
        // waves = ceil(P/64) * SPLIT; aim at >= 4 waves per SIMD (4096 on the chip)
        const int64_t groups = (n_periods + 63) / 64;
        if (groups >= 4096) return launch(pdm_scan_kernel<256, 1, false, K>, 256, 256);
        if (groups >= 2048) return launch(pdm_scan_kernel<256, 2, false, K>, 256, 128);
        return launch(pdm_scan_kernel<256, 4, false, K>, 256, 64);
    };
    if (kind == 4) PDC_TRY(launch_kind(std::integral_constant<int, 4>{}));
    else if (kind == 2) PDC_TRY(launch_kind(std::integral_constant<int, 2>{}));
    else if (kind == 1) PDC_TRY(launch_kind(std::integral_constant<int, 1>{}));
    else PDC_TRY(launch_kind(std::integral_constant<int, 0>{}));
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}

extern "C" {

int64_t pdc_phase_work_bytes(int kind, int64_t n, int64_t n_periods, int nb, int nc) {
    if (kind == 3) return pdc_stringlength_work_bytes(n, n_periods);
    if (kind == 5) return pdc_supersmoother_work_bytes(n, n_periods);
    return phase_stat_work_bytes(kind, n, n_periods, nb, nc);
}

int pdc_phase_scan_dev(int kind, int device, void *stream, const double *d_t, const double *d_v, int64_t n,
                       const double *d_periods, int64_t n_periods, int nb, int nc, double sigma, double *d_out,
                       void *work, int64_t work_bytes) {
    PDC_REQUIRE(kind >= 0 && kind <= 5, "phase scan: kind must be 0 (PDM), 1 (AoV), 2 (conditional entropy), 3 "
                                        "(StringLength), 4 (Gregory-Loredo) or 5 (Supersmoother; sigma = alpha)");
    if (kind == 5)
        return pdc_supersmoother_scan_dev(device, stream, d_t, d_v, n, d_periods, n_periods, sigma, d_out, work, work_bytes);
    if (kind == 3)
        return pdc_stringlength_scan_dev(device, stream, d_t, d_v, n, d_periods, n_periods, d_out, work, work_bytes);
    const int64_t need = phase_stat_work_bytes(kind, n, n_periods, nb, nc);
    PDC_REQUIRE(need >= 0, "phase scan: bad size");
    PDC_REQUIRE(need == 0 || (work && work_bytes >= need), "phase scan: workspace too small (%lld < %lld bytes)",
                (long long)work_bytes, (long long)need);
    static char dummy;   // (a non-NULL workspace keeps phase_stat_dev away from the per-stream cache)
    return phase_stat_dev(kind, device, (hipStream_t)stream, d_t, d_v, n, d_periods, n_periods, nb, nc, sigma, d_out,
                          work ? work : &dummy, work ? work_bytes : 0);
}

int pdc_pdm_scan_dev(int device, void *stream, const double *d_t, const double *d_x, int64_t n,
                     const double *d_periods, int64_t n_periods, int nb, int nc, double sigma,
                     double *d_theta) {
    return phase_stat_dev(0, device, (hipStream_t)stream, d_t, d_x, n, d_periods, n_periods, nb, nc, sigma,
                          d_theta, nullptr, 0);
}

int pdc_aov_scan_dev(int device, void *stream, const double *d_t, const double *d_x, int64_t n,
                     const double *d_periods, int64_t n_periods, int n_bins, double *d_theta) {
    return phase_stat_dev(1, device, (hipStream_t)stream, d_t, d_x, n, d_periods, n_periods, n_bins, 1, 1.0,
                          d_theta, nullptr, 0);
}

int pdc_cond_entropy_scan_dev(int device, void *stream, const double *d_t, const double *d_mag_bin, int64_t n,
                              const double *d_periods, int64_t n_periods, int n_phase, int n_mag,
                              double *d_entropy) {
    return phase_stat_dev(2, device, (hipStream_t)stream, d_t, d_mag_bin, n, d_periods, n_periods, n_phase,
                          n_mag, 1.0, d_entropy, nullptr, 0);
}

namespace {

int phase_stat_host(int kind, const double *t, const double *x, int64_t n, const double *periods,
                    int64_t n_periods, int nb, int nc, double sigma, double *out, int device) {
    PDC_REQUIRE(t && (x || kind == 4) && (periods || n_periods == 0) && (out || n_periods == 0),
                "phase scan: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "phase scan: negative size");
    if (kind == 2)   // the magnitude bins index the cell histogram: reject what is not one of 0 .. n_mag-1
        for (int64_t i = 0; i < n; ++i)
            PDC_REQUIRE(x[i] >= 0.0 && x[i] < (double)nc,
                        "cond_entropy: mag_bin[%lld] = %g is not a bin index in 0 .. %d", (long long)i, x[i], nc - 1);
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    void *d_t, *d_x, *d_p, *d_th, *d_work;
    const int64_t wb = phase_stat_work_bytes(kind, n, n_periods, nb < 1 ? 1 : nb, nc < 1 ? 1 : nc);
    PDC_TRY(cached(device, SLOT_WORK, wb > 0 ? wb : 0, &d_work));
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    d_x = nullptr;   // (Gregory-Loredo bins arrival times only: no values, no buffer, no statistics pass over them)
    if (x) PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_x));
    PDC_TRY(cached(device, SLOT_IN2, n_periods * 8, &d_p));
    PDC_TRY(cached(device, SLOT_OUT0, n_periods * 8, &d_th));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    if (x) PDC_HIP(hipMemcpyAsync(d_x, x, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_p, periods, n_periods * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(phase_stat_dev(kind, device, st, (double *)d_t, (double *)d_x, n, (double *)d_p, n_periods, nb, nc,
                           sigma, (double *)d_th, d_work, wb > 0 ? wb : 0));
    PDC_HIP(hipMemcpyAsync(out, d_th, n_periods * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

}  // namespace

int pdc_aov_scan(const double *t, const double *x, int64_t n, const double *periods, int64_t n_periods,
                 int n_bins, double *theta_out, int device) {
    return phase_stat_host(1, t, x, n, periods, n_periods, n_bins, 1, 1.0, theta_out, device);
}

int pdc_cond_entropy_scan(const double *t, const double *mag_bin, int64_t n, const double *periods,
                          int64_t n_periods, int n_phase, int n_mag, double *entropy_out, int device) {
    return phase_stat_host(2, t, mag_bin, n, periods, n_periods, n_phase, n_mag, 1.0, entropy_out, device);
}

int pdc_gl_scan_dev(int device, void *stream, const double *d_t, int64_t n, const double *d_periods,
                    int64_t n_periods, int m, int n_offsets, double *d_log_s) {
    PDC_REQUIRE(m >= 1 && n_offsets >= 1 && (int64_t)m * n_offsets <= 190, "gregory_loredo: m * n_offsets must be 1..190");
    return phase_stat_dev(4, device, (hipStream_t)stream, d_t, nullptr, n, d_periods, n_periods, m * n_offsets, m, 1.0,
                          d_log_s, nullptr, 0);
}

int pdc_gl_scan(const double *t, int64_t n, const double *periods, int64_t n_periods, int m, int n_offsets,
                double *log_s_out, int device) {
    PDC_REQUIRE(m >= 1 && n_offsets >= 1 && (int64_t)m * n_offsets <= 190, "gregory_loredo: m * n_offsets must be 1..190");
    return phase_stat_host(4, t, nullptr, n, periods, n_periods, m * n_offsets, m, 1.0, log_s_out, device);
}

int pdc_pdm_scan(const double *t, const double *x, int64_t n, const double *periods,
                 int64_t n_periods, int nb, int nc, double sigma, double *theta_out, int device) {
    return phase_stat_host(0, t, x, n, periods, n_periods, nb, nc, sigma, theta_out, device);
}

}  // extern "C"
