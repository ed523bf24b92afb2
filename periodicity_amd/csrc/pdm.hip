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
// Bin membership must be bit-identical to numpy's: phi = (t/period) % 1 with an IEEE division
// and Python modulo, compared against the doubles k/m0.  The fast path uses t * (1/period) and
// accepts its bin only when phi is provably farther from every edge than the worst-case error of
// that shortcut; otherwise (about 1e-9 of the pairs) the lane redoes the sample with the exact
// division and explicit edge comparisons.
#include "pdc_internal.h"

using namespace pdc;

namespace {

constexpr int kChunk = 128;

struct PdmArgs {
    const double *t, *x, *periods;
    int64_t n, n_periods;
    int nb, nc;
    double sigma;
    double *theta;
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

// SPLIT waves of a workgroup share one group of 64 trial periods (lane = period) and split every
// staged chunk of samples between them, so that a grid of ~1e5 periods still puts several waves on
// every SIMD (the loop is a chain of LDS read -> ALU -> LDS atomic: it needs wave-level parallelism
// to hide latency).  Each thread keeps its own private histogram; the SPLIT partial histograms of
// a period are summed in a fixed order at the end.
template <int BLOCK, int SPLIT>
__global__ __launch_bounds__(BLOCK) void pdm_scan_kernel(PdmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int m0 = a.nb * a.nc;
    const int nbins = m0 + 1;  // + overflow bin for phi == 1.0
    double2 *stage = reinterpret_cast<double2 *>(lds_raw);                  // [kChunk] (t, x')
    // the staging area doubles as the q_over/q_nan exchange ([2][BLOCK] doubles) at the end
    constexpr int kStage = kChunk > BLOCK ? kChunk : BLOCK;
    double *hsum = reinterpret_cast<double *>(stage + kStage);              // [nbins][BLOCK]
    unsigned *hcnt = reinterpret_cast<unsigned *>(hsum + (size_t)nbins * BLOCK);  // [nbins][BLOCK]
    double *edge = reinterpret_cast<double *>(hcnt + (size_t)nbins * BLOCK);      // [m0 + 2]
    double *red = edge + m0 + 2;                                                  // [BLOCK/64]
    const int tid = threadIdx.x;

    for (int k = tid; k < m0 + 2; k += BLOCK) edge[k] = (double)k / (double)m0;  // Python's k / m0
    for (int k = 0; k < nbins; ++k) {
        hsum[k * BLOCK + tid] = 0.0;
        hcnt[k * BLOCK + tid] = 0u;
    }

    // mean of x, total sum of squares about it, and max |t| (identical in every workgroup)
    double acc = 0.0, tmax = 0.0;
    for (int64_t i = tid; i < a.n; i += BLOCK) {
        acc += a.x[i];
        const double at = __builtin_fabs(a.t[i]);
        tmax = at > tmax ? at : tmax;
    }
    const double mean = block_reduce<BLOCK>(acc, red, false) / (double)a.n;
    tmax = block_reduce<BLOCK>(tmax, red, true);
    acc = 0.0;
    for (int64_t i = tid; i < a.n; i += BLOCK) {
        const double d = a.x[i] - mean;
        acc += d * d;
    }
    const double q_total = block_reduce<BLOCK>(acc, red, false);

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

    for (int64_t base = 0; base < a.n; base += kChunk) {
        __syncthreads();
        for (int i = tid; i < kChunk; i += BLOCK) {
            const int64_t g = base + i;
            stage[i] = g < a.n ? make_double2(a.t[g], a.x[g] - mean) : make_double2(0.0, 0.0);
        }
        __syncthreads();
        const int cnt = (int)((a.n - base) < kChunk ? (a.n - base) : kChunk);
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
            atomicAdd(&hsum[k * BLOCK + tid], val);
            atomicAdd(&hcnt[k * BLOCK + tid], inc);
        };
        auto update = [&](const double2 tx) {
            int k;
            double val = tx.y;
            unsigned inc = 1u;
            if (!fast_bin(tx.x, k)) exact_bin(tx, k, val, inc);
            add(k, val, inc);
        };
        int i = part * (kChunk / SPLIT);
        // the pair of the next trip is read before this trip's histogram atomics go out (the compiler
        // cannot move an LDS read above a possibly aliasing LDS atomic by itself); the staging area
        // is followed by the histograms, so reading one pair past the chunk is harmless
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
                for (int k = 0; k < nbins; ++k) {
                    hsum[k * BLOCK + tid] += hsum[k * BLOCK + other];
                    hcnt[k * BLOCK + tid] += hcnt[k * BLOCK + other];
                }
                q_over += qx[other];
                q_nan += qx[BLOCK + other];
            }
        }
    }
    if (part != 0 || pidx >= a.n_periods) return;
    // covers: phase.py:137-147
    double num = (double)a.nc * (q_total - q_nan) - q_over;
    long long n_sum = 0;
    int good = 0;
    for (int k = 0; k < m0; ++k) {
        double s = 0.0;
        long long c = 0;
        for (int j = 0; j < a.nc; ++j) {
            int b = k + j;
            if (b >= m0) {
                if (b == m0) {  // [1.0, (m0+1)/m0): only phi == 1.0 can live here
                    s += hsum[m0 * BLOCK + tid];
                    c += hcnt[m0 * BLOCK + tid];
                }
                b -= m0;
            }
            s += hsum[b * BLOCK + tid];
            c += hcnt[b * BLOCK + tid];
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
    a.theta[pidx] = good == 0 ? __builtin_nan("") : (num / (double)(n_sum - good)) / a.sigma;
}

size_t lds_bytes(int m0, int block) {
    const size_t stage = (size_t)(kChunk > block ? kChunk : block) * 16;
    return stage + (size_t)(m0 + 1) * block * 12 + (size_t)(m0 + 2) * 8 + 64;
}

}  // namespace

extern "C" {

int pdc_pdm_scan_dev(int device, void *stream, const double *d_t, const double *d_x, int64_t n,
                     const double *d_periods, int64_t n_periods, int nb, int nc, double sigma,
                     double *d_theta) {
    PDC_REQUIRE(d_t && d_x && (d_periods || n_periods == 0) && (d_theta || n_periods == 0),
                "pdm: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "pdm: negative size");
    PDC_REQUIRE(nb >= 1 && nc >= 1 && nc <= nb * nc, "pdm: nb and nc must be positive");
    PDC_REQUIRE((int64_t)nb * nc <= 190, "pdm: nb*nc = %lld exceeds the 190 bins that fit in LDS",
                (long long)nb * nc);
    if (n_periods == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    PdmArgs a{d_t, d_x, d_periods, n, n_periods, nb, nc, sigma, d_theta};
    const int m0 = nb * nc;
    hipStream_t st = (hipStream_t)stream;
    if (lds_bytes(m0, 256) <= 150 * 1024) {
        const size_t lds = lds_bytes(m0, 256);
        // waves = ceil(P/64) * SPLIT; aim at >= 4 waves per SIMD (4096 on the chip)
        const int64_t groups = (n_periods + 63) / 64;
        const int split = groups >= 4096 ? 1 : (groups >= 2048 ? 2 : 4);
        auto launch = [&](auto kernel, int periods_per_block) -> int {
            PDC_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
            hipLaunchKernelGGL(kernel, dim3((unsigned)((n_periods + periods_per_block - 1) / periods_per_block)),
                               dim3(256), lds, st, a);
            return PDC_OK;
        };
        if (split == 4) {
            PDC_TRY(launch(pdm_scan_kernel<256, 4>, 64));
        } else if (split == 2) {
            PDC_TRY(launch(pdm_scan_kernel<256, 2>, 128));
        } else {
            PDC_TRY(launch(pdm_scan_kernel<256, 1>, 256));
        }
    } else {
        const size_t lds = lds_bytes(m0, 64);
        PDC_HIP(hipFuncSetAttribute((const void *)pdm_scan_kernel<64, 1>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((pdm_scan_kernel<64, 1>), dim3((unsigned)((n_periods + 63) / 64)), dim3(64), lds,
                           st, a);
    }
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}

int pdc_pdm_scan(const double *t, const double *x, int64_t n, const double *periods,
                 int64_t n_periods, int nb, int nc, double sigma, double *theta_out, int device) {
    PDC_REQUIRE(t && x && (periods || n_periods == 0) && (theta_out || n_periods == 0),
                "pdm: NULL argument");
    PDC_REQUIRE(n >= 0 && n_periods >= 0, "pdm: negative size");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    void *d_t, *d_x, *d_p, *d_th;
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_x));
    PDC_TRY(cached(device, SLOT_IN2, n_periods * 8, &d_p));
    PDC_TRY(cached(device, SLOT_OUT0, n_periods * 8, &d_th));
    hipStream_t st = nullptr;
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_x, x, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_p, periods, n_periods * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(pdc_pdm_scan_dev(device, st, (double *)d_t, (double *)d_x, n, (double *)d_p, n_periods, nb,
                             nc, sigma, (double *)d_th));
    PDC_HIP(hipMemcpyAsync(theta_out, d_th, n_periods * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

}  // extern "C"
