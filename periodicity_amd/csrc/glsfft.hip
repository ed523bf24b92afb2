// Generalized Lomb-Scargle by the reference's OWN algorithm on the device (SURVEY.md §8 f1,
// "Tier F"): Press-Rybicki extirpolation + inverse FFT, exactly as
// /root/reference/src/periodicity/spectral.py:11-40 (`_trig_sum`) does it on the CPU, followed by
// the same epilogue (spectral.py:113-132).  O(N + nfft log nfft) instead of O(N nf); it inherits
// the reference's approximation error (median ~1e-3 relative vs the exact sums) and therefore
// reproduces `power_ref` itself, where the direct-sum kernel (gls.hip) reproduces the exact sums.
//
// Pipeline (all on one stream, nothing but power[nf] leaves the device):
//   glsfft_prep_{a,c}      weights, centring, YY, tmin (grid-wide, fixed-order partials)  (:99-108,120)
//   glsfft_spread_kernel   one thread per sample: weights pre-rotated to fmin about tmin (:19-20),
//                          grid position (:21), whole hits deposited (:22-24), the rest spread over
//                          four neighbours with cubic Lagrange weights (:25-33) — fp64 global
//                          atomics into 2 or 3 complex grids (w*y @ df, w @ 2df, w @ df)
//   fft_pass_kernel<R>     Stockham autosort radix-16/8/4/2 passes, out of place, ping-pong between
//                          the grid and one scratch buffer; every pass streams the array once
//                          (HBM-bound: 32 B per point per pass), twiddles from the cycle-domain
//                          sincos (no tables, no recurrences)                               (:34)
//   glsfft_epilogue_kernel undo the tmin shift (:35-37), scale by nfft (:38-39), epilogue
#include "pdc_internal.h"
#include "gls_epilogue.h"

#include <cstdlib>
#include <vector>

using namespace pdc;

namespace {

constexpr int kBlock = 256;

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx operator+(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx operator-(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// ---- in-register DFT of R = 2^m points, e^{+2 pi i jk/R} (inverse direction), natural order -------
template <int R, int K>
__device__ __forceinline__ cplx twiddle_const() {
    // exp(+2 pi i K / R) for R <= 16, K < R/2
    constexpr int q = K * (16 / R);  // sixteenths of a turn, 0..7
    constexpr double c[8] = {1.0, 0.92387953251128675613, 0.70710678118654752440,
                             0.38268343236508977173, 0.0, -0.38268343236508977173,
                             -0.70710678118654752440, -0.92387953251128675613};
    constexpr double s[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440,
                             0.92387953251128675613, 1.0, 0.92387953251128675613,
                             0.70710678118654752440, 0.38268343236508977173};
    return {c[q], s[q]};
}

template <int R>
struct Dft {
    template <int K>
    static __device__ __forceinline__ void combine(cplx *x, const cplx *e, const cplx *o) {
        if constexpr (K < R / 2) {
            cplx t;
            if constexpr (K == 0) {
                t = o[0];
            } else if constexpr (4 * K == R) {
                t = {-o[K].im, o[K].re};  // times +i
            } else {
                t = cmul(twiddle_const<R, K>(), o[K]);
            }
            x[K] = e[K] + t;
            x[K + R / 2] = e[K] - t;
            combine<K + 1>(x, e, o);
        }
    }
    static __device__ __forceinline__ void run(cplx *x) {
        cplx e[R / 2], o[R / 2];
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            e[i] = x[2 * i];
            o[i] = x[2 * i + 1];
        }
        Dft<R / 2>::run(e);
        Dft<R / 2>::run(o);
        combine<0>(x, e, o);
    }
};
template <>
struct Dft<1> {
    static __device__ __forceinline__ void run(cplx *) {}
};

// The extirpolated grids are empty beyond the last sample's position (80 % of the cells at the default
// five samples per peak): the deposit kernel does not write those cells and the FIRST pass does not read
// them.  ord = {tmin, tmax, sorted} of the curve (see glsfft_deposit_kernel); nullptr: every cell is live.
struct LiveArgs {
    const double *ord;
    int ord_stride, ngrid, g_first;
    double df;
};

__device__ __forceinline__ bool deposit_in_order(const double *ord, double nfftd, double dfg) {
    const double span = ord[1] - ord[0];   // (NaN or -inf when no time stamp is a number: the atomic kernels then)
    return ord[2] != 0.0 && span >= 0.0 && (span * nfftd) * dfg < nfftd;
}

// cells [0, live) of transform `y` of the launch may be non-zero
__device__ __forceinline__ int64_t live_cells(const LiveArgs &l, int64_t y, int64_t nfft) {
    if (!l.ord) return nfft;
    const int64_t curve = y / l.ngrid;
    const int g = l.g_first + (int)(y - curve * l.ngrid);
    const double *ord = l.ord + curve * l.ord_stride;
    const double nfftd = (double)nfft, dfg = g == 1 ? 2.0 * l.df : l.df;
    if (!deposit_in_order(ord, nfftd, dfg)) return nfft;
    const int64_t live = (int64_t)(((ord[1] - ord[0]) * nfftd) * dfg) + 4;   // last position + its neighbours
    return live < nfft ? live : nfft;
}

// One Stockham pass: N points, sub-transforms of length Ns become length Ns*R.
template <int R>
__global__ __launch_bounds__(kBlock) void fft_pass_kernel(const cplx *__restrict__ in,
                                                          cplx *__restrict__ out, int64_t N,
                                                          int64_t Ns, int64_t keep, LiveArgs lv) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t T = N / R;
    if (j >= T) return;
    const int64_t live = live_cells(lv, blockIdx.y, N);
    in += (int64_t)blockIdx.y * N;   // batch of independent transforms, one per blockIdx.y
    out += (int64_t)blockIdx.y * N;
    const int64_t k = j & (Ns - 1);
    cplx v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = j + r * T < live ? in[j + r * T] : cplx{0.0, 0.0};
    const double inv = 1.0 / (double)(Ns * R);  // power of two: exact
#pragma unroll
    for (int r = 1; r < R; ++r) {
        double s, c;
        sincos_cycles((double)(r * k) * inv, s, c);  // e^{+2 pi i r k / (Ns R)}
        v[r] = cmul(v[r], cplx{c, s});
    }
    Dft<R>::run(v);
    const int64_t j0 = (j - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (j0 + r * Ns < keep) out[j0 + r * Ns] = v[r];  // last pass: only the first `keep` outputs
}


// One Stockham pass of radix R = 16 x RB (RB = 16, 8, 4, 2) through LDS: one HBM trip does a radix-16
// stage and a radix-RB stage.  A workgroup of 256 threads owns JT = 256/RB consecutive butterflies j
// (each an R-point DFT over the inputs in[j + r*T], r = r1 + RB*r2, r1 < RB, r2 < 16):
//   stage A  thread (jj, r1): outer twiddles, 16-point DFT over r2, inner twiddle W_R^(r1*s2) -> LDS
//   stage B  (jj, s2) pairs, 16/RB per thread: RB-point DFT over r1 -> Y[s2 + 16 s1]
//            -> out[(j-k)*R + k + s*Ns]
// Global accesses are runs of >= 16 consecutive complex numbers (256 B).  Requires N >= 4096.
// (Round 6, measured and withdrawn - profiles/r06_fft_halves_experiment.patch: the exchange between the stages one
// component at a time - real parts, then imaginary parts, 37 KB of LDS instead of 74, three to four workgroups per CU
// instead of two, two more barriers: C2's FFT path 0.599 against 0.538 ms on the same box.  The passes are not short of
// resident waves.)
template <int RB>
__global__ __launch_bounds__(256) void fft_pass_lds_kernel(const cplx *__restrict__ in,
                                                            cplx *__restrict__ out, int64_t N,
                                                            int64_t Ns, int64_t keep, LiveArgs lv) {
    constexpr int R = 16 * RB;
    constexpr int JT = 256 / RB;          // butterflies per workgroup
    // Padded LDS row per butterfly: ONE complex number (16 B) of padding.  Round 6: it was R + 16 - 256 B, a whole number
    // of LDS bank rows (32 banks x 4 B), i.e. no skew at all: in stage A every lane of a store instruction (fixed s2)
    // hit the same four banks, a 64-way serialisation, and stage B's loads spread over half of them; the PMC counters
    // had it (SQ_LDS_BANK_CONFLICT 9.2e6 of 11.3e6 LDS-active cycles per pass, 18 us of a pass's 45 inside LDS).  With
    // R + 1 the 16-byte slot of (jj, r1, s2) is (jj + s2) mod 8: eight lanes per slot, the minimum for 64 x 16 B.
    constexpr int ROW = R + 1;
    __shared__ cplx tile[JT * ROW];       // [jj][r1][s2]
    __shared__ cplx wR[R];
    const int tid = threadIdx.x;
    const int64_t live = live_cells(lv, blockIdx.y, N);
    in += (int64_t)blockIdx.y * N;
    out += (int64_t)blockIdx.y * N;
    if (tid < R) {
        double s, c;
        sincos_cycles((double)tid * (1.0 / (double)R), s, c);
        wR[tid] = cplx{c, s};
    }
    const int64_t T = N / R;
    const int64_t j_base = (int64_t)blockIdx.x * JT;
    // ---- stage A -------------------------------------------------------------------------------
    {
        const int jj = tid % JT, r1 = tid / JT;
        const int64_t j = j_base + jj;
        const int64_t k = j & (Ns - 1);
        cplx v[16];
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) {
            const int64_t at = j + (int64_t)(r1 + RB * r2) * T;
            v[r2] = at < live ? in[at] : cplx{0.0, 0.0};
        }
        if (Ns > 1) {
            // outer twiddle e^{2 pi i r k / (R Ns)}, r = r1 + RB r2: tw(r1) * tw(RB)^r2
            const double inv = 1.0 / (double)(Ns * R);
            double s, c;
            sincos_cycles((double)((int64_t)r1 * k) * inv, s, c);
            cplx tw = cplx{c, s};
            sincos_cycles((double)((int64_t)RB * k) * inv, s, c);
            const cplx step = cplx{c, s};
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                v[r2] = cmul(v[r2], tw);
                tw = cmul(tw, step);
            }
        }
        Dft<16>::run(v);
        __syncthreads();  // wR ready
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)
            tile[jj * ROW + r1 * 16 + s2] = cmul(v[s2], wR[(r1 * s2) % R]);
    }
    __syncthreads();
    // ---- stage B -------------------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < 16 / RB; ++it) {
        const int pair = tid + it * 256;  // (jj, s2) pairs, JT * 16 of them
        // lanes run over jj when consecutive j write consecutive addresses (Ns >= 16), over s2 when
        // the R outputs of one j are contiguous (first pass, Ns == 1)
        const int jj = Ns > 1 ? pair % JT : pair / 16;
        const int s2 = Ns > 1 ? pair / JT : pair % 16;
        const int64_t j = j_base + jj;
        const int64_t k = j & (Ns - 1);
        cplx v[RB];
#pragma unroll
        for (int r1 = 0; r1 < RB; ++r1) v[r1] = tile[jj * ROW + r1 * 16 + s2];
        Dft<RB>::run(v);
        const int64_t j0 = (j - k) * R + k;
#pragma unroll
        for (int s1 = 0; s1 < RB; ++s1) {
            const int64_t o = j0 + (int64_t)(s2 + 16 * s1) * Ns;
            if (o < keep) out[o] = v[s1];  // last pass: only the first `keep` outputs are ever read
        }
    }
}

// ---- prologue: spectral.py:99-108, 120 -- two grid-wide kernels, deterministic partials ------------
// A: per-workgroup partial sums of err^-2 and err^-2 * y and the partial minimum of t.
// C: every workgroup re-reduces A's partials (<= kMaxPart, fixed order -> identical in all of them),
//    writes w and w*y and its partial of YY = sum w y^2.  Consumers reduce the YY partials the same way.
constexpr int kMaxPart = 512;

struct FftPrepArgs {
    const double *t, *y, *dy;
    int64_t n;
    int fit_mean, nparts;
    double *wy, *w;
    double *part;   // [5][kMaxPart]: sum err^-2 | sum err^-2 y | min t | -max t | out-of-order pairs
    double *ypart;  // [kMaxPart]: partial YY
    double *scal;   // {YY (filled by the epilogue's reduction), Werr, tmin, tmax, sorted}
};

__device__ __forceinline__ double block_min_256(double v, double *red) {
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_down(v, o, 64);
        v = u < v ? u : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < kBlock / 64; ++w) r = red[w] < r ? red[w] : r;
    return r;
}

__global__ __launch_bounds__(kBlock) void glsfft_prep_a_kernel(FftPrepArgs a) {
    __shared__ double red[kBlock / 64];
    double sw = 0.0, swy = 0.0, tmin = __builtin_inf(), ntmax = __builtin_inf(), disorder = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        const double e = a.dy ? a.dy[i] : 1.0;
        const double wr = 1.0 / (e * e);
        sw += wr;
        swy += wr * a.y[i];
        const double ti = a.t[i];
        tmin = ti < tmin ? ti : tmin;
        ntmax = -ti < ntmax ? -ti : ntmax;
        if (i > 0 && !(ti >= a.t[i - 1])) disorder = 1.0;   // (a NaN counts as disorder)
    }
    sw = block_sum<kBlock>(sw, red);
    swy = block_sum<kBlock>(swy, red);
    tmin = block_min_256(tmin, red);
    ntmax = block_min_256(ntmax, red);
    disorder = block_sum<kBlock>(disorder, red);
    if (threadIdx.x == 0) {
        a.part[blockIdx.x] = sw;
        a.part[kMaxPart + blockIdx.x] = swy;
        a.part[2 * kMaxPart + blockIdx.x] = tmin;
        a.part[3 * kMaxPart + blockIdx.x] = ntmax;
        a.part[4 * kMaxPart + blockIdx.x] = disorder;
    }
}

// fixed-order reduction of `count` partials by one workgroup (identical result in every workgroup)
__device__ __forceinline__ double reduce_partials(const double *p, int count, double *red) {
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += kBlock) v += p[i];
    return block_sum<kBlock>(v, red);
}

__global__ __launch_bounds__(kBlock) void glsfft_prep_c_kernel(FftPrepArgs a) {
    __shared__ double red[kBlock / 64];
    const double W = reduce_partials(a.part, a.nparts, red);
    const double ybar = a.fit_mean ? reduce_partials(a.part + kMaxPart, a.nparts, red) / W : 0.0;
    double tmin = __builtin_inf();
    for (int i = threadIdx.x; i < a.nparts; i += kBlock) {
        const double u = a.part[2 * kMaxPart + i];
        tmin = u < tmin ? u : tmin;
    }
    tmin = block_min_256(tmin, red);
    double ntmax = __builtin_inf();
    for (int i = threadIdx.x; i < a.nparts; i += kBlock) {
        const double u = a.part[3 * kMaxPart + i];
        ntmax = u < ntmax ? u : ntmax;
    }
    ntmax = block_min_256(ntmax, red);
    const double disorder = reduce_partials(a.part + 4 * kMaxPart, a.nparts, red);
    double yy = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * kBlock) {
        const double e = a.dy ? a.dy[i] : 1.0;
        const double w = (1.0 / (e * e)) / W;
        const double yc = a.y[i] - ybar;
        a.w[i] = w;
        a.wy[i] = w * yc;
        yy += w * yc * yc;
    }
    yy = block_sum<kBlock>(yy, red);
    if (threadIdx.x == 0) {
        a.ypart[blockIdx.x] = yy;
        if (blockIdx.x == 0) {
            a.scal[1] = W;
            a.scal[2] = tmin;
            a.scal[3] = -ntmax;                        // latest time stamp
            a.scal[4] = disorder == 0.0 ? 1.0 : 0.0;   // time stamps in non-decreasing order
        }
    }
}

// ---- extirpolation: spectral.py:18-33 ---------------------------------------------------------------
struct SpreadArgs {
    const double *t, *h;   // h = weights of this transform
    const double *scal;    // scal[2..4] = {tmin, tmax, sorted}
    double tmin_value;     // (unused)
    int64_t n, nfft;
    double df, fmin;
    double *grid;          // [nfft] complex, zeroed
};

__device__ __forceinline__ void grid_add(double *grid, int64_t idx, double re, double im) {
    unsafeAtomicAdd(grid + 2 * idx, re);
    unsafeAtomicAdd(grid + 2 * idx + 1, im);
}

__global__ __launch_bounds__(kBlock) void glsfft_spread_kernel(SpreadArgs a) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const double tmin = a.scal[2];
    if (deposit_in_order(a.scal + 2, (double)a.nfft, a.df)) return;   // glsfft_deposit_kernel has done it
    const double dt = a.t[i] - tmin;
    // w * np.exp(2j * np.pi * fmin * (t - tmin)): the phase is fl(fl(2 pi fmin) * dt) radians
    const double ang = (6.283185307179586 * a.fmin) * dt;
    double sn, cs;
    sincos(ang, &sn, &cs);
    const double hre = a.h[i] * cs, him = a.h[i] * sn;
    // tnorm = ((t - tmin) * nfft * df) % nfft
    const double nfftd = (double)a.nfft;
    double tn = fmod((dt * nfftd) * a.df, nfftd);
    if (tn != 0.0 && tn < 0.0) tn += nfftd;
    if (tn - __builtin_floor(tn) == 0.0) {  // tnorm % 1 == 0: deposit whole
        grid_add(a.grid, (int64_t)tn, hre, him);
        return;
    }
    int64_t ilo = (int64_t)(tn - 2.0);  // astype(int): truncation toward zero
    ilo = ilo < 0 ? 0 : (ilo > a.nfft - 4 ? a.nfft - 4 : ilo);
    const double x = tn - (double)ilo;
    const double prod = ((x * (x - 1.0)) * (x - 2.0)) * (x - 3.0);
    const double nre = hre * prod, nim = him * prod;
    const double den[4] = {6.0, -2.0, 2.0, -6.0};  // the reference's running denominator
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t ind = ilo + (3 - j);
        const double d = den[j] * (tn - (double)ind);
        grid_add(a.grid, ind, nre / d, nim / d);
    }
}

__device__ __forceinline__ void spread_one(double *grid, int64_t nfft, double dt, double h, double df,
                                           double fmin, const double *ord) {
    if (deposit_in_order(ord, (double)nfft, df)) return;   // glsfft_deposit_kernel has done this grid
    const double ang = (6.283185307179586 * fmin) * dt;
    double sn, cs;
    sincos(ang, &sn, &cs);
    const double hre = h * cs, him = h * sn;
    const double nfftd = (double)nfft;
    double tn = fmod((dt * nfftd) * df, nfftd);
    if (tn != 0.0 && tn < 0.0) tn += nfftd;
    if (tn - __builtin_floor(tn) == 0.0) {
        grid_add(grid, (int64_t)tn, hre, him);
        return;
    }
    int64_t ilo = (int64_t)(tn - 2.0);
    ilo = ilo < 0 ? 0 : (ilo > nfft - 4 ? nfft - 4 : ilo);
    const double x = tn - (double)ilo;
    const double prod = ((x * (x - 1.0)) * (x - 2.0)) * (x - 3.0);
    const double nre = hre * prod, nim = him * prod;
    const double den[4] = {6.0, -2.0, 2.0, -6.0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t ind = ilo + (3 - j);
        const double d = den[j] * (tn - (double)ind);
        grid_add(grid, ind, nre / d, nim / d);
    }
}

// ---- the same deposits without atomics, in a fixed order ---------------------------------------------
// With time stamps in non-decreasing order and a grid that does not wrap ((tmax - tmin) nfft df < nfft -
// always so for the default n >= 1 samples per peak), grid positions rise with the sample index, so the
// samples that touch a cell are consecutive.  A thread owns groups of four consecutive cells: a search
// finds the first sample that can reach them, the deposits are added in sample order and stored once -
// bitwise reproducible (the atomic kernels above add in arrival order), and the memset goes away: every
// cell up to the last sample's reach is written (the empty ones with zero), the rest is read by nobody.
// Otherwise this kernel only zeroes the grid and the atomic kernels do the work (`deposit_in_order` is
// false for them exactly then).
struct DepositArgs {
    const double *t, *h0, *h1;   // h0: weights of grid 0 (w y); h1: weights of grids 1 (@ 2 df) and 2 (@ df)
    const int64_t *offsets;      // per-curve sample ranges; nullptr: one curve of n samples
    int64_t n;
    int shared_t, ngrid;         // ngrid = grids per curve in `grids` (3 with fit_mean, else 2)
    int slot_first, nslots;      // slots of this launch: slot 0 = grid 0 (and grid 2, which shares its
                                 // positions and phases, when ngrid == 3); slot 1 = grid 1
    const double *ord;           // ord[curve * ord_stride + {0, 1, 2}] = {tmin, tmax, sorted}
    int ord_stride;
    int64_t nfft;
    double df, fmin;
    double *grids;               // [curve][ngrid][nfft] complex: every live cell is written
};

constexpr int kDepCells = 4;                     // cells per thread
#ifndef PDC_DEP_ITERS
#define PDC_DEP_ITERS 4
#endif
constexpr int kDepIters = PDC_DEP_ITERS;         // groups of cells per thread: the workgroup's search is shared
constexpr int kDepSpan = kDepCells * kBlock * kDepIters;   // cells per workgroup
constexpr int kDepStage = 768;                   // samples of a workgroup's range kept in LDS: position, pre-rotation, weights

__global__ __launch_bounds__(kBlock) void glsfft_deposit_kernel(DepositArgs a) {
    __shared__ double s_pos[kDepStage];
    // round 6: the samples themselves too, with their pre-rotation exp(2 pi i fmin (t - tmin)) evaluated ONCE, by the thread
    // that stages them (the deposit loop waited for three dependent global loads per sample and ran a sincos for every
    // (sample, group of cells) visit: 30 us of VALU issue in a 73-us launch)
    __shared__ double s_cs[kDepStage], s_sn[kDepStage], s_h[kDepStage], s_h2[kDepStage];
    __shared__ unsigned long long s_ballot[2][kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t curve = blockIdx.y / a.nslots;
    const int g = a.slot_first + (int)(blockIdx.y - curve * a.nslots);
    const bool twin = g == 0 && a.ngrid == 3;
    const int64_t G0 = (int64_t)blockIdx.x * kDepSpan;
    // (cells from `live` on are never read: the first FFT pass takes them as zero)
    const int64_t live = live_cells(LiveArgs{a.ord, a.ord_stride, 1, g, a.df}, curve, a.nfft);
    if (G0 >= live) return;   // (workgroup-uniform)
    const int64_t off = a.offsets ? a.offsets[curve] : 0;
    const int64_t n = a.offsets ? a.offsets[curve + 1] - off : a.n;
    const double *t = a.t + (a.shared_t ? 0 : off);
    const double *h = (g == 0 ? a.h0 : a.h1) + off, *h2 = a.h1 + off;
    const double *ord = a.ord + curve * a.ord_stride;
    const double dfg = g == 1 ? 2.0 * a.df : a.df, fming = g == 1 ? 2.0 * a.fmin : a.fmin;
    const double nfftd = (double)a.nfft, tmin = ord[0];
    auto position = [&](int64_t i) -> double {   // tnorm of sample i; used only when the grid does not wrap,
        return ((t[i] - tmin) * nfftd) * dfg;    // where the reference's "% nfft" (fmod) returns its argument
    };
    const bool ordered = n > 0 && deposit_in_order(ord, nfftd, dfg);   // (workgroup-uniform)
    int64_t w_first = 0, w_last = 0;
    bool staged = false;
    if (ordered) {
        // The samples that can reach this workgroup's cells, [w_first, w_last): first sample at or beyond
        // G0 - 4 and first at or beyond G0 + span + 4, by a 256-ary search made by the whole workgroup
        // (three rounds for 1e5 samples instead of 17 dependent loads per thread).
        const double bound[2] = {(double)(G0 - 4), (double)(G0 + kDepSpan + 4)};
        int64_t lo[2] = {0, 0}, hi[2] = {n, n};   // the answer lies in [lo, hi]
        for (;;) {
            const int64_t w0 = hi[0] - lo[0], w1 = hi[1] - lo[1];
            if (w0 <= 0 && w1 <= 0) break;        // (workgroup-uniform)
            bool pred[2];
            int64_t step[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t w = hi[q] - lo[q];
                step[q] = (w + kBlock - 1) / kBlock;
                const int64_t at = lo[q] + (int64_t)tid * step[q];
                pred[q] = w > 0 && at < hi[q] && position(at) >= bound[q];
            }
            __syncthreads();
            const unsigned long long b0 = __ballot(pred[0]), b1 = __ballot(pred[1]);
            if (lane == 0) {
                s_ballot[0][wave] = b0;
                s_ballot[1][wave] = b1;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (hi[q] - lo[q] <= 0) continue;
                int first = kBlock;               // first probe at or beyond the bound
                for (int w = kBlock / 64 - 1; w >= 0; --w)
                    if (s_ballot[q][w]) first = w * 64 + __builtin_ctzll(s_ballot[q][w]);
                const int64_t probes = (hi[q] - lo[q] + step[q] - 1) / step[q];   // probes inside [lo, hi)
                const int64_t new_hi = first < probes ? lo[q] + (int64_t)first * step[q] : hi[q];
                const int64_t new_lo = first > 0 ? lo[q] + (int64_t)((first < probes ? first : probes) - 1) * step[q] + 1 : lo[q];
                lo[q] = new_lo < new_hi ? new_lo : new_hi;
                hi[q] = new_hi;
            }
        }
        w_first = lo[0];
        w_last = lo[1];
        staged = w_last - w_first <= kDepStage;   // (workgroup-uniform)
        if (staged)
            for (int64_t i = w_first + tid; i < w_last; i += kBlock) {
                s_pos[i - w_first] = position(i);
                double sn0, cs0;
                sincos((6.283185307179586 * fming) * (t[i] - tmin), &sn0, &cs0);
                s_cs[i - w_first] = cs0;
                s_sn[i - w_first] = sn0;
                s_h[i - w_first] = h[i];
                s_h2[i - w_first] = twin ? h2[i] : 0.0;
            }
        __syncthreads();
    }
    auto pos_at = [&](int64_t i) -> double { return staged ? s_pos[i - w_first] : position(i); };
    for (int it = 0; it < kDepIters; ++it) {
    const int64_t g0 = G0 + ((int64_t)it * kBlock + tid) * kDepCells;
    if (g0 >= live) break;
    double acc_re[kDepCells], acc_im[kDepCells], twin_re[kDepCells], twin_im[kDepCells];
#pragma unroll
    for (int c = 0; c < kDepCells; ++c) acc_re[c] = acc_im[c] = twin_re[c] = twin_im[c] = 0.0;
    if (ordered) {
        const double lo_t = (double)(g0 - 4), hi_t = (double)(g0 + kDepCells + 4);
        int64_t first = w_first, last = w_last;   // first sample of the workgroup's range at or beyond lo_t
        while (first < last) {
            const int64_t mid = (first + last) >> 1;
            if (pos_at(mid) >= lo_t) last = mid; else first = mid + 1;
        }
        for (int64_t i = first; i < w_last; ++i) {
            const double tn = pos_at(i);
            if (!(tn < hi_t)) break;
            int64_t ilo = 0;
            const bool whole = tn - __builtin_floor(tn) == 0.0;
            if (!whole) {
                ilo = (int64_t)(tn - 2.0);
                ilo = ilo < 0 ? 0 : (ilo > a.nfft - 4 ? a.nfft - 4 : ilo);
                if (ilo + 3 < g0 || ilo >= g0 + kDepCells) continue;
            } else if ((int64_t)tn < g0 || (int64_t)tn >= g0 + kDepCells) {
                continue;
            }
            const double hi_ = staged ? s_h[i - w_first] : h[i];
            const double h2i = !twin ? 0.0 : (staged ? s_h2[i - w_first] : h2[i]);
            double sn, cs;
            if (staged) {
                sn = s_sn[i - w_first];
                cs = s_cs[i - w_first];
            } else {
                sincos((6.283185307179586 * fming) * (t[i] - tmin), &sn, &cs);
            }
            const double hre = hi_ * cs, him = hi_ * sn;
            const double kre = twin ? h2i * cs : 0.0, kim = twin ? h2i * sn : 0.0;
            if (whole) {   // one deposit
                const int64_t ind = (int64_t)tn;
#pragma unroll
                for (int c = 0; c < kDepCells; ++c)
                    if (ind == g0 + c) {
                        acc_re[c] += hre;
                        acc_im[c] += him;
                        twin_re[c] += kre;
                        twin_im[c] += kim;
                    }
                continue;
            }
            const double x = tn - (double)ilo;
            const double prod = ((x * (x - 1.0)) * (x - 2.0)) * (x - 3.0);
            const double nre = hre * prod, nim = him * prod, mre = kre * prod, mim = kim * prod;
            const double den[4] = {6.0, -2.0, 2.0, -6.0};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t ind = ilo + (3 - j);
                if (ind < g0 || ind >= g0 + kDepCells) continue;
                const double d = den[j] * (tn - (double)ind);
                const double vre = nre / d, vim = nim / d;
                const double ure = twin ? mre / d : 0.0, uim = twin ? mim / d : 0.0;
#pragma unroll
                for (int c = 0; c < kDepCells; ++c)
                    if (ind == g0 + c) {
                        acc_re[c] += vre;
                        acc_im[c] += vim;
                        twin_re[c] += ure;
                        twin_im[c] += uim;
                    }
            }
        }
    }
    double *out = a.grids + 2 * ((curve * a.ngrid + g) * a.nfft + g0);
    double *out2 = a.grids + 2 * ((curve * a.ngrid + 2) * a.nfft + g0);
#pragma unroll
    for (int c = 0; c < kDepCells; ++c)
        if (g0 + c < a.nfft) {
            out[2 * c] = acc_re[c];
            out[2 * c + 1] = acc_im[c];
            if (twin) {
                out2[2 * c] = twin_re[c];
                out2[2 * c + 1] = twin_im[c];
            }
        }
    }
}

// All three (or two) grids of one curve in a single launch: (w*y @ df), (w @ 2 df), (w @ df).
struct Spread3Args {
    const double *t, *wy, *w, *scal;
    int64_t n, nfft;
    double df, fmin;
    double *grids;  // [ngrid][nfft] complex, zeroed
    int fit_mean;
};

__global__ __launch_bounds__(kBlock) void glsfft_spread3_kernel(Spread3Args a) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    const double dt = a.t[i] - a.scal[2];
    spread_one(a.grids, a.nfft, dt, a.wy[i], a.df, a.fmin, a.scal + 2);
    spread_one(a.grids + 2 * a.nfft, a.nfft, dt, a.w[i], 2.0 * a.df, 2.0 * a.fmin, a.scal + 2);   // (:110)
    if (a.fit_mean) spread_one(a.grids + 4 * a.nfft, a.nfft, dt, a.w[i], a.df, a.fmin, a.scal + 2);
}

// ---- epilogue: spectral.py:34-39 then :113-132 ----------------------------------------------------------
struct FftEpiArgs {
    const cplx *gh, *g2, *g1;  // transforms of (w y @ df), (w @ 2 df), (w @ df, may be null)
    const double *scal;
    const double *ypart;       // partial YY sums (nparts of them)
    int nparts;
    int64_t nfft, nf;
    double df, fmin;
    int fit_mean, psd;
    double *power;             // or raw outputs
    double *raw_s, *raw_c;
    double tmin_value;
    int raw;
};

// fftgrid *= exp(2j pi tmin f), f = fmin + df*arange(nf); C = nfft*re, S = nfft*im
__device__ __forceinline__ void finish_sum(cplx z, double tmin, double f, double nfftd, double &S,
                                           double &C) {
    if (tmin != 0.0) {
        const double ang = (6.283185307179586 * tmin) * f;
        double sn, cs;
        sincos(ang, &sn, &cs);
        z = cmul(z, cplx{cs, sn});
    }
    C = nfftd * z.re;
    S = nfftd * z.im;
}

__global__ __launch_bounds__(kBlock) void glsfft_epilogue_kernel(FftEpiArgs a) {
    __shared__ double red[kBlock / 64];
    const double YY = a.raw ? 0.0 : reduce_partials(a.ypart, a.nparts, red);
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= a.nf) return;
    const double nfftd = (double)a.nfft;
    // pocketfft's ifft divides by nfft and the reference multiplies it back: do both so that the
    // roundings match
    const double scale = 1.0 / nfftd;
    if (a.raw) {
        const double tmin = a.tmin_value;
        cplx z = a.gh[j];
        z.re *= scale;
        z.im *= scale;
        double S, C;
        finish_sum(z, tmin, a.fmin + a.df * (double)j, nfftd, S, C);
        a.raw_s[j] = S;
        a.raw_c[j] = C;
        return;
    }
    const double tmin = a.scal[2];
    const double f = a.fmin + a.df * (double)j;
    double Sh, Ch, S2, C2, S = 0.0, C = 0.0;
    cplx z = a.gh[j];
    z.re *= scale;
    z.im *= scale;
    finish_sum(z, tmin, f, nfftd, Sh, Ch);
    z = a.g2[j];
    z.re *= scale;
    z.im *= scale;
    finish_sum(z, tmin, 2.0 * a.fmin + (2.0 * a.df) * (double)j, nfftd, S2, C2);
    double p;
    if (a.fit_mean) {
        z = a.g1[j];
        z.re *= scale;
        z.im *= scale;
        finish_sum(z, tmin, f, nfftd, S, C);
        p = gls_power_from_sums<true>(Sh, Ch, S, C, S2, C2, YY, a.scal[1], a.psd);
    } else {
        p = gls_power_from_sums<false>(Sh, Ch, S, C, S2, C2, YY, a.scal[1], a.psd);
    }
    a.power[j] = p;
}


// ---- batched form: B light curves on one grid (bootstrap replicates share t) -----------------------
struct FftBatchArgs {
    const double *t, *y, *dy;
    const int64_t *offsets;  // [B + 1]
    int shared_t, fit_mean, psd, ngrid;
    int64_t nfft, nf;
    double df, fmin;
    double *w, *wy;   // [n_total]
    double *scal;     // [B][8] = {YY, Werr, tmin, tmax, sorted, -, -, -}
    cplx *grids;      // [B][ngrid][nfft]
    const cplx *result;  // where the transforms ended up (grids or scratch), same layout
    double *power;    // [B][nf]
    // bootstrap replicates by index (spectral.py:146-148): y, dy = ONE curve, sample i of replicate b is picks[b n + i]
    const int32_t *picks = nullptr;
};

__global__ __launch_bounds__(1024) void glsfft_prep_batch_kernel(FftBatchArgs a) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    const int64_t off = a.offsets[blockIdx.x], n = a.offsets[blockIdx.x + 1] - off;
    const double *t = a.shared_t ? a.t : a.t + off;
    const int32_t *pk = a.picks ? a.picks + off : nullptr;
    const double *y = pk ? a.y : a.y + off;
    const double *dy = a.dy ? (pk ? a.dy : a.dy + off) : nullptr;
    auto at = [pk](int64_t i) -> int64_t { return pk ? (int64_t)pk[i] : i; };
    double acc = 0.0, tmin = __builtin_inf(), ntmax = __builtin_inf(), disorder = 0.0;
    for (int64_t i = tid; i < n; i += 1024) {
        const double e = dy ? dy[at(i)] : 1.0;
        acc += 1.0 / (e * e);
        const double ti = t[i];
        tmin = ti < tmin ? ti : tmin;
        ntmax = -ti < ntmax ? -ti : ntmax;
        if (i > 0 && !(ti >= t[i - 1])) disorder = 1.0;   // (a NaN counts as disorder)
    }
    const double W = block_sum<1024>(acc, red);
    disorder = block_sum<1024>(disorder, red);
    auto block_min = [&](double v) -> double {
        for (int o = 32; o > 0; o >>= 1) {
            const double u = __shfl_down(v, o, 64);
            v = u < v ? u : v;
        }
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        v = red[0];
        for (int w = 1; w < 16; ++w) v = red[w] < v ? red[w] : v;
        __syncthreads();
        return v;
    };
    tmin = block_min(tmin);
    ntmax = block_min(ntmax);
    double ybar = 0.0;
    if (a.fit_mean) {
        acc = 0.0;
        for (int64_t i = tid; i < n; i += 1024) {
            const double e = dy ? dy[at(i)] : 1.0;
            acc += (1.0 / (e * e)) / W * y[at(i)];
        }
        ybar = block_sum<1024>(acc, red);
    }
    double yy = 0.0;
    for (int64_t i = tid; i < n; i += 1024) {
        const double e = dy ? dy[at(i)] : 1.0;
        const double w = (1.0 / (e * e)) / W;
        const double yc = y[at(i)] - ybar;
        a.w[off + i] = w;
        a.wy[off + i] = w * yc;
        yy += w * yc * yc;
    }
    yy = block_sum<1024>(yy, red);
    if (tid == 0) {
        double *s = a.scal + (int64_t)blockIdx.x * 8;
        s[0] = yy;
        s[1] = W;
        s[2] = tmin;
        s[3] = -ntmax;
        s[4] = disorder == 0.0 ? 1.0 : 0.0;
    }
}

__global__ __launch_bounds__(kBlock) void glsfft_spread_batch_kernel(FftBatchArgs a) {
    const int64_t b = blockIdx.y;
    const int64_t off = a.offsets[b], n = a.offsets[b + 1] - off;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double *t = a.shared_t ? a.t : a.t + off;
    const double *ord = a.scal + b * 8 + 2;
    const double dt = t[i] - ord[0];
    double *g0 = reinterpret_cast<double *>(a.grids + (b * a.ngrid) * a.nfft);
    spread_one(g0, a.nfft, dt, a.wy[off + i], a.df, a.fmin, ord);
    spread_one(g0 + 2 * a.nfft, a.nfft, dt, a.w[off + i], 2.0 * a.df, 2.0 * a.fmin, ord);
    if (a.fit_mean) spread_one(g0 + 4 * a.nfft, a.nfft, dt, a.w[off + i], a.df, a.fmin, ord);
}

__global__ __launch_bounds__(kBlock) void glsfft_epilogue_batch_kernel(FftBatchArgs a) {
    const int64_t b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= a.nf) return;
    const double nfftd = (double)a.nfft, scale = 1.0 / nfftd;
    const double *sc = a.scal + b * 8;
    const double tmin = sc[2];
    const cplx *g = a.result + (b * a.ngrid) * a.nfft;
    const double f = a.fmin + a.df * (double)j;
    double Sh, Ch, S2, C2, S = 0.0, C = 0.0;
    cplx z = g[j];
    z.re *= scale;
    z.im *= scale;
    finish_sum(z, tmin, f, nfftd, Sh, Ch);
    z = g[a.nfft + j];
    z.re *= scale;
    z.im *= scale;
    finish_sum(z, tmin, 2.0 * a.fmin + (2.0 * a.df) * (double)j, nfftd, S2, C2);
    double p;
    if (a.fit_mean) {
        z = g[2 * a.nfft + j];
        z.re *= scale;
        z.im *= scale;
        finish_sum(z, tmin, f, nfftd, S, C);
        p = gls_power_from_sums<true>(Sh, Ch, S, C, S2, C2, sc[0], sc[1], a.psd);
    } else {
        p = gls_power_from_sums<false>(Sh, Ch, S, C, S2, C2, sc[0], sc[1], a.psd);
    }
    a.power[b * a.nf + j] = p;
}

// NaN-aware maximum and its first index per row (Signal.amax / argmax, core.py:202-215)
__global__ __launch_bounds__(kBlock) void row_nanmax_kernel(const double *power, int64_t nf,
                                                            double *amax, int64_t *argmax) {
    __shared__ double rv[kBlock / 64];
    __shared__ long long ri[kBlock / 64];
    const double *x = power + (int64_t)blockIdx.x * nf;
    double best = 0.0;
    long long bi = -1;
    for (int64_t i = threadIdx.x; i < nf; i += kBlock) {
        const double v = x[i];
        if (v == v && (bi < 0 || v > best)) {
            best = v;
            bi = i;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(best, o, 64);
        const long long oi = __shfl_down(bi, o, 64);
        if (oi >= 0 && (bi < 0 || ov > best || (ov == best && oi < bi))) {
            best = ov;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        rv[threadIdx.x >> 6] = best;
        ri[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w)
            if (ri[w] >= 0 && (bi < 0 || rv[w] > best || (rv[w] == best && ri[w] < bi))) {
                best = rv[w];
                bi = ri[w];
            }
        if (amax) amax[blockIdx.x] = bi >= 0 ? best : __builtin_nan("");
        if (argmax) argmax[blockIdx.x] = bi;
    }
}

// ---- host side -----------------------------------------------------------------------------------------
int64_t fft_length(int64_t nf) {
    // 1 << int(nf * 5 - 1).bit_length()                                         (spectral.py:18)
    int64_t v = nf * 5 - 1, bits = 0;
    while (v > 0) {
        ++bits;
        v >>= 1;
    }
    return (int64_t)1 << bits;
}

template <int R>
void launch_pass(hipStream_t st, const cplx *in, cplx *out, int64_t N, int64_t Ns, int batch,
                 int64_t keep, const LiveArgs &lv) {
    const int64_t T = N / R;
    hipLaunchKernelGGL(fft_pass_kernel<R>, dim3((unsigned)((T + kBlock - 1) / kBlock), (unsigned)batch),
                       dim3(kBlock), 0, st, in, out, N, Ns, keep, lv);
}

// Unnormalised inverse FFT of `batch` contiguous arrays of N = 2^bits points in `a`, using `b` (same
// size) as the other half of the ping-pong; returns the buffer that holds the results.  Only the first
// `keep_out` outputs of every transform are needed by the caller (spectral.py:34 keeps [:nf]): the
// final pass skips the stores beyond them.
cplx *inverse_fft(hipStream_t st, cplx *a, cplx *b, int64_t N, int batch = 1, int64_t keep_out = -1,
                  LiveArgs input_live = LiveArgs{nullptr, 0, 1, 0, 0.0}) {
    int bits = 0;
    while (((int64_t)1 << bits) < N) ++bits;
    int64_t Ns = 1;
    cplx *src = a, *dst = b;
    static const bool no_lds = [] { const char *e = getenv("PDC_FFT_NO256"); return e && e[0] == '1'; }();
    const LiveArgs all_live{nullptr, 0, 1, 0, 0.0};
    while (bits > 0) {
        const LiveArgs &lv = Ns == 1 ? input_live : all_live;   // only the first pass reads the deposits
        int r = bits >= 4 ? 4 : bits;
        if (bits == 5) r = 3;  // 5 = 3 + 2 rather than 4 + 1
        // LDS-blocked passes (radix 16 x RB in one trip through HBM): 8 bits at a time, then the
        // remainder in one more LDS pass if it is 5..7 bits, else in a plain radix-2/4/8/16 pass
        int lds_bits = 0;
        if (!no_lds && bits >= 5 && N >= 4096)  // a workgroup covers R * JT = 4096 points
            lds_bits = bits >= 8 ? 8 : bits;
        if (lds_bits) {
            const int64_t keep = (bits == lds_bits && keep_out >= 0) ? keep_out : N;  // final pass only
            const int64_t R = (int64_t)1 << lds_bits;
            const int jt = 256 / (int)(R / 16);
            const dim3 grid((unsigned)((N / R) / jt), (unsigned)batch);
            switch (lds_bits) {
                case 8: hipLaunchKernelGGL(fft_pass_lds_kernel<16>, grid, dim3(256), 0, st, src, dst, N, Ns, keep, lv); break;
                case 7: hipLaunchKernelGGL(fft_pass_lds_kernel<8>, grid, dim3(256), 0, st, src, dst, N, Ns, keep, lv); break;
                case 6: hipLaunchKernelGGL(fft_pass_lds_kernel<4>, grid, dim3(256), 0, st, src, dst, N, Ns, keep, lv); break;
                default: hipLaunchKernelGGL(fft_pass_lds_kernel<2>, grid, dim3(256), 0, st, src, dst, N, Ns, keep, lv); break;
            }
            Ns <<= lds_bits;
            bits -= lds_bits;
            cplx *tmp = src;
            src = dst;
            dst = tmp;
            continue;
        }
        const int64_t keep = (bits == r && keep_out >= 0) ? keep_out : N;  // final pass only
        switch (r) {
            case 4: launch_pass<16>(st, src, dst, N, Ns, batch, keep, lv); break;
            case 3: launch_pass<8>(st, src, dst, N, Ns, batch, keep, lv); break;
            case 2: launch_pass<4>(st, src, dst, N, Ns, batch, keep, lv); break;
            default: launch_pass<2>(st, src, dst, N, Ns, batch, keep, lv); break;
        }
        Ns <<= r;
        bits -= r;
        cplx *tmp = src;
        src = dst;
        dst = tmp;
    }
    return src;
}

void launch_deposit(hipStream_t st, const DepositArgs &d, int64_t n_curves) {
    hipLaunchKernelGGL(glsfft_deposit_kernel, dim3((unsigned)((d.nfft + kDepSpan - 1) / kDepSpan), (unsigned)(n_curves * d.nslots)),
                       dim3(kBlock), 0, st, d);
}

struct FftLayout {
    int64_t wy, w, scal, grid[3], scratch, total;
};

FftLayout fft_layout(int64_t n, int64_t nfft) {
    auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    FftLayout L;
    L.wy = 0;
    L.w = up(n * 8);
    L.scal = L.w + up(n * 8);
    int64_t off = L.scal + up((6 * kMaxPart + 8) * 8);  // scal[8] | part[5][kMaxPart] | ypart[kMaxPart]
    for (int g = 0; g < 3; ++g) L.grid[g] = off + g * nfft * 16;  // contiguous: batched FFT stride = nfft
    off += up(3 * nfft * 16);
    L.scratch = off;
    L.total = off + up(3 * nfft * 16);  // ping-pong partner of all three grids
    return L;
}

}  // namespace

extern "C" {

int64_t pdc_gls_fft_work_bytes(int64_t n, int64_t nf) {
    if (n < 0 || nf < 0) return -1;
    return fft_layout(n, fft_length(nf > 0 ? nf : 1)).total;
}

int pdc_gls_scan_fft_dev(int device, void *stream, const double *d_t, const double *d_y,
                         const double *d_dy, int64_t n, double fmin, double df, int64_t nf,
                         int fit_mean, int psd, double *d_power, void *work, int64_t work_bytes) {
    PDC_REQUIRE(d_t && d_y && (d_power || nf == 0), "gls_fft: NULL argument");
    PDC_REQUIRE(n >= 0 && nf >= 0 && nf < ((int64_t)1 << 40), "gls_fft: bad size");
    if (nf == 0) return PDC_OK;
    const int64_t nfft = fft_length(nf);
    const FftLayout L = fft_layout(n, nfft);
    PDC_REQUIRE(work && work_bytes >= L.total, "gls_fft: workspace too small (%lld < %lld bytes)",
                (long long)work_bytes, (long long)L.total);
    PDC_TRY(use_device(device));
    hipStream_t st = (hipStream_t)stream;
    char *base = static_cast<char *>(work);
    double *wy = reinterpret_cast<double *>(base + L.wy);
    double *w = reinterpret_cast<double *>(base + L.w);
    double *scal = reinterpret_cast<double *>(base + L.scal);
    cplx *grid[3], *scratch = reinterpret_cast<cplx *>(base + L.scratch);
    for (int g = 0; g < 3; ++g) grid[g] = reinterpret_cast<cplx *>(base + L.grid[g]);

    int nparts = (int)((n + 4 * kBlock - 1) / (4 * kBlock));
    nparts = nparts < 1 ? 1 : (nparts > kMaxPart ? kMaxPart : nparts);
    FftPrepArgs p{d_t, d_y, d_dy, n, fit_mean, nparts, wy, w, scal + 8, scal + 8 + 5 * kMaxPart, scal};
    hipLaunchKernelGGL(glsfft_prep_a_kernel, dim3(nparts), dim3(kBlock), 0, st, p);
    hipLaunchKernelGGL(glsfft_prep_c_kernel, dim3(nparts), dim3(kBlock), 0, st, p);
    PDC_HIP(hipGetLastError());

    const int ngrid = fit_mean ? 3 : 2;
    // One grid's ping-pong (32 B per point) may fit in the 256 MiB Infinity Cache while all of them do
    // not: then zero, fill and transform the grids one after the other so that each stays cache-warm
    // (measured at nfft = 2^23: 0.66 vs 0.69 ms); otherwise one memset, one fused spread launch and
    // batched FFT passes (gridDim.y = ngrid; 6.4 vs 6.7 ms at nfft = 2^26).
    const int64_t mall = (int64_t)256 << 20;
    const cplx *result[3] = {nullptr, nullptr, nullptr};
    // (round 6: ALL grids' deposits in one launch up front in both branches - a deposit launch is a chain of dependent
    // searches, ~50 us whatever it writes, and only the live fifth of a grid is written: 27 MB that the other grids'
    // passes may push out of the Infinity Cache cost ~5 us to read back, a second launch cost 50.)
    launch_deposit(st, DepositArgs{d_t, wy, w, nullptr, n, 0, ngrid, 0, 2, scal + 2, 0, nfft, df, fmin,
                                   reinterpret_cast<double *>(grid[0])}, 1);
    if (n > 0) {
        Spread3Args s3{d_t, wy, w, scal, n, nfft, df, fmin, reinterpret_cast<double *>(grid[0]), fit_mean};
        hipLaunchKernelGGL(glsfft_spread3_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, st, s3);
    }
    if (nfft * 32 <= mall && (int64_t)ngrid * nfft * 32 > mall) {
        // the PASSES grid by grid: a grid's 256 MiB ping-pong is what has to stay cache-warm
        for (int g = 0; g < ngrid; ++g)
            result[g] = inverse_fft(st, grid[g], scratch + g * nfft, nfft, 1, nf, LiveArgs{scal + 2, 0, 1, g, df});
    } else {
        const cplx *res = inverse_fft(st, grid[0], scratch, nfft, ngrid, nf, LiveArgs{scal + 2, 0, ngrid, 0, df});
        for (int g = 0; g < ngrid; ++g) result[g] = res + g * nfft;
    }
    PDC_HIP(hipGetLastError());
    FftEpiArgs e;
    e.gh = result[0];
    e.g2 = result[1];
    e.g1 = result[2];
    e.scal = scal;
    e.ypart = p.ypart;
    e.nparts = nparts;
    e.nfft = nfft;
    e.nf = nf;
    e.df = df;
    e.fmin = fmin;
    e.fit_mean = fit_mean;
    e.psd = psd;
    e.power = d_power;
    e.raw_s = e.raw_c = nullptr;
    e.tmin_value = 0.0;
    e.raw = 0;
    hipLaunchKernelGGL(glsfft_epilogue_kernel, dim3((unsigned)((nf + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, st, e);
    PDC_HIP(hipGetLastError());
    return PDC_OK;
}

int pdc_gls_scan_fft(const double *t, const double *y, const double *dy, int64_t n, double fmin,
                     double df, int64_t nf, int fit_mean, int psd, double *power_out, int device) {
    PDC_REQUIRE(t && y && (power_out || nf == 0), "gls_fft: NULL argument");
    PDC_REQUIRE(n >= 0 && nf >= 0, "gls_fft: negative size");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    const int64_t wb = pdc_gls_fft_work_bytes(n, nf);
    void *d_t, *d_y, *d_dy = nullptr, *d_pow, *d_work;
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_y));
    if (dy) PDC_TRY(cached(device, SLOT_IN2, n * 8, &d_dy));
    PDC_TRY(cached(device, SLOT_OUT0, nf * 8, &d_pow));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_work));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_y, y, n * 8, hipMemcpyHostToDevice, st));
    if (dy) PDC_HIP(hipMemcpyAsync(d_dy, dy, n * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(pdc_gls_scan_fft_dev(device, st, (double *)d_t, (double *)d_y, (double *)d_dy, n, fmin, df,
                                 nf, fit_mean, psd, (double *)d_pow, d_work, wb));
    PDC_HIP(hipMemcpyAsync(power_out, d_pow, nf * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}


}  // extern "C"

namespace {
// `picks` != NULL: the bootstrap form - y and dy hold ONE curve of offsets[1] samples, replicate b reads
// sample picks[b n + i]; only the indices cross PCIe
int fft_batch_host(const double *t, const double *y, const double *dy, const int64_t *offsets,
                   int64_t n_curves, int shared_t, double fmin, double df, int64_t nf,
                   int fit_mean, int psd, double *power_out, double *amax_out,
                   int64_t *argmax_out, int device, const int32_t *picks) {
    PDC_REQUIRE(t && y && offsets, "gls_fft_batch: t, y and offsets must not be NULL");
    PDC_REQUIRE(n_curves >= 1 && nf >= 0, "gls_fft_batch: bad size");
    PDC_REQUIRE(power_out || amax_out || argmax_out, "gls_fft_batch: no output requested");
    PDC_REQUIRE(offsets[0] == 0, "gls_fft_batch: offsets[0] must be 0");
    int64_t n_max = 0;
    for (int64_t b = 0; b < n_curves; ++b) {
        const int64_t nb = offsets[b + 1] - offsets[b];
        PDC_REQUIRE(nb >= 0, "gls_fft_batch: offsets must be non-decreasing");
        PDC_REQUIRE(!shared_t || nb == offsets[1] - offsets[0],
                    "gls_fft_batch: with a shared time axis every curve must have the same length");
        n_max = nb > n_max ? nb : n_max;
    }
    if (nf == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    const int64_t n_total = offsets[n_curves];
    const int64_t n_t = shared_t ? offsets[1] : n_total;
    const int64_t nfft = fft_length(nf);
    const int ngrid = fit_mean ? 3 : 2;
    // curves per pass: grids + ping-pong scratch + power within ~8 GiB, and gridDim.y <= 65535
    const int64_t per_curve = 2 * ngrid * nfft * 16 + nf * 8;
    int64_t chunk = ((int64_t)8 << 30) / per_curve;
    chunk = chunk < 1 ? 1 : chunk;
    chunk = chunk > n_curves ? n_curves : chunk;
    chunk = chunk > 65535 / ngrid ? 65535 / ngrid : chunk;
    auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    const int64_t o_w = 0, o_wy = up(n_total * 8), o_scal = o_wy + up(n_total * 8);
    const int64_t o_grid = o_scal + up(n_curves * 64);
    const int64_t o_scratch = o_grid + up(chunk * ngrid * nfft * 16);
    const int64_t o_pow = o_scratch + up(chunk * ngrid * nfft * 16);
    const int64_t wb = o_pow + up(chunk * nf * 8);
    void *d_t, *d_y, *d_dy = nullptr, *d_off, *d_amax = nullptr, *d_arg = nullptr, *d_work;
    const int64_t n_y = picks ? n_t : n_total;   // values / errors on the device: one curve, or all of them
    void *d_picks = nullptr;
    PDC_TRY(cached(device, SLOT_IN0, n_t * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n_y * 8, &d_y));
    if (dy) PDC_TRY(cached(device, SLOT_IN2, n_y * 8, &d_dy));
    PDC_TRY(cached(device, SLOT_IN3, (n_curves + 1) * 8, &d_off));
    if (picks) PDC_TRY(cached(device, SLOT_OUT0, n_total * 4, &d_picks));
    if (amax_out) PDC_TRY(cached(device, SLOT_OUT1, n_curves * 8, &d_amax));
    if (argmax_out) PDC_TRY(cached(device, SLOT_OUT2, n_curves * 8, &d_arg));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_work));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_t, t, n_t * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_y, y, n_y * 8, hipMemcpyHostToDevice, st));
    if (dy) PDC_HIP(hipMemcpyAsync(d_dy, dy, n_y * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_off, offsets, (n_curves + 1) * 8, hipMemcpyHostToDevice, st));
    if (picks) PDC_HIP(hipMemcpyAsync(d_picks, picks, n_total * 4, hipMemcpyHostToDevice, st));
    char *base = static_cast<char *>(d_work);
    FftBatchArgs a;
    a.picks = static_cast<const int32_t *>(d_picks);
    a.t = (double *)d_t;
    a.y = (double *)d_y;
    a.dy = (double *)d_dy;
    a.offsets = (int64_t *)d_off;
    a.shared_t = shared_t;
    a.fit_mean = fit_mean;
    a.psd = psd;
    a.ngrid = ngrid;
    a.nfft = nfft;
    a.nf = nf;
    a.df = df;
    a.fmin = fmin;
    a.w = reinterpret_cast<double *>(base + o_w);
    a.wy = reinterpret_cast<double *>(base + o_wy);
    a.scal = reinterpret_cast<double *>(base + o_scal);
    a.grids = reinterpret_cast<cplx *>(base + o_grid);
    a.result = nullptr;
    a.power = reinterpret_cast<double *>(base + o_pow);
    cplx *scratch = reinterpret_cast<cplx *>(base + o_scratch);
    hipLaunchKernelGGL(glsfft_prep_batch_kernel, dim3((unsigned)n_curves), dim3(1024), 0, st, a);
    PDC_HIP(hipGetLastError());
    const double *scal_all = a.scal;
    for (int64_t c0 = 0; c0 < n_curves; c0 += chunk) {
        const int64_t bc = n_curves - c0 < chunk ? n_curves - c0 : chunk;
        FftBatchArgs c = a;
        c.offsets = a.offsets + c0;
        c.scal = const_cast<double *>(scal_all) + c0 * 8;
        launch_deposit(st, DepositArgs{a.t, a.wy, a.w, c.offsets, 0, shared_t, ngrid, 0, 2, c.scal + 2, 8, nfft, df, fmin,
                                       reinterpret_cast<double *>(c.grids)}, bc);
        if (n_max > 0) {
            hipLaunchKernelGGL(glsfft_spread_batch_kernel,
                               dim3((unsigned)((n_max + kBlock - 1) / kBlock), (unsigned)bc), dim3(kBlock),
                               0, st, c);
            PDC_HIP(hipGetLastError());
        }
        c.result = inverse_fft(st, c.grids, scratch, nfft, (int)(bc * ngrid), nf, LiveArgs{c.scal + 2, 8, ngrid, 0, df});
        PDC_HIP(hipGetLastError());
        hipLaunchKernelGGL(glsfft_epilogue_batch_kernel,
                           dim3((unsigned)((nf + kBlock - 1) / kBlock), (unsigned)bc), dim3(kBlock), 0, st, c);
        PDC_HIP(hipGetLastError());
        if (amax_out || argmax_out) {
            hipLaunchKernelGGL(row_nanmax_kernel, dim3((unsigned)bc), dim3(kBlock), 0, st, c.power, nf,
                               d_amax ? (double *)d_amax + c0 : nullptr,
                               d_arg ? (int64_t *)d_arg + c0 : nullptr);
            PDC_HIP(hipGetLastError());
        }
        if (power_out)
            PDC_HIP(hipMemcpyAsync(power_out + c0 * nf, c.power, (size_t)(bc * nf * 8),
                                   hipMemcpyDeviceToHost, st));
    }
    if (amax_out) PDC_HIP(hipMemcpyAsync(amax_out, d_amax, n_curves * 8, hipMemcpyDeviceToHost, st));
    if (argmax_out) PDC_HIP(hipMemcpyAsync(argmax_out, d_arg, n_curves * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}
}  // namespace

int pdc::gls_bootstrap_fft(const double *t, const double *y, const double *dy, int64_t n, const int32_t *picks,
                           int64_t n_boot, double fmin, double df, int64_t nf, int fit_mean, int psd,
                           double *amax_out, int64_t *argmax_out, int device) {
    std::vector<int64_t> offsets((size_t)n_boot + 1);
    for (int64_t b = 0; b <= n_boot; ++b) offsets[(size_t)b] = b * n;
    return fft_batch_host(t, y, dy, offsets.data(), n_boot, 1, fmin, df, nf, fit_mean, psd, nullptr, amax_out,
                          argmax_out, device, picks);
}

extern "C" {

int pdc_gls_scan_fft_batch(const double *t, const double *y, const double *dy, const int64_t *offsets,
                           int64_t n_curves, int shared_t, double fmin, double df, int64_t nf,
                           int fit_mean, int psd, double *power_out, double *amax_out,
                           int64_t *argmax_out, int device) {
    return fft_batch_host(t, y, dy, offsets, n_curves, shared_t, fmin, df, nf, fit_mean, psd, power_out, amax_out,
                          argmax_out, device, nullptr);
}

int pdc_trig_sums_fft(const double *t, const double *h, int64_t n, double df, int64_t nf,
                      double fmin, double *S_out, double *C_out, int device) {
    PDC_REQUIRE(t && h && S_out && C_out, "trig_sums_fft: NULL argument");
    PDC_REQUIRE(n >= 1 && nf >= 1, "trig_sums_fft: need at least one sample and one frequency");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    const int64_t nfft = fft_length(nf);
    double tmin = t[0], tmax = t[0];
    bool sorted = t[0] == t[0];
    for (int64_t i = 1; i < n; ++i) {
        tmin = t[i] < tmin ? t[i] : tmin;
        tmax = t[i] > tmax ? t[i] : tmax;
        sorted = sorted && t[i] >= t[i - 1];
    }
    const double scal_host[8] = {0.0, 0.0, tmin, tmax, sorted ? 1.0 : 0.0, 0.0, 0.0, 0.0};
    void *d_t, *d_h, *d_s, *d_c, *d_work;
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_h));
    PDC_TRY(cached(device, SLOT_OUT0, nf * 8, &d_s));
    PDC_TRY(cached(device, SLOT_OUT1, nf * 8, &d_c));
    PDC_TRY(cached(device, SLOT_WORK, nfft * 32 + 512, &d_work));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    cplx *grid = reinterpret_cast<cplx *>(d_work), *scratch = grid + nfft;
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_h, h, n * 8, hipMemcpyHostToDevice, st));
    double *scal = reinterpret_cast<double *>(static_cast<char *>(d_work) + nfft * 32);
    PDC_HIP(hipMemcpyAsync(scal, scal_host, sizeof(scal_host), hipMemcpyHostToDevice, st));
    launch_deposit(st, DepositArgs{(double *)d_t, (double *)d_h, (double *)d_h, nullptr, n, 0, 1, 0, 1, scal + 2, 0, nfft, df,
                                   fmin, reinterpret_cast<double *>(grid)}, 1);
    SpreadArgs s{(double *)d_t, (double *)d_h, scal, tmin, n, nfft, df, fmin,
                 reinterpret_cast<double *>(grid)};
    hipLaunchKernelGGL(glsfft_spread_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock),
                       0, st, s);
    PDC_HIP(hipGetLastError());
    cplx *res = inverse_fft(st, grid, scratch, nfft, 1, nf, LiveArgs{scal + 2, 0, 1, 0, df});
    PDC_HIP(hipGetLastError());
    FftEpiArgs e{};
    e.gh = res;
    e.nfft = nfft;
    e.nf = nf;
    e.df = df;
    e.fmin = fmin;
    e.raw_s = (double *)d_s;
    e.raw_c = (double *)d_c;
    e.tmin_value = tmin;
    e.raw = 1;
    hipLaunchKernelGGL(glsfft_epilogue_kernel, dim3((unsigned)((nf + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, st, e);
    PDC_HIP(hipGetLastError());
    PDC_HIP(hipMemcpyAsync(S_out, d_s, nf * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipMemcpyAsync(C_out, d_c, nf * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

}  // extern "C"
