// Generalized Lomb-Scargle as an exact direct summation on gfx950 (fp64 VALU bound).
//
// Replaces GLS.__call__ from the weights onward (/root/reference/src/periodicity/
// spectral.py:99-132) and the _trig_sum seam (spectral.py:11-40, by the sums its docstring
// defines at :13-15).
//
// Decomposition
//   gls_prep_kernel   one workgroup per light curve: weights (spectral.py:99-108), YY (:120) and
//                     one 48-byte record per sample {sqrt(w)*y, sqrt(w), cos(2 pi delta t'),
//                     sin(2 pi delta t'), 2 cos(2 pi delta t'), t' = t - t0} ({w, w, ..} for the raw
//                     trig sums, whose weights the caller supplies).
//   gls_scan_kernel   each thread owns K consecutive trial frequencies.  A wave reads the curve's
//                     records through the scalar cache (they are wave-uniform: s_load, SGPR
//                     operands of the fmas).  Thread L of a tile starts at phase
//                     theta_tile + L Theta, Theta = 2 pi K delta t': per chunk the workgroup builds,
//                     with three software sincos per SAMPLE (phase carried in cycles with an exact
//                     fma product) and short rotation chains, the LDS tables {sin, cos}(theta_tile
//                     + 8 q Theta) and {sin, cos}(b Theta), b < 8, so that the seed of a (sample,
//                     thread) is ONE plane rotation of two table entries.  Then one rotation by
//                     the per-sample angle 2 pi delta t' and K-2 steps of the three-term
//                     recurrence x[k+1] = 2cos(theta) x[k] - x[k-1] walk the uniform grid; 6
//                     running sums per frequency (Sh, Ch, S, C, sum w s^2, sum w s c — the
//                     2-omega sums follow from the double-angle identities), each one fma because
//                     sin/cos are carried pre-multiplied by sqrt(w).  The epilogue
//                     (spectral.py:113-132) is fused, so only power[nf] is written.
//   gls_peak_kernel   NaN-aware max / argmax per curve from per-workgroup partials.
//
// No MFMA: with one weight vector per curve the accumulation is a matrix-vector product and the
// cost is the transcendental/rotation work on the vector ALU.
#include "pdc_internal.h"
#include "gls_epilogue.h"

#include <cstdlib>

using namespace pdc;

namespace {

constexpr int kBlock = 256;
constexpr int kPrepBlock = 1024;

// MODE_TREND (round 6, BGLST): the six sums of MODE_FIT_MEAN on the UNcentred values plus sum w t' sin, sum w t' cos,
// and the marginal log-likelihood of the harmonic + linear-trend model as epilogue (bglst_loglik below).
enum Mode { MODE_FIT_MEAN = 0, MODE_NO_MEAN = 1, MODE_RAW = 2, MODE_TREND = 3 };
constexpr bool mode_fits_mean(int m) { return m == MODE_FIT_MEAN; }
constexpr bool mode_sums_w(int m) { return m == MODE_FIT_MEAN || m == MODE_TREND; }   // sum w sin, sum w cos are wanted

// Frequency-independent inputs of the BGLST epilogue, computed by the host in fp64 (O(N)); sums are over the
// normalised weights w_i = sigma_i^-2 / W, tau = (t - t_ref) / span.
struct TrendScalars {
    double W;                 // sum sigma_i^-2
    double yy, y1, ty, tt, t1; // sum w y^2, sum w y, sum w tau y, sum w tau^2, sum w tau
    double shift;             // (t0 - t_ref) / span: the kernel's sums are over t' = t - t0, tau = t' / span + shift
    double inv_span;          // 1 / span
    double prec_a, prec_alpha, prec_beta;   // prior precisions 1 / sigma^2 (alpha in units of 1 / span)
    double log_const;         // sum log(2 pi sigma_i^2) + log(sigma_A^4 sigma_alpha^2 sigma_beta^2)
};

struct GlsArgs {
    const double *rec;       // [n_total][6]
    const int64_t *offsets;  // [n_curves + 1] or nullptr (single curve)
    const double *scal;      // [n_curves][4] = {YY, sum w, sum err^-2, t0}
    int64_t n_total, n_curves, tiles;
    double f0, delta;
    int64_t j_begin, nf;
    int psd;
    double *power;     // [n_curves][nf] or nullptr
    double *raw_s;     // MODE_RAW outputs
    double *raw_c;
    double *blk_max;   // [n_curves * tiles] or nullptr
    int64_t *blk_arg;
    // few frequencies x many samples (single curve): the samples are cut into gridDim.y parts of z_len
    // (a multiple of the chunk size); every part leaves its six sums per frequency in `partial`
    // ([part][6][nf]) and gls_finish_kernel adds the parts in order and applies the epilogue
    int64_t z_len = 0;
    int parts = 1, parts_by_xcd = 0;   // parts_by_xcd: part z runs on XCD z % 8 (1-D grid), see the kernel
    double *partial = nullptr;
    // balanced pieces (gls_scan_kernel<..., BAL = true>, single curve): the (tile, chunk) space - bal_units =
    // tiles x bal_chunks chunk-units, tile-major - is cut into bal_slots equal runs, one per workgroup, so that
    // every resident workgroup slot gets the same work whatever tiles / slots is; a run covers pieces of at most
    // two tiles, a tile falls into at most three runs (`partial` = [3][6][nf]), gls_finish_kernel adds them
    int64_t bal_slots = 0, bal_units = 0, bal_chunks = 0, bal_tile_freqs = 0;
    TrendScalars trend = {};   // MODE_TREND only
};

// balanced pieces: run s covers the units [s U / W, (s + 1) U / W); the run that holds unit x
__device__ __forceinline__ int64_t bal_slot_of(const GlsArgs &a, int64_t x) {
    int64_t s = x * a.bal_slots / a.bal_units;
    while ((s + 1) * a.bal_units / a.bal_slots <= x) ++s;
    while (s * a.bal_units / a.bal_slots > x) --s;
    return s;
}

struct PrepArgs {
    const double *t, *y, *dy;
    const int64_t *offsets;
    int64_t n_total;
    int shared_t, fit_mean, raw;
    double delta;
    double *rec, *scal;
    // shared-time-axis batches (gls_shared_kernel): weights transposed to [sample][curve]
    double *rw = nullptr;   // {w (y - ybar), w} pairs, or w (y - ybar) alone when all weights are equal
    double *tp = nullptr;   // [n] t - t0
    int64_t bpad = 0;
    // bootstrap replicates by INDEX (spectral.py:146-148): y and dy are ONE curve of n samples, replicate b's
    // sample i is (y[picks[b n + i]], dy[picks[b n + i]]) - the resampled arrays never exist
    const int32_t *picks = nullptr;
};

// ---- prologue: spectral.py:99-108, 120 ------------------------------------------------------------
__global__ __launch_bounds__(kPrepBlock) void gls_prep_kernel(PrepArgs a) {
    __shared__ double red[kPrepBlock / 64];
    const int tid = threadIdx.x;
    const int64_t off = a.offsets ? a.offsets[blockIdx.x] : 0;
    const int64_t n = a.offsets ? a.offsets[blockIdx.x + 1] - off : a.n_total;
    const double *t = a.shared_t ? a.t : a.t + off;
    const int32_t *pk = a.picks ? a.picks + off : nullptr;
    const double *y = pk ? a.y : a.y + off;
    const double *dy = a.dy ? (pk ? a.dy : a.dy + off) : nullptr;
    auto at = [pk](int64_t i) -> int64_t { return pk ? (int64_t)pk[i] : i; };
    double *rec = a.rec + off * 6;
    const double t0 = n > 0 ? t[0] : 0.0;

    double W = 1.0, ybar = 0.0;
    if (!a.raw) {
        double acc = 0.0;  // w = err**-2 ; w.sum()
        for (int64_t i = tid; i < n; i += kPrepBlock) {
            const double e = dy ? dy[at(i)] : 1.0;
            acc += 1.0 / (e * e);
        }
        W = block_sum<kPrepBlock>(acc, red);
        if (a.fit_mean) {  // np.dot(w / w.sum(), values)
            acc = 0.0;
            for (int64_t i = tid; i < n; i += kPrepBlock) {
                const double e = dy ? dy[at(i)] : 1.0;
                acc += (1.0 / (e * e)) / W * y[at(i)];
            }
            ybar = block_sum<kPrepBlock>(acc, red);
        }
    }
    double yy = 0.0, wsum = 0.0;
    if (a.rw) {
        // shared time axis: no per-curve records, only this curve's column of the weight table
        for (int64_t i = tid; i < n; i += kPrepBlock) {
            const double e = dy ? dy[at(i)] : 1.0;
            const double w = (1.0 / (e * e)) / W;
            const double yc = y[at(i)] - ybar;
            const double wy = w * yc;
            yy += wy * yc;
            wsum += w;
            if (dy) reinterpret_cast<double2 *>(a.rw)[i * a.bpad + blockIdx.x] = make_double2(wy, w);
            else a.rw[i * a.bpad + blockIdx.x] = wy;
            if (blockIdx.x == 0) a.tp[i] = t[i] - t0;
        }
    } else
    for (int64_t i = tid; i < n; i += kPrepBlock) {
        const double tp = t[i] - t0;
        double w, wy;
        if (a.raw) {
            w = y[at(i)];  // the caller's weights, used as given (spectral.py:13-15)
            wy = w;
        } else {
            const double e = dy ? dy[at(i)] : 1.0;
            w = (1.0 / (e * e)) / W;
            const double yc = y[at(i)] - ybar;
            wy = w * yc;
            yy += wy * yc;
            wsum += w;
        }
        double sd, cd;
        sincos_cycles(frac_product(a.delta, tp), sd, cd);
        if (!a.raw) {  // the kernel carries sqrt(w) sin / sqrt(w) cos
            w = sqrt(w);
            wy = w * (y[at(i)] - ybar);
        }
        double2 *r = reinterpret_cast<double2 *>(rec + i * 6);
        r[0] = make_double2(wy, w);
        r[1] = make_double2(cd, sd);
        r[2] = make_double2(cd + cd, tp);
    }
    yy = block_sum<kPrepBlock>(yy, red);
    wsum = block_sum<kPrepBlock>(wsum, red);
    if (tid == 0) {
        double *s = a.scal + (int64_t)blockIdx.x * 4;
        s[0] = yy;
        s[1] = wsum;
        s[2] = W;
        s[3] = t0;
    }
}

// ---- the same prologue, grid-wide, for ONE long light curve (three short launches instead of one
// 1024-thread workgroup walking the whole curve): A = partial sums of err^-2 and err^-2*y;
// B = every workgroup re-reduces A's partials in a fixed order (identical W and ybar everywhere),
// writes its records and its partials of YY and sum w; C = one workgroup folds those into scal.
constexpr int kPrepParts = 512;

struct WidePrepArgs {
    PrepArgs p;
    int nparts;
    double *part;  // [4][kPrepParts]: sum err^-2 | sum err^-2 y | YY | sum w
};

__global__ __launch_bounds__(kBlock) void gls_prep_wide_a(WidePrepArgs a) {
    __shared__ double red[kBlock / 64];
    double sw = 0.0, swy = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.p.n_total;
         i += (int64_t)gridDim.x * kBlock) {
        const double e = a.p.dy ? a.p.dy[i] : 1.0;
        const double wr = 1.0 / (e * e);
        sw += wr;
        swy += wr * a.p.y[i];
    }
    sw = block_sum<kBlock>(sw, red);
    swy = block_sum<kBlock>(swy, red);
    if (threadIdx.x == 0) {
        a.part[blockIdx.x] = sw;
        a.part[kPrepParts + blockIdx.x] = swy;
    }
}

__device__ __forceinline__ double fold_partials(const double *p, int count, double *red) {
    double v = 0.0;
    for (int i = threadIdx.x; i < count; i += kBlock) v += p[i];
    return block_sum<kBlock>(v, red);
}

__global__ __launch_bounds__(kBlock) void gls_prep_wide_b(WidePrepArgs a) {
    __shared__ double red[kBlock / 64];
    const double W = fold_partials(a.part, a.nparts, red);
    const double ybar = a.p.fit_mean ? fold_partials(a.part + kPrepParts, a.nparts, red) / W : 0.0;
    const double t0 = a.p.t[0];
    double yy = 0.0, wsum = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.p.n_total;
         i += (int64_t)gridDim.x * kBlock) {
        const double tp = a.p.t[i] - t0;
        const double e = a.p.dy ? a.p.dy[i] : 1.0;
        const double w = (1.0 / (e * e)) / W;
        const double yc = a.p.y[i] - ybar;
        const double wy = w * yc;
        yy += wy * yc;
        wsum += w;
        double sd, cd;
        sincos_cycles(frac_product(a.p.delta, tp), sd, cd);
        const double rw = sqrt(w);  // the scan carries sqrt(w) sin / sqrt(w) cos
        double2 *r = reinterpret_cast<double2 *>(a.p.rec + i * 6);
        r[0] = make_double2(rw * yc, rw);
        r[1] = make_double2(cd, sd);
        r[2] = make_double2(cd + cd, tp);
    }
    yy = block_sum<kBlock>(yy, red);
    wsum = block_sum<kBlock>(wsum, red);
    if (threadIdx.x == 0) {
        a.part[2 * kPrepParts + blockIdx.x] = yy;
        a.part[3 * kPrepParts + blockIdx.x] = wsum;
    }
}

__global__ __launch_bounds__(kBlock) void gls_prep_wide_c(WidePrepArgs a) {
    __shared__ double red[kBlock / 64];
    const double W = fold_partials(a.part, a.nparts, red);
    const double yy = fold_partials(a.part + 2 * kPrepParts, a.nparts, red);
    const double wsum = fold_partials(a.part + 3 * kPrepParts, a.nparts, red);
    if (threadIdx.x == 0) {
        a.p.scal[0] = yy;
        a.p.scal[1] = wsum;
        a.p.scal[2] = W;
        a.p.scal[3] = a.p.t[0];
    }
}

// ---- epilogue: spectral.py:113-132 (gls_epilogue.h); the 2-omega sums come from the double-angle
// identities sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a ------------------------------------------
template <int MODE>
__device__ __forceinline__ double gls_power(double Sh, double Ch, double S, double C, double SS,
                                            double SC, double YY, double Wsum, double Werr,
                                            int psd) {
    const double S2 = 2.0 * SC;            // sum w sin(2 omega t)
    const double C2 = Wsum - 2.0 * SS;     // sum w cos(2 omega t)
    return gls_power_from_sums<mode_fits_mean(MODE)>(Sh, Ch, S, C, S2, C2, YY, Werr, psd);
}

// ---- BGLST epilogue: log marginal likelihood of y = A cos(2 pi f t) + B sin(2 pi f t) + alpha tau + beta + noise ------
// (Olspert, Pelt, Kapyla & Lehtinen 2018, A&A 615, A111: Bayesian generalised Lomb-Scargle with trend; the name the
// reference exports at spectral.py:7,207-208 for an empty class.)  Independent zero-mean Gaussian priors N(0, sigma_A^2)
// on A and B, N(0, sigma_alpha^2), N(0, sigma_beta^2) on the trend, Gaussian noise sigma_i: the model is linear in its
// four parameters, so they integrate out in closed form.  With Phi = [cos, sin, tau, 1], N = diag sigma_i^2,
// M = Phi^T N^-1 Phi + diag(prior precisions), b = Phi^T N^-1 y:
//     log p(y | f) = -1/2 [ y^T N^-1 y - b^T M^-1 b + log |M| + log |Sigma_prior| + sum log(2 pi sigma_i^2) ]
// (Woodbury + the matrix determinant lemma on C = Phi Sigma Phi^T + N).  The paper reaches the same number by
// rotating (cos, sin) so that their cross term vanishes and completing squares one parameter at a time; with
// sigma_A = sigma_B the likelihood does not depend on that rotation - nor on the kernel's own time origin t0.
// M is 4 x 4, symmetric positive definite (the priors make it so): an unrolled Cholesky.
__device__ __forceinline__ double bglst_loglik(double Sh, double Ch, double S, double C, double SS, double SC,
                                               double TS, double TC, double Wsum, const TrendScalars &q) {
    // sums over tau = t' / span + shift from the kernel's sums over t'
    const double tc = TC * q.inv_span + q.shift * C, ts = TS * q.inv_span + q.shift * S;
    const double W = q.W;
    // M, lower triangle, basis order (cos, sin, tau, 1)
    const double m00 = W * (Wsum - SS) + q.prec_a;
    const double m10 = W * SC, m11 = W * SS + q.prec_a;
    const double m20 = W * tc, m21 = W * ts, m22 = W * q.tt + q.prec_alpha;
    const double m30 = W * C, m31 = W * S, m32 = W * q.t1, m33 = W * Wsum + q.prec_beta;
    const double b0 = W * Ch, b1 = W * Sh, b2 = W * q.ty, b3 = W * q.y1;
    // Cholesky M = L L^T
    const double l00 = __builtin_sqrt(m00);
    const double l10 = m10 / l00, l20 = m20 / l00, l30 = m30 / l00;
    const double l11 = __builtin_sqrt(m11 - l10 * l10);
    const double l21 = (m21 - l20 * l10) / l11, l31 = (m31 - l30 * l10) / l11;
    const double l22 = __builtin_sqrt(m22 - l20 * l20 - l21 * l21);
    const double l32 = (m32 - l30 * l20 - l31 * l21) / l22;
    const double l33 = __builtin_sqrt(m33 - l30 * l30 - l31 * l31 - l32 * l32);
    // z = L^-1 b;  b^T M^-1 b = |z|^2
    const double z0 = b0 / l00;
    const double z1 = (b1 - l10 * z0) / l11;
    const double z2 = (b2 - l20 * z0 - l21 * z1) / l22;
    const double z3 = (b3 - l30 * z0 - l31 * z1 - l32 * z2) / l33;
    const double quad = W * q.yy - (z0 * z0 + z1 * z1 + z2 * z2 + z3 * z3);
    const double logdet = 2.0 * (log(l00) + log(l11) + log(l22) + log(l33));
    return -0.5 * (quad + logdet + q.log_const);
}

// ---- the scan ------------------------------------------------------------------------------------------
// K = trial frequencies per thread; SPLIT ("S") = waves of the workgroup that share one 64-lane frequency
// tile and split every staged chunk of samples between them (S = 1, 2 or 4), so that a short grid
// still puts >= 2 waves on every SIMD; their partial sums are combined through LDS in a fixed
// order before the epilogue, so results do not depend on timing.
template <int K, int MODE, int SPLIT, bool BAL = false>
__global__ __launch_bounds__(kBlock, (K >= 16 ? 2 : 1)) void gls_scan_kernel(GlsArgs a) {
    constexpr int FT = kBlock / SPLIT;        // frequency-owning threads per workgroup
    constexpr bool TIGHT = BAL || K >= 16;    // 192 accumulators: nothing loop-invariant may stay in registers across the loops (see the table fill)
    constexpr int COLS = FT / 64;             // 64-lane columns of the tile
    constexpr int kChunk = SPLIT == 1 ? 64 : 128;  // samples per rotation-table chunk
    // per sample: {sin, cos} of theta_tile + 8 q Theta, q < 8 COLS (the seed of lanes 8q .. 8q+7 before
    // their own offset), scaled by sqrt(w) where the sums want it | {sin, cos}(b Theta), b < 8
    __shared__ double2 tab[kChunk + 1][COLS * 8 + 8 + 1];  // + 1: rows start 16 B apart modulo 128 B (bank spread)
    __shared__ double red_v[4];
    __shared__ long long red_i[4];
    __shared__ double2 sc_k[6];   // TIGHT: the sincos coefficients of the table fill (pdc_device.h: sincos_cycles_k)
    const int tid_ = threadIdx.x;
    if (TIGHT && tid_ < 6) sc_k[tid_] = sincos_coefficient(tid_);   // (read after the chunk loop's first barrier)

    // Workgroup p runs on XCD p % 8 (observed dispatch rule, used for speed only): hand each XCD a
    // contiguous run of logical tiles so the tiles of one curve share that XCD's L2.
    const int64_t G = BAL ? a.bal_slots : a.n_curves * a.tiles;
    const int64_t per_xcd = (G + 7) / 8;
    int64_t L = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    int zpart = blockIdx.y;
    if (a.parts_by_xcd) {
        // sample parts (8 or more): XCD e takes the parts e, e + 8, ... with all their tiles, so that an XCD's
        // L2 streams one or two stretches of the records at a time instead of all of them
        const int64_t q = blockIdx.x / 8, zi = q / G;
        L = q - zi * G;
        zpart = (int)(blockIdx.x % 8) + 8 * (int)zi;
        if (zpart >= a.parts) return;
    }
    if (L >= G) return;
    int64_t bal_u = 0, bal_hi = 0;   // BAL: this workgroup's run of chunk-units
    if (BAL) {
        bal_u = L * a.bal_units / a.bal_slots;
        bal_hi = (L + 1) * a.bal_units / a.bal_slots;
        if (bal_u >= bal_hi) return;
    }
    do {   // (one trip; BAL: one trip per piece - the stretch of ONE tile's chunks inside the run)
    int64_t curve = L / a.tiles;
    int64_t tile = L - curve * a.tiles;
    int64_t bal_take = 0;
    int bal_piece = 0;
    int64_t bal_c0 = 0;
    if (BAL) {
        // the LAST piece of the run first: a run is the tail of one tile's chunks + the head [0, c1) of the next
        // tile's, and starting with the head every workgroup sweeps the records from chunk 0 upwards in step
        // with all the others - it merely skips the ~(1 - tiles / slots) of the chunks its neighbour covers - so
        // the 4.8 MB of records stream through each XCD's L2 once, as in the unbalanced launch (pieces in run
        // order put every workgroup at another offset: 1.3 GB of L2 misses per launch instead of 0.07)
        curve = 0;
        // (the divisors through an opaque copy per piece: the reciprocal sequences of these uniform divisions are
        // otherwise computed once on the VALU, kept in VGPRs across the whole piece loop - and spilled, round 5)
        int64_t b_chunks = a.bal_chunks, b_slots = a.bal_slots, b_units = a.bal_units;
        asm volatile("" : "+s"(b_chunks), "+s"(b_slots), "+s"(b_units));
        tile = (bal_hi - 1) / b_chunks;
        const int64_t t0 = tile * b_chunks;
        bal_c0 = (bal_u > t0 ? bal_u : t0) - t0;
        bal_take = bal_hi - t0 - bal_c0;
        int64_t s_of = t0 * b_slots / b_units;              // bal_slot_of(a, t0)
        while ((s_of + 1) * b_units / b_slots <= t0) ++s_of;
        while (s_of * b_units / b_slots > t0) --s_of;
        bal_piece = (int)(L - s_of);   // 0, 1 or 2: which of the tile's pieces
    }

    const int64_t off = a.offsets ? a.offsets[curve] : 0;
    const int64_t n = a.offsets ? a.offsets[curve + 1] - off : a.n_total;
    const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6), part = wave % SPLIT;
    const int col = wave / SPLIT;
    // first local frequency of the tile and of this thread
    const int64_t jt = tile * FT * (int64_t)K;
    const int64_t jl = jt + (col * 64 + (tid_ & 63)) * (int64_t)K;
    // numpy's arange fill rule: start + i*delta, two roundings (no fma)
    const double f_tile = __dadd_rn(a.f0, __dmul_rn((double)(a.j_begin + jt), a.delta));
    const double kdelta = (double)K * a.delta;  // spacing of the threads' first frequencies (exact)

    static_assert(MODE != MODE_TREND || (SPLIT == 1 && !BAL && K <= 8), "BGLST: eight sums per frequency, whole curves per tile");
    double Sh[K], Ch[K], S[K], C[K], SS[K], SC[K];
    double TS[K], TC[K];   // MODE_TREND: sum w t' sin, sum w t' cos (dead code in the other instances)
#pragma unroll
    for (int k = 0; k < K; ++k) Sh[k] = Ch[k] = S[k] = C[k] = SS[k] = SC[k] = TS[k] = TC[k] = 0.0;

    // plane rotation of {sin, cos} pairs: angle(x) + angle(y)
    auto rot = [](const double2 x, const double2 y) {
        return make_double2(__builtin_fma(x.x, y.y, x.y * y.x), __builtin_fma(x.y, y.y, -(x.x * y.x)));
    };
    const int64_t s_begin = BAL ? bal_c0 * kChunk : (a.partial ? (int64_t)zpart * a.z_len : 0);
    const int64_t s_stop = BAL ? s_begin + bal_take * kChunk : s_begin + a.z_len;
    const int64_t s_end = (BAL || a.partial) ? (s_stop < n ? s_stop : n) : n;
    for (int64_t base = s_begin; base < s_end; base += kChunk) {
        __syncthreads();  // everyone is done with the previous chunk's tables
        // (TIGHT: the thread id through an opaque copy once per chunk - the LDS addresses and masks built from it are
        // loop-invariant, and hoisted out of the piece loop they were spilled; rebuilt per chunk they cost a few
        // 32-bit instructions per 128 samples)
        int tid = tid_;
        if (TIGHT) asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int slot_a = col * 8 + (lane >> 3), slot_b = COLS * 8 + (lane & 7);
        // ---- per-sample rotation tables (two threads per sample) ------------------------------------
        // Thread (col, lane) starts at phase theta_tile + (64 col + 8 a + b) Theta with a = lane / 8,
        // b = lane % 8 and Theta = 2 pi K delta t': its seed is tab[8 col + a] rotated by tab[8 COLS + b]
        // instead of a sincos.  The even thread of a sample makes {sin, cos}(b Theta) (one sincos, a
        // chain of 6 rotations) and the tile's base phase (one sincos, scaled by sqrt(w) where the
        // sums want it); the odd thread walks that base in steps of 8 Theta (one sincos, 8 COLS - 1
        // rotations).
        if (tid < 2 * kChunk) {
            const int il = tid >> 1;
            // (rows past the end of the curve are never accumulated; they only need finite input)
            const bool live = base + il < s_end;
            const double tp = live ? a.rec[(off + base + il) * 6 + 5] : 0.0;
            const double sqw = live ? a.rec[(off + base + il) * 6 + 1] : 0.0;
            double2 step1, cur;
            double2 kk[6];
            if (TIGHT) {
#pragma unroll
                for (int q = 0; q < 6; ++q) kk[q] = sc_k[q];
            }
            auto sincos_fill = [&](const double r, double &s_, double &c_) {
                if (TIGHT) sincos_cycles_k(r, kk, s_, c_);
                else sincos_cycles(r, s_, c_);
            };
            if ((tid & 1) == 0) {
                // lane offsets b Theta, b < 8; and the tile's base phase for the neighbour
                sincos_fill(frac_product(kdelta, tp), step1.x, step1.y);
                double one = 1.0, zero = 0.0;
                if (TIGHT) asm volatile("" : "+v"(one), "+v"(zero));   // (made here, not kept in four registers across the loops)
                tab[il][COLS * 8] = make_double2(zero, one);
                tab[il][COLS * 8 + 1] = step1;
                cur = step1;
#pragma unroll
                for (int q = 2; q < 8; ++q) {
                    cur = rot(cur, step1);
                    tab[il][COLS * 8 + q] = cur;
                }
                sincos_fill(frac_product(f_tile, tp), cur.x, cur.y);
                if (MODE == MODE_FIT_MEAN || MODE == MODE_NO_MEAN || MODE == MODE_TREND) {
                    // carry u = sqrt(w) sin, v = sqrt(w) cos: rotations and the recurrence are linear,
                    // and every sum becomes one fma (the record holds sqrt(w) and sqrt(w) y)
                    cur.x *= sqw;
                    cur.y *= sqw;
                }
            } else {
                sincos_fill(frac_product(8.0 * kdelta, tp), step1.x, step1.y);
            }
            // the odd thread walks the base in steps of 8 Theta
            double2 b0;
            b0.x = __shfl_xor(cur.x, 1, 64);
            b0.y = __shfl_xor(cur.y, 1, 64);
            if (tid & 1) {
                tab[il][0] = b0;
#pragma unroll
                for (int q = 1; q < COLS * 8; ++q) {
                    b0 = rot(b0, step1);
                    tab[il][q] = b0;
                }
            }
        }
        __syncthreads();
        const int cnt = (int)((s_end - base) < kChunk ? (s_end - base) : kChunk);
        const int i_end = cnt < (part + 1) * (kChunk / SPLIT) ? cnt : (part + 1) * (kChunk / SPLIT);
        const int i_beg = part * (kChunk / SPLIT);
        // Software pipeline: everything sample i+1 needs is requested while sample i is accumulated;
        // two samples per trip with two register sets (A, B) that swap roles, so the read-ahead costs
        // no register copies.  The record fields are wave-uniform: they come through the scalar
        // cache (s_load, constant address space) and feed the fmas as SGPR operands - no LDS or VGPR
        // traffic for them; the two table entries are LDS reads.  Scalar loads return out of order,
        // so the wait for set A (lgkmcnt(0)) sits right before the request for set B goes out.
        // (The read-ahead touches up to two records past the curve and one padding table row.)
        using d4 = double __attribute__((ext_vector_type(4)));
        using cd4 = __attribute__((address_space(4))) const d4;
        using cdbl = __attribute__((address_space(4))) const double;
        const cd4 *srec = reinterpret_cast<const cd4 *>(reinterpret_cast<uintptr_t>(a.rec + (off + base) * 6));
        struct Ahead {
            d4 r;  // {sqrt(w) y, sqrt(w), cos, sin (2 pi delta t')}
            double cd2;
            double tp;   // (MODE_TREND) t'
            double2 qa, qt;
        };
        auto fetch = [&](const int i) {
            Ahead h;
            h.qa = tab[i][slot_a];
            h.qt = tab[i][slot_b];
            const cd4 *rp = reinterpret_cast<const cd4 *>(reinterpret_cast<const cdbl *>(srec) + i * 6);
            h.r = rp[0];
            h.cd2 = reinterpret_cast<const cdbl *>(rp)[4];
            h.tp = MODE == MODE_TREND ? reinterpret_cast<const cdbl *>(rp)[5] : 0.0;
            return h;
        };
        auto accumulate = [&](const Ahead &h) {
            const double2 seed = rot(h.qa, h.qt);
            const double wy = h.r[0], w = h.r[1], cd = h.r[2], sd = h.r[3], cd2 = h.cd2;
            const double wt = MODE == MODE_TREND ? w * h.tp : 0.0;   // sqrt(w) t': s and c carry the other sqrt(w)
            double s = seed.x, c = seed.y;
            double sp = 0.0, cp = 0.0;  // previous step of the recurrence
#pragma unroll
            for (int k = 0; k < K; ++k) {
                Sh[k] = __builtin_fma(wy, s, Sh[k]);
                Ch[k] = __builtin_fma(wy, c, Ch[k]);
                if (MODE != MODE_RAW) {
                    if (mode_sums_w(MODE)) {
                        S[k] = __builtin_fma(w, s, S[k]);
                        C[k] = __builtin_fma(w, c, C[k]);
                    }
                    if (MODE == MODE_TREND) {
                        TS[k] = __builtin_fma(wt, s, TS[k]);
                        TC[k] = __builtin_fma(wt, c, TC[k]);
                    }
                    SS[k] = __builtin_fma(s, s, SS[k]);
                    SC[k] = __builtin_fma(s, c, SC[k]);
                }
                if (k + 1 < K) {
                    double sn, cn;
                    if (k == 0) {
                        // first grid step: plane rotation by 2 pi delta t'
                        cn = __builtin_fma(c, cd, -(s * sd));
                        sn = __builtin_fma(s, cd, c * sd);
                    } else {
                        // later steps: x[k+1] = 2 cos(theta) x[k] - x[k-1]  (1 fma per component;
                        // rounding grows like K^2 eps, far below the 1e-6 gate for K <= 16)
                        cn = __builtin_fma(cd2, c, -cp);
                        sn = __builtin_fma(cd2, s, -sp);
                    }
                    cp = c;
                    sp = s;
                    c = cn;
                    s = sn;
                }
            }
        };
        Ahead A = fetch(i_beg);
        int i = i_beg;
        for (; i + 1 < i_end; i += 2) {
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): set A has arrived
            Ahead B = fetch(i + 1);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(A);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            A = fetch(i + 2);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(B);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (i < i_end) accumulate(A);
    }

    const int tid = tid_;
    if (SPLIT > 1) {
        // fold the partial sums of parts 1..S-1 into part 0, one frequency at a time, through the
        // rotation-table buffer (6 doubles per contributing thread)
        double *xch = reinterpret_cast<double *>(tab);
        const int slot = ((wave / SPLIT) * (SPLIT - 1) + (part - 1)) * 64 + (tid & 63);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            __syncthreads();
            if (part > 0) {
                double *d = xch + slot * 6;
                d[0] = Sh[k]; d[1] = Ch[k]; d[2] = S[k]; d[3] = C[k]; d[4] = SS[k]; d[5] = SC[k];
            }
            __syncthreads();
            if (part == 0) {
#pragma unroll
                for (int q = 0; q < SPLIT - 1; ++q) {
                    const double *d = xch + (((wave / SPLIT) * (SPLIT - 1) + q) * 64 + (tid & 63)) * 6;
                    Sh[k] += d[0]; Ch[k] += d[1]; S[k] += d[2]; C[k] += d[3]; SS[k] += d[4]; SC[k] += d[5];
                }
            }
        }
    }
    const bool owner = part == 0;  // only part 0 holds complete sums

    if (MODE != MODE_RAW && (BAL || a.partial)) {   // (workgroup-uniform) this sample part's sums; gls_finish_kernel does the rest
        double *out = a.partial + (int64_t)(BAL ? bal_piece : zpart) * 6 * a.nf;
        // array by array: a lane's K frequencies are one 128-byte line per array, and a line written to the end
        // before the next is begun leaves L2 whole (with the six arrays interleaved per frequency 50 MB of
        // half-written lines were in flight chip-wide and went out to HBM 3.3 times)
        auto put = [&](const double (&v)[K], const int q) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (owner && jl + k < a.nf) out[(int64_t)q * a.nf + jl + k] = v[k];
        };
        put(Sh, 0);
        put(Ch, 1);
        put(S, 2);
        put(C, 3);
        put(SS, 4);
        put(SC, 5);
        if (BAL) {
            bal_hi -= bal_take;
            continue;   // the piece before, if any
        }
        return;
    }
    if (BAL) return;   // (not reached: a balanced launch always leaves partial sums)

    const double *sc = a.scal + curve * 4;
    if (MODE == MODE_RAW) {
        // undo the t0 shift: sums were taken over t' = t - t0
        const double t0 = sc[3];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t j = jl + k;
            if (owner && j < a.nf) {
                const double f = __dadd_rn(a.f0, __dmul_rn((double)(a.j_begin + j), a.delta));
                double s0, c0;
                sincos_cycles(frac_product(f, t0), s0, c0);
                a.raw_s[curve * a.nf + j] = Sh[k] * c0 + Ch[k] * s0;
                a.raw_c[curve * a.nf + j] = Ch[k] * c0 - Sh[k] * s0;
            }
        }
        return;
    }

    const double YY = sc[0], Wsum = sc[1], Werr = sc[2];
    double best = 0.0;
    long long best_j = -1;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t j = jl + k;
        if (owner && j < a.nf) {
            const double p = MODE == MODE_TREND ? bglst_loglik(Sh[k], Ch[k], S[k], C[k], SS[k], SC[k], TS[k], TC[k], Wsum, a.trend)
                                                : gls_power<MODE>(Sh[k], Ch[k], S[k], C[k], SS[k], SC[k], YY, Wsum, Werr, a.psd);
            if (a.power) a.power[curve * a.nf + j] = p;
            if (p == p && (best_j < 0 || p > best)) {
                best = p;
                best_j = j;
            }
        }
    }
    if (a.blk_max) {
        // NaN-aware max with lowest-index ties (np.nanargmax): lanes hold ascending index ranges
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_down(best, o, 64);
            const long long oj = __shfl_down(best_j, o, 64);
            if (oj >= 0 && (best_j < 0 || ov > best || (ov == best && oj < best_j))) {
                best = ov;
                best_j = oj;
            }
        }
        if ((tid & 63) == 0) {
            red_v[tid >> 6] = best;
            red_i[tid >> 6] = best_j;
        }
        __syncthreads();
        if (tid == 0) {
            for (int wv = 1; wv < 4; ++wv) {
                if (red_i[wv] >= 0 && (best_j < 0 || red_v[wv] > best)) {
                    best = red_v[wv];
                    best_j = red_i[wv];
                }
            }
            a.blk_max[L] = best;
            a.blk_arg[L] = best_j;
        }
    }
    } while (BAL && bal_u < bal_hi);
}

// Sample parts of gls_scan_kernel (GlsArgs::partial) added up in a fixed order + the epilogue: four lanes
// per frequency (lane q of the quad takes parts q, q + 4, ...; the quad's four sums are then added in a
// fixed tree), 64 frequencies and one block maximum (for the peak reduction) per workgroup.
template <int MODE>
__global__ __launch_bounds__(kBlock) void gls_finish_kernel(GlsArgs a, int parts) {
    __shared__ double red_v[4];
    __shared__ long long red_i[4];
    const int tid = threadIdx.x, q4 = tid & 3;
    const int64_t j = (int64_t)blockIdx.x * (kBlock / 4) + (tid >> 2);
    double best = 0.0;
    long long best_j = -1;
    const bool live = j < a.nf;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (live) {
        if (a.bal_slots) {   // balanced pieces: the runs that cut this frequency's tile
            const int64_t tile = j / a.bal_tile_freqs;
            parts = (int)(bal_slot_of(a, (tile + 1) * a.bal_chunks - 1) - bal_slot_of(a, tile * a.bal_chunks)) + 1;
        }
        for (int z = q4; z < parts; z += 4) {
            const double *in = a.partial + (int64_t)z * 6 * a.nf + j;
#pragma unroll
            for (int q = 0; q < 6; ++q) v[q] += in[q * a.nf];
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {   // (lane ^ 1, then lane ^ 2: the same tree in every quad)
        v[q] += __shfl_xor(v[q], 1, 64);
        v[q] += __shfl_xor(v[q], 2, 64);
    }
    if (live && q4 == 0) {
        const double *sc = a.scal;
        const double p = gls_power<MODE>(v[0], v[1], v[2], v[3], v[4], v[5], sc[0], sc[1], sc[2], a.psd);
        if (a.power) a.power[j] = p;
        if (p == p) {
            best = p;
            best_j = j;
        }
    }
    if (a.blk_max) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_down(best, o, 64);
            const long long oj = __shfl_down(best_j, o, 64);
            if (oj >= 0 && (best_j < 0 || ov > best || (ov == best && oj < best_j))) {
                best = ov;
                best_j = oj;
            }
        }
        if ((tid & 63) == 0) {
            red_v[tid >> 6] = best;
            red_i[tid >> 6] = best_j;
        }
        __syncthreads();
        if (tid == 0) {
            for (int wv = 1; wv < 4; ++wv) {
                if (red_i[wv] >= 0 && (best_j < 0 || red_v[wv] > best)) {
                    best = red_v[wv];
                    best_j = red_i[wv];
                }
            }
            a.blk_max[blockIdx.x] = best;
            a.blk_arg[blockIdx.x] = best_j;
        }
    }
}

// ---- batches on ONE time axis (GLS.bootstrap's shape, spectral.py:140-152) -----------------------------
// sin/cos of a (frequency, sample) pair are the same for every curve of the batch, so they are
// computed once per workgroup and shared: lanes own 64 consecutive frequencies; the workgroup's 16
// waves first fill an LDS tile {s, c, s^2, s c}[sample][lane] for 32 samples (2 samples per wave,
// direct sincos, no recurrence), then each wave accumulates ITS group of RT curves against the
// tile.  A curve's per-sample weights are wave-uniform, so they arrive through the scalar cache
// (s_load from the [sample][curve] table) and feed v_fma_f64 as SGPR operands: 6 fma per (pair,
// curve) - 2 when all weights are equal, where the four weight-only sums are shared too - against
// ~10.5 instructions per pair in gls_scan_kernel.  (v_mfma_f64_16x16x4 would take 64 cycles for the
// 128 pair-curves it can hold with 6 of 8 outputs useful; the 96 v_fma_f64 it replaces take 48.)
constexpr int kShBlock = 1024;
constexpr int kShWaves = kShBlock / 64;
// samples per LDS tile (48 and 64 - fewer fill phases and barriers - measured 2-3 % SLOWER with individual
// weights, equal within noise with equal weights: the barriers are not what the kernel waits for)
#ifndef PDC_SH_CHUNK
#define PDC_SH_CHUNK 32
#endif
constexpr int kShChunk = PDC_SH_CHUNK;
constexpr int kShCurvesW = 8;   // curves per wave, individual weights (48 accumulators)
constexpr int kShCurvesU = 16;  // curves per wave, equal weights (32 + 4 accumulators)

struct SharedArgs {
    const double *tp;    // [n] t - t0
    const double *rw;    // [n][bpad] {w (y - ybar), w}, or [n][bpad] w (y - ybar) for equal weights
    const double *scal;  // [n_curves][4] = {YY, sum w, sum err^-2, t0}
    int64_t n, n_curves, bpad, tiles, groups;
    double f0, delta;
    int64_t j_begin, nf;
    int psd;
    double *power;
    double *blk_max;  // [n_curves][tiles] or nullptr
    int64_t *blk_arg;
};

using w8 = double __attribute__((ext_vector_type(8)));
using cw8 = __attribute__((address_space(4))) w8;

template <bool FIT_MEAN, bool UNI>
__global__ __launch_bounds__(kShBlock) void gls_shared_kernel(SharedArgs a) {
    constexpr int RT = UNI ? kShCurvesU : kShCurvesW;
    constexpr int NW = UNI ? 1 : RT;  // sets of weight-only sums
    constexpr int MODE = FIT_MEAN ? MODE_FIT_MEAN : MODE_NO_MEAN;
    __shared__ double2 trig_sc[kShChunk + 2][64];  // {sin, cos}; + rows for the read-ahead
    __shared__ double2 trig_qq[kShChunk + 2][64];  // {sin^2, sin cos}
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // consecutive logical workgroups of one XCD share a curve group: its weight table stays in L2
    const int64_t G = a.groups * a.tiles;
    const int64_t per_xcd = (G + 7) / 8;
    const int64_t L = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (L >= G) return;
    const int64_t group = L / a.tiles, tile = L - group * a.tiles;
    const int64_t j = tile * 64 + lane;
    const double f = __dadd_rn(a.f0, __dmul_rn((double)(a.j_begin + j), a.delta));  // numpy arange
    const int64_t r0 = (group * kShWaves + wave) * RT;  // first curve of this wave

    double Sh[RT], Ch[RT], S[NW], C[NW], SS[NW], SC[NW];
#pragma unroll
    for (int r = 0; r < RT; ++r) Sh[r] = Ch[r] = 0.0;
#pragma unroll
    for (int r = 0; r < NW; ++r) S[r] = C[r] = SS[r] = SC[r] = 0.0;

    for (int64_t base = 0; base < a.n; base += kShChunk) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kShChunk / kShWaves; ++q) {
            const int il = wave * (kShChunk / kShWaves) + q;
            const int64_t i = base + il;
            const double t = i < a.n ? a.tp[i] : 0.0;
            double s, c;
            sincos_cycles_half(frac_product(f, t), s, c);
            if (i >= a.n) s = c = 0.0;  // padding rows contribute nothing
            trig_sc[il][lane] = make_double2(s, c);
            trig_qq[il][lane] = make_double2(s * s, s * c);
        }
        __syncthreads();
        const int cnt = (int)((a.n - base) < kShChunk ? (a.n - base) : kShChunk);
        // Software pipeline: scalar loads return out of order, so waiting for one waits for all.
        // Per sample the wave needs two 64-byte blocks of weights (4 + 4 curves, or 8 + 8 with
        // equal weights) and one LDS tile row; those of sample il+1 are requested right after the
        // wait for sample il's and before its fmas.  Two samples per trip so that the two register
        // sets swap roles without copies; an odd chunk runs one padding sample (zero tile row, zero
        // weights: the table carries two zeroed rows after the last sample).
        const cw8 *wrow = reinterpret_cast<const cw8 *>(
            reinterpret_cast<uintptr_t>(a.rw + ((base * a.bpad + r0) * (UNI ? 1 : 2))));
        const int64_t wstride = a.bpad * (UNI ? 1 : 2) / 8;  // w8 per sample row
        auto accumulate = [&](const w8 &b0, const w8 &b1, const double2 sc, const double2 qq) {
            if (UNI) {
                if (FIT_MEAN) {
                    S[0] += sc.x;
                    C[0] += sc.y;
                }
                SS[0] += qq.x;
                SC[0] += qq.y;
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const double wy = r < 8 ? b0[r & 7] : b1[r & 7];
                    Sh[r] = __builtin_fma(wy, sc.x, Sh[r]);
                    Ch[r] = __builtin_fma(wy, sc.y, Ch[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const double wy = r < 4 ? b0[2 * (r & 3)] : b1[2 * (r & 3)];
                    const double w = r < 4 ? b0[2 * (r & 3) + 1] : b1[2 * (r & 3) + 1];
                    Sh[r] = __builtin_fma(wy, sc.x, Sh[r]);
                    Ch[r] = __builtin_fma(wy, sc.y, Ch[r]);
                    if (FIT_MEAN) {
                        S[r] = __builtin_fma(w, sc.x, S[r]);
                        C[r] = __builtin_fma(w, sc.y, C[r]);
                    }
                    SS[r] = __builtin_fma(w, qq.x, SS[r]);
                    SC[r] = __builtin_fma(w, qq.y, SC[r]);
                }
            }
        };
        w8 a0 = wrow[0], a1 = wrow[1];
        double2 sca = trig_sc[0][lane], qqa = trig_qq[0][lane];
        for (int il = 0; il < cnt; il += 2) {
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): set A has arrived
            const w8 b0 = wrow[wstride], b1 = wrow[wstride + 1];
            const double2 scb = trig_sc[il + 1][lane], qqb = trig_qq[il + 1][lane];
            __builtin_amdgcn_sched_barrier(0);
            accumulate(a0, a1, sca, qqa);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // set B has arrived
            a0 = wrow[2 * wstride];
            a1 = wrow[2 * wstride + 1];
            sca = trig_sc[il + 2][lane];
            qqa = trig_qq[il + 2][lane];
            __builtin_amdgcn_sched_barrier(0);
            accumulate(b0, b1, scb, qqb);
            __builtin_amdgcn_sched_barrier(0);
            wrow += 2 * wstride;
        }
    }

    const bool valid = j < a.nf;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t b = r0 + r;
        if (b >= a.n_curves) break;  // wave-uniform
        const double *sc = a.scal + b * 4;
        const double YY = sc[0], Wsum = sc[1], Werr = sc[2];
        double s1 = S[UNI ? 0 : r], c1 = C[UNI ? 0 : r], ss = SS[UNI ? 0 : r], sc2 = SC[UNI ? 0 : r];
        if (UNI) {  // w = (1/1)/W for every sample
            const double w0 = 1.0 / Werr;
            s1 *= w0;
            c1 *= w0;
            ss *= w0;
            sc2 *= w0;
        }
        const double p = gls_power<MODE>(Sh[r], Ch[r], s1, c1, ss, sc2, YY, Wsum, Werr, a.psd);
        if (valid && a.power) a.power[b * a.nf + j] = p;
        if (a.blk_max) {
            double best = 0.0;
            long long best_j = -1;
            if (valid && p == p) {
                best = p;
                best_j = j;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_down(best, o, 64);
                const long long oj = __shfl_down(best_j, o, 64);
                if (oj >= 0 && (best_j < 0 || ov > best || (ov == best && oj < best_j))) {
                    best = ov;
                    best_j = oj;
                }
            }
            if (lane == 0) {
                a.blk_max[b * a.tiles + tile] = best;
                a.blk_arg[b * a.tiles + tile] = best_j;
            }
        }
    }
}

// ---- the same, TWO frequencies per lane (round 5; individual weights) --------------------------------------------
// In gls_shared_kernel every scalar load of a sample's weights and every LDS tile row feeds 6 fmas per curve.  Here a
// lane owns frequencies j and j + 64 of a 128-frequency tile: 512 threads, 8 waves x 8 curves, 96 accumulators per lane
// (256 registers), so the two 64-byte weight loads of a sample feed 96 fmas instead of 48 and twice as many fmas stand
// between a request and its wait.  The second frequency's sin / cos come from the first's by ONE plane rotation with
// (cos, sin)(2 pi 64 delta t) - the same for every lane, so 32 lanes compute a chunk's worth one chunk ahead - instead
// of a second software sincos: the tile fill costs 4 sincos + 4 rotations per thread and 3072 fmas.
constexpr int kSh2Block = 512;
constexpr int kSh2Waves = kSh2Block / 64;
constexpr int kSh2Chunk = 32;
constexpr int kSh2Curves = 8;
constexpr size_t kSh2Lds = (size_t)(kSh2Chunk + 2) * 4 * 64 * sizeof(double2);

template <bool FIT_MEAN>
__global__ __launch_bounds__(kSh2Block) void gls_shared2_kernel(SharedArgs a) {
    constexpr int RT = kSh2Curves;
    constexpr int MODE = FIT_MEAN ? MODE_FIT_MEAN : MODE_NO_MEAN;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh2_raw[];
    double2 (*trig)[4][64] = reinterpret_cast<double2 (*)[4][64]>(sh2_raw);   // [sample]{sc_a, qq_a, sc_b, qq_b}[lane]
    __shared__ double2 rot_s[2][kSh2Chunk];                                     // (cos, sin)(2 pi 64 delta t) of a chunk's samples
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const int64_t tiles128 = (a.nf + 127) / 128;
    const int64_t G = a.groups * tiles128;
    const int64_t per_xcd = (G + 7) / 8;
    const int64_t L = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (L >= G) return;
    const int64_t group = L / tiles128, tile = L - group * tiles128;
    const int64_t ja = tile * 128 + lane, jb = ja + 64;
    const double fa = __dadd_rn(a.f0, __dmul_rn((double)(a.j_begin + ja), a.delta));  // numpy arange
    const double d64 = 64.0 * a.delta;
    const int64_t r0 = (group * kSh2Waves + wave) * RT;  // first curve of this wave

    double Sh[2][RT], Ch[2][RT], S[2][RT], C[2][RT], SS[2][RT], SC[2][RT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < RT; ++r) Sh[h][r] = Ch[h][r] = S[h][r] = C[h][r] = SS[h][r] = SC[h][r] = 0.0;

    auto rotation = [&](int64_t i) -> double2 {
        const double t = i < a.n ? a.tp[i] : 0.0;
        double s, c;
        sincos_cycles_half(frac_product(d64, t), s, c);
        return make_double2(c, s);
    };
    if (threadIdx.x < kSh2Chunk) rot_s[0][threadIdx.x] = rotation(threadIdx.x);
    int cur = 0;
    for (int64_t base = 0; base < a.n; base += kSh2Chunk, cur ^= 1) {
        __syncthreads();
        if (threadIdx.x < kSh2Chunk) rot_s[cur ^ 1][threadIdx.x] = rotation(base + kSh2Chunk + threadIdx.x);
#pragma unroll
        for (int q = 0; q < kSh2Chunk / kSh2Waves; ++q) {
            const int il = wave * (kSh2Chunk / kSh2Waves) + q;
            const int64_t i = base + il;
            const double t = i < a.n ? a.tp[i] : 0.0;
            double s, c;
            sincos_cycles_half(frac_product(fa, t), s, c);
            const double2 rt = rot_s[cur][il];
            double s2 = __builtin_fma(s, rt.x, c * rt.y), c2 = __builtin_fma(c, rt.x, -(s * rt.y));
            if (i >= a.n) s = c = s2 = c2 = 0.0;  // padding rows contribute nothing
            trig[il][0][lane] = make_double2(s, c);
            trig[il][1][lane] = make_double2(s * s, s * c);
            trig[il][2][lane] = make_double2(s2, c2);
            trig[il][3][lane] = make_double2(s2 * s2, s2 * c2);
        }
        __syncthreads();
        const int cnt = (int)((a.n - base) < kSh2Chunk ? (a.n - base) : kSh2Chunk);
        const cw8 *wrow = reinterpret_cast<const cw8 *>(reinterpret_cast<uintptr_t>(a.rw + ((base * a.bpad + r0) * 2)));
        const int64_t wstride = a.bpad * 2 / 8;  // w8 per sample row
        auto accumulate = [&](const w8 &b0, const w8 &b1, const double2 (&tv)[4]) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const double wy = r < 4 ? b0[2 * (r & 3)] : b1[2 * (r & 3)];
                const double w = r < 4 ? b0[2 * (r & 3) + 1] : b1[2 * (r & 3) + 1];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double2 sc = tv[2 * h], qq = tv[2 * h + 1];
                    Sh[h][r] = __builtin_fma(wy, sc.x, Sh[h][r]);
                    Ch[h][r] = __builtin_fma(wy, sc.y, Ch[h][r]);
                    if (FIT_MEAN) {
                        S[h][r] = __builtin_fma(w, sc.x, S[h][r]);
                        C[h][r] = __builtin_fma(w, sc.y, C[h][r]);
                    }
                    SS[h][r] = __builtin_fma(w, qq.x, SS[h][r]);
                    SC[h][r] = __builtin_fma(w, qq.y, SC[h][r]);
                }
            }
        };
        // (software pipeline as in gls_shared_kernel: the weights and the tile row of sample il + 1 are requested right
        // after the wait for sample il's and before its 96 fmas; two register sets that swap roles)
        w8 a0 = wrow[0], a1 = wrow[1];
        double2 ta[4], tb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ta[k] = trig[0][k][lane];
        for (int il = 0; il < cnt; il += 2) {
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): set A has arrived
            const w8 b0 = wrow[wstride], b1 = wrow[wstride + 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) tb[k] = trig[il + 1][k][lane];
            __builtin_amdgcn_sched_barrier(0);
            accumulate(a0, a1, ta);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // set B has arrived
            a0 = wrow[2 * wstride];
            a1 = wrow[2 * wstride + 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) ta[k] = trig[il + 2][k][lane];
            __builtin_amdgcn_sched_barrier(0);
            accumulate(b0, b1, tb);
            __builtin_amdgcn_sched_barrier(0);
            wrow += 2 * wstride;
        }
    }

    const int64_t tiles64 = (a.nf + 63) / 64;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t b = r0 + r;
        if (b >= a.n_curves) break;  // wave-uniform
        const double *sc = a.scal + b * 4;
        const double YY = sc[0], Wsum = sc[1], Werr = sc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t j = h ? jb : ja;
            const bool valid = j < a.nf;
            const double p = gls_power<MODE>(Sh[h][r], Ch[h][r], S[h][r], C[h][r], SS[h][r], SC[h][r], YY, Wsum, Werr, a.psd);
            if (valid && a.power) a.power[b * a.nf + j] = p;
            if (a.blk_max && 2 * tile + h < tiles64) {   // (wave-uniform)
                double best = 0.0;
                long long best_j = -1;
                if (valid && p == p) {
                    best = p;
                    best_j = j;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double ov = __shfl_down(best, o, 64);
                    const long long oj = __shfl_down(best_j, o, 64);
                    if (oj >= 0 && (best_j < 0 || ov > best || (ov == best && oj < best_j))) {
                        best = ov;
                        best_j = oj;
                    }
                }
                if (lane == 0) {
                    a.blk_max[b * tiles64 + 2 * tile + h] = best;
                    a.blk_arg[b * tiles64 + 2 * tile + h] = best_j;
                }
            }
        }
    }
}

__global__ __launch_bounds__(64) void gls_peak_kernel(const double *blk_max, const int64_t *blk_arg,
                                                      int64_t tiles, double *amax, int64_t *argmax) {
    const int64_t curve = blockIdx.x;
    double best = 0.0;
    long long best_j = -1;
    for (int64_t i = threadIdx.x; i < tiles; i += 64) {  // ascending per lane
        const double v = blk_max[curve * tiles + i];
        const long long j = blk_arg[curve * tiles + i];
        if (j >= 0 && (best_j < 0 || v > best)) {
            best = v;
            best_j = j;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_down(best, o, 64);
        const long long oj = __shfl_down(best_j, o, 64);
        if (oj >= 0 && (best_j < 0 || ov > best || (ov == best && oj < best_j))) {
            best = ov;
            best_j = oj;
        }
    }
    if (threadIdx.x == 0) {
        if (amax) amax[curve] = best_j >= 0 ? best : __builtin_nan("");
        if (argmax) argmax[curve] = best_j;
    }
}

// Tile shape (K frequencies per thread, S waves per frequency tile).  Cost model: a wave executes
// ~(12 + 8K) VALU instructions per sample and handles 1/S of the samples; the waves on one SIMD
// share its issue slots, so time ~ ceil(waves / 1024) * (12 + 8K) / S (up to a constant): few large
// tiles when the grid fills the chip, many small ones when it does not.  A lone wave per SIMD cannot
// hide its own LDS/dependency stalls (+10 %).  PDC_GLS_K / PDC_GLS_S override for experiments.
// sample parts (GlsArgs::partial): single curves only, parts x nf <= 2^20 cells of six sums (50 MB)
constexpr int64_t kPartialCells = (int64_t)1 << 20;
constexpr int kPartsMax = 512;

void tile_shape(int64_t n_curves, int64_t nf, int64_t n_total, bool may_split, int *K_out, int *S_out, int *Z_out) {
    static const int envK = [] { const char *e = getenv("PDC_GLS_K"); return e ? atoi(e) : 0; }();
    static const int envS = [] { const char *e = getenv("PDC_GLS_S"); return e ? atoi(e) : 0; }();
    static const int envZ = [] { const char *e = getenv("PDC_GLS_Z"); return e ? atoi(e) : 0; }();
    // parts of at least 2048 samples, and no more cells of partial sums than the workspace holds
    int64_t z_max = 1;
    if (may_split && n_curves == 1 && n_total >= 16384) {
        z_max = n_total / 2048;
        const int64_t room = kPartialCells / (nf > 0 ? nf : 1);
        z_max = z_max < room ? z_max : room;
        z_max = z_max > kPartsMax ? kPartsMax : (z_max < 1 ? 1 : z_max);
    }
    double best = 1e300;
    int bk = 8, bs = 1, bz = 1;
    for (int K : {16, 8, 4}) {
        if (envK && K != envK) continue;
        for (int S : {1, 2, 4}) {
            if (envS && S != envS) continue;
            const double tile_waves = (double)n_curves * (double)((nf + 64 * K - 1) / (64 * K)) * S;
            // a short grid over a long curve: cut the samples into parts until every SIMD has ~2 waves
            int64_t Z = (int64_t)__builtin_ceil(2048.0 / tile_waves);
            Z = Z > z_max ? z_max : (Z < 1 ? 1 : Z);
            if (Z >= 8) Z -= Z % 8;   // eight or more parts are dealt to the 8 XCDs: equal shares
            if (envZ) Z = envZ > z_max ? z_max : envZ;
            const double waves = tile_waves * (double)Z;
            const double rounds = __builtin_ceil(waves / 1024.0);
            double cost = rounds * (12.0 + 8.0 * K) / ((double)S * (double)Z);
            if (waves / 1024.0 <= 1.0) cost *= 1.10;
            cost *= 1.0 + 0.02 * (S > 2 ? S - 2 : 2 - S);  // measured: two waves per tile is the sweet spot
            if (Z > 1) cost *= 1.02;                       // (the finishing launch)
            if (cost < best) {
                best = cost;
                bk = K;
                bs = S;
                bz = (int)Z;
            }
        }
    }
    *K_out = bk;
    *S_out = bs;
    *Z_out = bz;
}

template <int MODE, int S>
void launch_scan_ks(int K, dim3 grid, hipStream_t st, const GlsArgs &a) {
    switch (K) {
        case 4: hipLaunchKernelGGL((gls_scan_kernel<4, MODE, S>), grid, dim3(kBlock), 0, st, a); break;
        case 16: hipLaunchKernelGGL((gls_scan_kernel<16, MODE, S>), grid, dim3(kBlock), 0, st, a); break;
        default: hipLaunchKernelGGL((gls_scan_kernel<8, MODE, S>), grid, dim3(kBlock), 0, st, a); break;
    }
}

template <int MODE>
void launch_scan(int K, int S, dim3 grid, hipStream_t st, const GlsArgs &a) {
    switch (S) {
        case 4: launch_scan_ks<MODE, 4>(K, grid, st, a); break;
        case 2: launch_scan_ks<MODE, 2>(K, grid, st, a); break;
        default: launch_scan_ks<MODE, 1>(K, grid, st, a); break;
    }
}

struct WorkLayout {
    int64_t rec, scal, parts, blk_max, blk_arg, partial, total;
};


WorkLayout layout(int64_t n_total, int64_t n_curves, int64_t nf) {
    auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    const int64_t tiles_max = (nf + 63) / 64;  // smallest tile: gls_shared_kernel, 64 frequencies
    WorkLayout w;
    w.rec = 0;
    w.scal = up(n_total * 48);
    w.parts = w.scal + up(n_curves * 32);
    w.blk_max = w.parts + up(4 * kPrepParts * 8);
    w.blk_arg = w.blk_max + up(n_curves * tiles_max * 8);
    w.partial = w.blk_arg + up(n_curves * tiles_max * 8);
    // (grows with nf: a plan sized for a long grid also serves short ones)
    int64_t cells = kPartsMax * nf < kPartialCells ? kPartsMax * nf : kPartialCells;
    cells = cells > 3 * nf ? cells : 3 * nf;   // balanced pieces: up to three partial sums per frequency
    w.total = w.partial + (n_curves == 1 ? up(cells * 6 * 8) : 0);
    return w;
}

// Shared by the GLS and raw (trig_sums) paths.
int scan_dev(int device, hipStream_t st, const double *d_t, const double *d_y, const double *d_dy,
             const int64_t *d_offsets, int64_t n_total, int64_t n_curves, int shared_t, double f0,
             double delta, int64_t j_begin, int64_t nf, int mode, int psd, double *d_power,
             double *d_raw_s, double *d_raw_c, double *d_amax, int64_t *d_argmax, void *work,
             int64_t work_bytes, const int32_t *d_picks = nullptr, const TrendScalars *trend = nullptr) {
    PDC_REQUIRE(d_t && d_y, "gls: t and y must not be NULL");
    PDC_REQUIRE((mode == MODE_TREND) == (trend != nullptr) && (mode != MODE_TREND || (n_curves == 1 && !d_picks && !shared_t)),
                "gls: the trend mode takes one curve and its scalars");
    PDC_REQUIRE(!d_picks || (shared_t && d_offsets && mode != MODE_RAW), "gls: picks need a shared time axis");
    PDC_REQUIRE(n_total >= 0 && n_curves >= 1 && nf >= 0 && j_begin >= 0, "gls: negative size");
    PDC_REQUIRE(n_curves == 1 || d_offsets, "gls: a batch needs offsets");
    PDC_REQUIRE(n_curves * ((nf + 255) / 256 + 1) < (int64_t)1 << 31, "gls: grid too large");
    const WorkLayout w = layout(n_total, n_curves, nf);
    PDC_REQUIRE(work && work_bytes >= w.total, "gls: workspace too small (%lld < %lld bytes)",
                (long long)work_bytes, (long long)w.total);
    if (nf == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    char *base = static_cast<char *>(work);

    PrepArgs p;
    p.t = d_t;
    p.y = d_y;
    p.dy = d_dy;
    p.offsets = d_offsets;
    p.n_total = n_total;
    p.shared_t = shared_t;
    p.fit_mean = mode == MODE_FIT_MEAN;
    p.raw = mode == MODE_RAW;
    p.delta = delta;
    p.rec = reinterpret_cast<double *>(base + w.rec);
    p.scal = reinterpret_cast<double *>(base + w.scal);
    p.picks = d_picks;
    const bool peaks = d_amax || d_argmax;

    // Batches on one time axis: share the trigonometry between curves (gls_shared_kernel) when
    // enough curves fill its curve groups and the [sample][curve] table fits the record area.
    // PDC_GLS_SHARED=0/1 forces the choice (tests).
    static const int env_shared = [] { const char *e = getenv("PDC_GLS_SHARED"); return e ? atoi(e) : -1; }();
    if (shared_t && mode != MODE_RAW && n_curves >= 2 && env_shared != 0) {
        const bool uni = d_dy == nullptr;
        const int64_t n = n_total / n_curves;
        const int64_t group = (int64_t)kShWaves * (uni ? kShCurvesU : kShCurvesW);
        const int64_t bpad = (n_curves + group - 1) / group * group;
        const int64_t table = (n + 2) * bpad * (uni ? 8 : 16);  // + two rows: the kernel reads ahead
        if ((n_curves >= 96 || env_shared == 1) && table + n * 8 <= w.scal && n * n_curves == n_total) {
            p.rw = p.rec;
            p.tp = reinterpret_cast<double *>(base + table);
            p.bpad = bpad;
            PDC_HIP(hipMemsetAsync(base + table / (n + 2) * n, 0, table / (n + 2) * 2, st));
            hipLaunchKernelGGL(gls_prep_kernel, dim3((unsigned)n_curves), dim3(kPrepBlock), 0, st, p);
            PDC_HIP(hipGetLastError());
            SharedArgs sa;
            sa.tp = p.tp;
            sa.rw = p.rw;
            sa.scal = p.scal;
            sa.n = n;
            sa.n_curves = n_curves;
            sa.bpad = bpad;
            sa.tiles = (nf + 63) / 64;
            sa.groups = bpad / group;
            sa.f0 = f0;
            sa.delta = delta;
            sa.j_begin = j_begin;
            sa.nf = nf;
            sa.psd = psd;
            sa.power = d_power;
            sa.blk_max = peaks ? reinterpret_cast<double *>(base + w.blk_max) : nullptr;
            sa.blk_arg = peaks ? reinterpret_cast<int64_t *>(base + w.blk_arg) : nullptr;
            static const bool sh2 = [] { const char *e = getenv("PDC_GLS_SH2"); return !(e && e[0] == '0'); }();
            if (!uni && sh2) {
                // individual weights: two frequencies per lane, 64 curves per workgroup (bpad is a multiple of 128)
                sa.groups = bpad / (kSh2Waves * kSh2Curves);
                const int64_t G2 = sa.groups * ((nf + 127) / 128);
                PDC_REQUIRE(G2 < (int64_t)1 << 31, "gls: grid too large");
                const dim3 grid2((unsigned)(((G2 + 7) / 8) * 8));
                if (mode == MODE_FIT_MEAN) {
                    PDC_TRY(allow_dynamic_lds((const void *)gls_shared2_kernel<true>, (int)kSh2Lds));
                    hipLaunchKernelGGL((gls_shared2_kernel<true>), grid2, dim3(kSh2Block), kSh2Lds, st, sa);
                } else {
                    PDC_TRY(allow_dynamic_lds((const void *)gls_shared2_kernel<false>, (int)kSh2Lds));
                    hipLaunchKernelGGL((gls_shared2_kernel<false>), grid2, dim3(kSh2Block), kSh2Lds, st, sa);
                }
                PDC_HIP(hipGetLastError());
                if (peaks) {
                    hipLaunchKernelGGL(gls_peak_kernel, dim3((unsigned)n_curves), dim3(64), 0, st, sa.blk_max,
                                       sa.blk_arg, sa.tiles, d_amax, d_argmax);
                    PDC_HIP(hipGetLastError());
                }
                return PDC_OK;
            }
            const int64_t G = sa.groups * sa.tiles;
            PDC_REQUIRE(G < (int64_t)1 << 31, "gls: grid too large");
            const dim3 grid((unsigned)(((G + 7) / 8) * 8));
            if (mode == MODE_FIT_MEAN) {
                if (uni) hipLaunchKernelGGL((gls_shared_kernel<true, true>), grid, dim3(kShBlock), 0, st, sa);
                else hipLaunchKernelGGL((gls_shared_kernel<true, false>), grid, dim3(kShBlock), 0, st, sa);
            } else {
                if (uni) hipLaunchKernelGGL((gls_shared_kernel<false, true>), grid, dim3(kShBlock), 0, st, sa);
                else hipLaunchKernelGGL((gls_shared_kernel<false, false>), grid, dim3(kShBlock), 0, st, sa);
            }
            PDC_HIP(hipGetLastError());
            if (peaks) {
                hipLaunchKernelGGL(gls_peak_kernel, dim3((unsigned)n_curves), dim3(64), 0, st, sa.blk_max,
                                   sa.blk_arg, sa.tiles, d_amax, d_argmax);
                PDC_HIP(hipGetLastError());
            }
            return PDC_OK;
        }
    }
    if (n_curves == 1 && mode != MODE_RAW && n_total >= 16384 && !d_picks) {
        WidePrepArgs wp;
        wp.p = p;
        wp.nparts = (int)((n_total + 4 * kBlock - 1) / (4 * kBlock));
        wp.nparts = wp.nparts > kPrepParts ? kPrepParts : wp.nparts;
        wp.part = reinterpret_cast<double *>(base + w.parts);
        hipLaunchKernelGGL(gls_prep_wide_a, dim3(wp.nparts), dim3(kBlock), 0, st, wp);
        hipLaunchKernelGGL(gls_prep_wide_b, dim3(wp.nparts), dim3(kBlock), 0, st, wp);
        hipLaunchKernelGGL(gls_prep_wide_c, dim3(1), dim3(kBlock), 0, st, wp);
    } else {
        hipLaunchKernelGGL(gls_prep_kernel, dim3((unsigned)n_curves), dim3(kPrepBlock), 0, st, p);
    }
    PDC_HIP(hipGetLastError());

    int K, S, parts;
    tile_shape(n_curves, nf, n_total, mode != MODE_RAW, &K, &S, &parts);
    if (mode == MODE_TREND) {   // eight sums per frequency: K <= 8; every tile streams the whole curve (no sample parts)
        K = K > 8 ? 8 : K;
        S = 1;
        parts = 1;
    }
    GlsArgs a;
    if (trend) a.trend = *trend;
    a.rec = p.rec;
    a.offsets = d_offsets;
    a.scal = p.scal;
    a.n_total = n_total;
    a.n_curves = n_curves;
    const int64_t tile_freqs = (int64_t)(kBlock / S) * K;
    a.tiles = (nf + tile_freqs - 1) / tile_freqs;
    a.f0 = f0;
    a.delta = delta;
    a.j_begin = j_begin;
    a.nf = nf;
    a.psd = psd;
    a.power = d_power;
    a.raw_s = d_raw_s;
    a.raw_c = d_raw_c;
    a.blk_max = peaks ? reinterpret_cast<double *>(base + w.blk_max) : nullptr;
    a.blk_arg = peaks ? reinterpret_cast<int64_t *>(base + w.blk_arg) : nullptr;
    const int64_t G = a.n_curves * a.tiles;
    dim3 grid((unsigned)(((G + 7) / 8) * 8));
    // A short grid over a long curve leaves most of the chip idle (nf = 1e4: 40 tiles of four waves for
    // 1024 SIMDs): tile_shape then cuts the samples into parts as well.
    if (parts > 1) {
        const int64_t chunk = 128;
        a.z_len = ((n_total + parts - 1) / parts + chunk - 1) / chunk * chunk;
        parts = (int)((n_total + a.z_len - 1) / a.z_len);
        a.partial = reinterpret_cast<double *>(base + w.partial);
        a.parts = parts;
        if (parts >= 8) {
            a.parts_by_xcd = 1;
            grid.x = (unsigned)(8 * ((parts + 7) / 8) * G);
        } else {
            grid.y = (unsigned)parts;
        }
    }
    // Balanced pieces (one long curve on a long grid, K = 16): a workgroup slot holds 4 waves at 2 waves per
    // SIMD, 2 per CU, 512 on the chip, and the tiles rarely fill their last round of slots - C2's 489 tiles
    // leave 23 slots idle for the whole launch, C4's slabs end with a round of 197.  When that costs more than
    // 2 % the (tile, chunk) space is cut into ceil(tiles / 512) * 512 equal runs instead (GlsArgs::bal_*), every
    // slot gets the same work, and gls_finish_kernel adds the <= 3 pieces of a tile.  PDC_GLS_BAL=0/1 forces it.
    static const int env_bal = [] { const char *e = getenv("PDC_GLS_BAL"); return e ? atoi(e) : -1; }();
    bool balanced = false;
    if (n_curves == 1 && parts == 1 && mode != MODE_RAW && mode != MODE_TREND && K == 16 && S >= 2 && env_bal != 0) {
        const int64_t slots = 512, nchunks = (n_total + 127) / 128;
        const int64_t full = a.tiles / slots, rem = a.tiles % slots;
        // (a slot's partner gone, a workgroup runs alone on its CU at about twice the speed)
        const double t_tiles = (double)full + (rem == 0 ? 0.0 : (rem <= slots / 2 ? 0.5 : 1.0));
        const double t_bal = (double)a.tiles / (double)slots + 0.004;   // (+ the pieces through HBM, the finishing launch)
        const int64_t w_bal = (a.tiles + slots - 1) / slots * slots;
        if ((env_bal == 1 || t_tiles > 1.02 * t_bal) && a.tiles * 2 > w_bal && nchunks >= 32) {
            balanced = true;
            a.bal_slots = w_bal;
            a.bal_chunks = nchunks;
            a.bal_units = a.tiles * nchunks;
            a.bal_tile_freqs = tile_freqs;
            a.partial = reinterpret_cast<double *>(base + w.partial);
            grid = dim3((unsigned)(((w_bal + 7) / 8) * 8));
            if (mode == MODE_FIT_MEAN) {
                if (S == 2) hipLaunchKernelGGL((gls_scan_kernel<16, MODE_FIT_MEAN, 2, true>), grid, dim3(kBlock), 0, st, a);
                else hipLaunchKernelGGL((gls_scan_kernel<16, MODE_FIT_MEAN, 4, true>), grid, dim3(kBlock), 0, st, a);
            } else {
                if (S == 2) hipLaunchKernelGGL((gls_scan_kernel<16, MODE_NO_MEAN, 2, true>), grid, dim3(kBlock), 0, st, a);
                else hipLaunchKernelGGL((gls_scan_kernel<16, MODE_NO_MEAN, 4, true>), grid, dim3(kBlock), 0, st, a);
            }
        }
    }
    if (balanced) {
        // (launched above)
    } else if (mode == MODE_TREND) {
        if (K == 4) hipLaunchKernelGGL((gls_scan_kernel<4, MODE_TREND, 1>), grid, dim3(kBlock), 0, st, a);
        else hipLaunchKernelGGL((gls_scan_kernel<8, MODE_TREND, 1>), grid, dim3(kBlock), 0, st, a);
    } else if (mode == MODE_FIT_MEAN) {
        launch_scan<MODE_FIT_MEAN>(K, S, grid, st, a);
    } else if (mode == MODE_NO_MEAN) {
        launch_scan<MODE_NO_MEAN>(K, S, grid, st, a);
    } else {
        launch_scan<MODE_RAW>(K, S, grid, st, a);
    }
    PDC_HIP(hipGetLastError());
    int64_t peak_tiles = a.tiles;
    if (a.partial) {
        peak_tiles = (nf + kBlock / 4 - 1) / (kBlock / 4);   // 64 frequencies per finishing workgroup
        if (mode == MODE_FIT_MEAN)
            hipLaunchKernelGGL(gls_finish_kernel<MODE_FIT_MEAN>, dim3((unsigned)peak_tiles), dim3(kBlock), 0, st, a, parts);
        else
            hipLaunchKernelGGL(gls_finish_kernel<MODE_NO_MEAN>, dim3((unsigned)peak_tiles), dim3(kBlock), 0, st, a, parts);
        PDC_HIP(hipGetLastError());
    }
    if (peaks) {
        hipLaunchKernelGGL(gls_peak_kernel, dim3((unsigned)n_curves), dim3(64), 0, st, a.blk_max,
                           a.blk_arg, peak_tiles, d_amax, d_argmax);
        PDC_HIP(hipGetLastError());
    }
    return PDC_OK;
}

}  // namespace

extern "C" {

int64_t pdc_gls_work_bytes(int64_t n_total, int64_t n_curves, int64_t nf) {
    if (n_total < 0 || n_curves < 1 || nf < 0) return -1;
    return layout(n_total, n_curves, nf).total;
}

int pdc_gls_scan_dev(int device, void *stream, const double *d_t, const double *d_y,
                     const double *d_dy, const int64_t *d_offsets, int64_t n_total,
                     int64_t n_curves, int shared_t, double f0, double delta, int64_t j_begin,
                     int64_t nf, int fit_mean, int psd, double *d_power, double *d_amax,
                     int64_t *d_argmax, void *work, int64_t work_bytes) {
    PDC_REQUIRE(d_power || d_amax || d_argmax, "gls: no output requested");
    return scan_dev(device, (hipStream_t)stream, d_t, d_y, d_dy, d_offsets, n_total, n_curves,
                    shared_t, f0, delta, j_begin, nf, fit_mean ? MODE_FIT_MEAN : MODE_NO_MEAN, psd,
                    d_power, nullptr, nullptr, d_amax, d_argmax, work, work_bytes);
}

namespace {
__global__ void gls_iota_offsets_kernel(int64_t *offsets, int64_t n, int64_t count) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= count) offsets[b] = b * n;
}
}  // namespace

int64_t pdc_gls_bootstrap_work_bytes(int64_t n, int64_t n_boot, int64_t nf) {
    if (n < 0 || n_boot < 1 || nf < 0) return -1;
    return layout(n * n_boot, n_boot, nf).total + (((n_boot + 1) * 8 + 255) & ~(int64_t)255);
}

int pdc_gls_bootstrap_dev(int device, void *stream, const double *d_t, const double *d_y, const double *d_dy,
                          int64_t n, const int32_t *d_picks, int64_t n_boot, double f0, double delta, int64_t nf,
                          int fit_mean, int psd, double *d_amax, int64_t *d_argmax, void *work,
                          int64_t work_bytes) {
    PDC_REQUIRE(d_t && d_y && d_picks && (d_amax || d_argmax), "gls_bootstrap: NULL argument");
    PDC_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && n_boot >= 1 && nf >= 0, "gls_bootstrap: bad size");
    const int64_t need = pdc_gls_bootstrap_work_bytes(n, n_boot, nf);
    PDC_REQUIRE(work && work_bytes >= need, "gls_bootstrap: workspace too small (%lld < %lld bytes)",
                (long long)work_bytes, (long long)need);
    if (nf == 0) return PDC_OK;
    PDC_TRY(use_device(device));
    const int64_t inner = layout(n * n_boot, n_boot, nf).total;
    int64_t *d_off = reinterpret_cast<int64_t *>(static_cast<char *>(work) + inner);
    hipLaunchKernelGGL(gls_iota_offsets_kernel, dim3((unsigned)((n_boot + 256) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_off, n, n_boot);
    PDC_HIP(hipGetLastError());
    return scan_dev(device, (hipStream_t)stream, d_t, d_y, d_dy, d_off, n * n_boot, n_boot, 1, f0, delta, 0, nf,
                    fit_mean ? MODE_FIT_MEAN : MODE_NO_MEAN, psd, nullptr, nullptr, nullptr, d_amax, d_argmax, work,
                    inner, d_picks);
}

int pdc_gls_scan_batch(const double *t, const double *y, const double *dy, const int64_t *offsets,
                       int64_t n_curves, int shared_t, double f0, double delta, int64_t j_begin,
                       int64_t nf, int fit_mean, int psd, double *power_out, double *amax_out,
                       int64_t *argmax_out, int device) {
    PDC_REQUIRE(t && y && offsets, "gls: t, y and offsets must not be NULL");
    PDC_REQUIRE(n_curves >= 1 && nf >= 0 && j_begin >= 0, "gls: negative size");
    PDC_REQUIRE(power_out || amax_out || argmax_out, "gls: no output requested");
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
    void *d_t, *d_y, *d_dy = nullptr, *d_off, *d_pow = nullptr, *d_amax = nullptr, *d_arg = nullptr,
                     *d_work;
    PDC_TRY(cached(device, SLOT_IN0, n_t * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n_total * 8, &d_y));
    if (dy) PDC_TRY(cached(device, SLOT_IN2, n_total * 8, &d_dy));
    PDC_TRY(cached(device, SLOT_IN3, (n_curves + 1) * 8, &d_off));
    if (power_out) PDC_TRY(cached(device, SLOT_OUT0, n_curves * nf * 8, &d_pow));
    if (amax_out) PDC_TRY(cached(device, SLOT_OUT1, n_curves * 8, &d_amax));
    if (argmax_out) PDC_TRY(cached(device, SLOT_OUT2, n_curves * 8, &d_arg));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_work));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_t, t, n_t * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_y, y, n_total * 8, hipMemcpyHostToDevice, st));
    if (dy) PDC_HIP(hipMemcpyAsync(d_dy, dy, n_total * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_off, offsets, (n_curves + 1) * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(scan_dev(device, st, (double *)d_t, (double *)d_y, (double *)d_dy, (int64_t *)d_off,
                     n_total, n_curves, shared_t, f0, delta, j_begin, nf,
                     fit_mean ? MODE_FIT_MEAN : MODE_NO_MEAN, psd, (double *)d_pow, nullptr, nullptr,
                     (double *)d_amax, (int64_t *)d_arg, d_work, wb));
    if (power_out) PDC_HIP(hipMemcpyAsync(power_out, d_pow, n_curves * nf * 8, hipMemcpyDeviceToHost, st));
    if (amax_out) PDC_HIP(hipMemcpyAsync(amax_out, d_amax, n_curves * 8, hipMemcpyDeviceToHost, st));
    if (argmax_out) PDC_HIP(hipMemcpyAsync(argmax_out, d_arg, n_curves * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

int pdc_gls_scan(const double *t, const double *y, const double *dy, int64_t n, double f0,
                 double delta, int64_t j_begin, int64_t nf, int fit_mean, int psd,
                 double *power_out, int device) {
    PDC_REQUIRE(n >= 0, "gls: negative sample count");
    PDC_REQUIRE(power_out || nf == 0, "gls: power_out is NULL");
    const int64_t offsets[2] = {0, n};
    return pdc_gls_scan_batch(t, y, dy, offsets, 1, 0, f0, delta, j_begin, nf, fit_mean, psd,
                              power_out, nullptr, nullptr, device);
}

// ---- BGLST: Bayesian generalised Lomb-Scargle with linear trend (spectral.py:7,207-208 exports the name; the class
// body upstream is `pass`) ---------------------------------------------------------------------------------------
// scalars[12] = {W, yy, y1, ty, tt, t1, shift, inv_span, prec_a, prec_alpha, prec_beta, log_const}: TrendScalars in
// its own order (the host computes them in fp64 - periodicity_amd/spectral.py:BGLST._scalars).
int pdc_bglst_scan_dev(int device, void *stream, const double *d_t, const double *d_y, const double *d_dy, int64_t n,
                       double f0, double delta, int64_t j_begin, int64_t nf, const double *scalars, double *d_loglik,
                       void *work, int64_t work_bytes) {
    PDC_REQUIRE(scalars && (d_loglik || nf == 0), "bglst: NULL argument");
    PDC_REQUIRE(n >= 4, "bglst: at least four samples (four parameters are marginalised)");
    TrendScalars q;
    q.W = scalars[0]; q.yy = scalars[1]; q.y1 = scalars[2]; q.ty = scalars[3]; q.tt = scalars[4]; q.t1 = scalars[5];
    q.shift = scalars[6]; q.inv_span = scalars[7]; q.prec_a = scalars[8]; q.prec_alpha = scalars[9]; q.prec_beta = scalars[10];
    q.log_const = scalars[11];
    PDC_REQUIRE(q.W > 0.0 && q.prec_a > 0.0 && q.prec_alpha > 0.0 && q.prec_beta > 0.0 && q.inv_span > 0.0,
                "bglst: weights, prior precisions and the time span must be positive");
    return scan_dev(device, (hipStream_t)stream, d_t, d_y, d_dy, nullptr, n, 1, 0, f0, delta, j_begin, nf, MODE_TREND, 0,
                    d_loglik, nullptr, nullptr, nullptr, nullptr, work, work_bytes, nullptr, &q);
}

int pdc_bglst_scan(const double *t, const double *y, const double *dy, int64_t n, double f0, double delta, int64_t j_begin,
                   int64_t nf, const double *scalars, double *loglik_out, int device) {
    PDC_REQUIRE(t && y && scalars && (loglik_out || nf == 0), "bglst: NULL argument");
    PDC_REQUIRE(n >= 0 && nf >= 0 && j_begin >= 0, "bglst: negative size");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    const int64_t wb = pdc_gls_work_bytes(n, 1, nf);
    void *d_t, *d_y, *d_dy = nullptr, *d_out, *d_work;
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_y));
    if (dy) PDC_TRY(cached(device, SLOT_IN2, n * 8, &d_dy));
    PDC_TRY(cached(device, SLOT_OUT0, nf * 8, &d_out));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_work));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_y, y, n * 8, hipMemcpyHostToDevice, st));
    if (dy) PDC_HIP(hipMemcpyAsync(d_dy, dy, n * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(pdc_bglst_scan_dev(device, st, (double *)d_t, (double *)d_y, (double *)d_dy, n, f0, delta, j_begin, nf, scalars,
                               (double *)d_out, d_work, wb));
    PDC_HIP(hipMemcpyAsync(loglik_out, d_out, nf * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

int pdc_trig_sums(const double *t, const double *w, int64_t n, double f0, double delta, int64_t nf,
                  double *S_out, double *C_out, int device) {
    PDC_REQUIRE(t && w && S_out && C_out, "trig_sums: NULL argument");
    PDC_REQUIRE(n >= 0 && nf >= 0, "trig_sums: negative size");
    PDC_TRY(use_device(device));
    DeviceLock lock(device);
    const int64_t wb = pdc_gls_work_bytes(n, 1, nf);
    void *d_t, *d_w, *d_s, *d_c, *d_work;
    PDC_TRY(cached(device, SLOT_IN0, n * 8, &d_t));
    PDC_TRY(cached(device, SLOT_IN1, n * 8, &d_w));
    PDC_TRY(cached(device, SLOT_OUT0, nf * 8, &d_s));
    PDC_TRY(cached(device, SLOT_OUT1, nf * 8, &d_c));
    PDC_TRY(cached(device, SLOT_WORK, wb, &d_work));
    hipStream_t st = nullptr;
    PDC_TRY(host_stream(device, &st));
    PDC_HIP(hipMemcpyAsync(d_t, t, n * 8, hipMemcpyHostToDevice, st));
    PDC_HIP(hipMemcpyAsync(d_w, w, n * 8, hipMemcpyHostToDevice, st));
    PDC_TRY(scan_dev(device, st, (double *)d_t, (double *)d_w, nullptr, nullptr, n, 1, 0, f0, delta,
                     0, nf, MODE_RAW, 0, nullptr, (double *)d_s, (double *)d_c, nullptr, nullptr,
                     d_work, wb));
    PDC_HIP(hipMemcpyAsync(S_out, d_s, nf * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipMemcpyAsync(C_out, d_c, nf * 8, hipMemcpyDeviceToHost, st));
    PDC_HIP(hipStreamSynchronize(st));
    return PDC_OK;
}

}  // extern "C"
