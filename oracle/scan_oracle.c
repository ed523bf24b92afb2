/*
 * TEST INFRASTRUCTURE — plain-C restatement of the reference's scan arithmetic, for parity
 * cases too large for the numpy oracle (oracle/scan_oracle.py is the primary statement; this
 * file is checked against it in tests/test_oracle_golden.py).  Never linked into the product.
 *
 * Paths cited are relative to /root/reference/src/periodicity/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* S_j = sum_i h_i sin(2 pi f_j t_i), C_j likewise with cos: the definition at spectral.py:13-15,
 * evaluated directly in x87 80-bit long double. */
void oracle_trig_sums_exact(const double *t, const double *h, int64_t n, const double *freq,
                            int64_t nf, double *S, double *C) {
    const long double two_pi = 2.0L * acosl(-1.0L);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < nf; ++j) {
        const long double w = two_pi * (long double)freq[j];
        long double s = 0.0L, c = 0.0L;
        for (int64_t i = 0; i < n; ++i) {
            const long double ph = w * (long double)t[i];
            s += (long double)h[i] * sinl(ph);
            c += (long double)h[i] * cosl(ph);
        }
        S[j] = (double)s;
        C[j] = (double)c;
    }
}

/* numpy's float remainder for a positive divisor of 1 (Python modulo, phase.py:131,
 * core.py:544): fmod, then shift negative results up by the divisor. */
static double mod1(double q) {
    double r = fmod(q, 1.0);
    if (r != 0.0) {
        if (r < 0.0) r += 1.0;
    } else {
        r = copysign(0.0, 1.0);
    }
    return r;
}

/* PDM._pdm (phase.py:128-149) for every period.  The argsort at :132-134 only permutes the
 * inputs of order-independent masks, so bins are filled in sample order; the variance is the
 * two-pass mean-centred form of np.var(ddof=1). */
void oracle_pdm_scan(const double *t, const double *x, int64_t n, const double *periods,
                     int64_t n_periods, int nb, int nc, double sigma, double *theta) {
    const int m0 = nb * nc;
#pragma omp parallel
    {
        double *phi = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 4)
        for (int64_t p = 0; p < n_periods; ++p) {
            for (int64_t i = 0; i < n; ++i) phi[i] = mod1(t[i] / periods[p]);
            double num = 0.0;
            int64_t n_sum = 0, good = 0;
            for (int k = 0; k < m0; ++k) {
                const double lo = (double)k / m0, hi = (double)(k + nc) / m0;
                const double wrap = (double)(k - (m0 - nc)) / m0;
                int64_t cnt = 0;
                double mean = 0.0;
                for (int64_t i = 0; i < n; ++i)
                    if ((phi[i] >= lo && phi[i] < hi) || phi[i] < wrap) {
                        ++cnt;
                        mean += x[i];
                    }
                if (cnt > 1) {
                    mean /= (double)cnt;
                    double ss = 0.0;
                    for (int64_t i = 0; i < n; ++i)
                        if ((phi[i] >= lo && phi[i] < hi) || phi[i] < wrap) {
                            const double d = x[i] - mean;
                            ss += d * d;
                        }
                    num += ss; /* (n_j - 1) * s_j */
                    n_sum += cnt;
                    ++good;
                }
            }
            theta[p] = (num / (double)(n_sum - good)) / sigma;
        }
        free(phi);
    }
}

typedef struct {
    double phi, m;
    int64_t idx;
} sl_item;

static int sl_cmp(const void *a, const void *b) {
    const sl_item *x = (const sl_item *)a, *y = (const sl_item *)b;
    if (x->phi < y->phi) return -1;
    if (x->phi > y->phi) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx); /* stable: ties keep time order */
}

/* StringLength._stringlength (phase.py:45-51) through TSeries.fold (core.py:543-544) and the
 * stable sort by phase of the TSeries constructor (core.py:473-477): closed polygon, closing
 * segment not phase-wrapped. */
void oracle_stringlength_scan(const double *t, const double *m, int64_t n, const double *periods,
                              int64_t n_periods, double *ell) {
#pragma omp parallel
    {
        sl_item *it = (sl_item *)malloc(sizeof(sl_item) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 4)
        for (int64_t p = 0; p < n_periods; ++p) {
            for (int64_t i = 0; i < n; ++i) {
                it[i].phi = mod1((t[i] - 0.0) / periods[p]);
                it[i].m = m[i];
                it[i].idx = i;
            }
            qsort(it, (size_t)n, sizeof(sl_item), sl_cmp);
            double sum = 0.0;
            for (int64_t i = 0; i < n; ++i) {
                const int64_t k = (i + 1 == n) ? 0 : i + 1;
                sum += hypot(it[k].m - it[i].m, it[k].phi - it[i].phi);
            }
            ell[p] = sum;
        }
        free(it);
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Double-precision direct sums for EXHAUSTIVE parity at the BASELINE sizes (1e11 ... 1e12 pairs), where the
 * long-double evaluation above would take the better part of an hour.  Same definition (spectral.py:13-15:
 * S_j = sum_i h_i sin(2 pi f_j t_i), C_j with cos), evaluated pair by pair - no recurrence over frequencies or
 * samples, nothing shared with the device kernel's rotation scheme:
 *   phase in CYCLES  p = f t  with its rounding error recovered exactly, e = fma(f, t, -p); k = nearest integer
 *   (p - k is exact), r = (p - k) + e in [-1/2, 1/2] carries ONE rounding (the long-double path rounds a phase of
 *   up to 1e6 rad to 64 bits: this is the tighter of the two at large phases);
 *   quadrant n = round(4 r), z = r - n/4 (exact), a = 2 pi z in [-pi/4, pi/4] with 2 pi split in two doubles;
 *   sin a, cos a from the Cephes double-precision minimax polynomials for that interval (S. Moshier, sin.c:
 *   sincof / coscof, |error| < 1.1e-16), put back in their quadrant by swaps and sign flips;
 *   sin / cos of the doubled frequency by the double-angle identities on those two values (one more rounding
 *   each; a second reduction of 2p bought nothing measurable and cost 40 % of the run time).
 * Sums: blocks of 512 samples in double (SIMD lanes, `omp simd reduction`), blocks added in long double.
 * tests/test_oracle_golden.py pins it to oracle_trig_sums_exact (<= 1e-13 of sum |h|); the GPU suite re-proves
 * that on the sampled bins of each full-size config before it trusts it on all of them.
 * ---------------------------------------------------------------------------------------------------------- */
#define ORACLE_CLONES __attribute__((target_clones("avx512f", "fma", "default")))

static inline void sincos_cycles(double p, double e, double *s_out, double *c_out) {
    const double big = 6755399441055744.0; /* 1.5 * 2^52: adding and subtracting rounds to nearest integer */
    const double k = (p + big) - big;
    const double r = (p - k) + e;
    const double n4 = (4.0 * r + big) - big;
    const double z = r - 0.25 * n4;
    const double two_pi_hi = 6.283185307179586, two_pi_lo = 2.4492935982947064e-16;
    const double a = fma(z, two_pi_lo, z * two_pi_hi);
    const double zz = a * a;
    double ps = 1.58962301576546568060E-10;
    ps = ps * zz + -2.50507477628578072866E-8;
    ps = ps * zz + 2.75573136213857245213E-6;
    ps = ps * zz + -1.98412698295895385996E-4;
    ps = ps * zz + 8.33333333332211858878E-3;
    ps = ps * zz + -1.66666666666666307295E-1;
    const double sz = a + a * zz * ps;
    double pc = -1.13585365213876817300E-11;
    pc = pc * zz + 2.08757008419747316778E-9;
    pc = pc * zz + -2.75573141792967388112E-7;
    pc = pc * zz + 2.48015872888517045348E-5;
    pc = pc * zz + -1.38888888888730564116E-3;
    pc = pc * zz + 4.16666666666665929218E-2;
    const double cz = 1.0 - 0.5 * zz + zz * zz * pc;
    const double an = fabs(n4);
    const int swap = an == 1.0;
    const double s = swap ? cz : sz, c = swap ? sz : cz;
    *s_out = (n4 < -0.5 || an > 1.5) ? -s : s;
    *c_out = (n4 > 0.5 || n4 < -1.5) ? -c : c;
}

/* six sums of one frequency over a run of samples, ADDED to `acc`: (hy, h) at f and h at 2 f. */
ORACLE_CLONES
void oracle_gls_sums_f64_acc(const double *t, const double *hy, const double *h, int64_t n, double f,
                             long double *acc /* Sh Ch S C S2 C2 */) {
    for (int64_t b = 0; b < n; b += 512) {
        const int64_t m = (n - b < 512) ? n - b : 512;
        const double *tb = t + b, *hb = h + b, *yb = hy + b;
        double sh = 0, ch = 0, s1 = 0, c1 = 0, s2 = 0, c2 = 0;
#pragma omp simd reduction(+ : sh, ch, s1, c1, s2, c2)
        for (int64_t i = 0; i < m; ++i) {
            const double p = f * tb[i];
            const double e = fma(f, tb[i], -p);
            double s, c;
            sincos_cycles(p, e, &s, &c);
            const double sd = 2.0 * s * c, cd = (c - s) * (c + s); /* doubled frequency: exact identities, 1 ulp */
            sh += yb[i] * s;
            ch += yb[i] * c;
            s1 += hb[i] * s;
            c1 += hb[i] * c;
            s2 += hb[i] * sd;
            c2 += hb[i] * cd;
        }
        acc[0] += sh; acc[1] += ch; acc[2] += s1; acc[3] += c1; acc[4] += s2; acc[5] += c2;
    }
}

/* The grid is handed over as the doubles the caller scans (np.arange's own values or f0 + delta j), like
 * oracle_trig_sums_exact.  Tiles of 32 frequencies walk the samples in runs of 4096 (a run stays in the core's
 * L2 while the tile's frequencies pass over it); a frequency's additions happen in sample order whatever the
 * tiling, so the result does not depend on it or on the thread count. */
void oracle_gls_sums_f64(const double *t, const double *hy, const double *h, int64_t n, const double *freq,
                         int64_t nf, double *Sh, double *Ch, double *S, double *C, double *S2, double *C2) {
    enum { TILE = 32, RUN = 4096 };
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t j0 = 0; j0 < nf; j0 += TILE) {
        const int64_t nj = (nf - j0 < TILE) ? nf - j0 : TILE;
        long double acc[TILE][6];
        memset(acc, 0, sizeof acc);
        for (int64_t b = 0; b < n; b += RUN) {
            const int64_t m = (n - b < RUN) ? n - b : RUN;
            for (int64_t j = 0; j < nj; ++j)
                oracle_gls_sums_f64_acc(t + b, hy + b, h + b, m, freq[j0 + j], acc[j]);
        }
        for (int64_t j = 0; j < nj; ++j) {
            Sh[j0 + j] = (double)acc[j][0]; Ch[j0 + j] = (double)acc[j][1];
            S[j0 + j] = (double)acc[j][2];  C[j0 + j] = (double)acc[j][3];
            S2[j0 + j] = (double)acc[j][4]; C2[j0 + j] = (double)acc[j][5];
        }
    }
}

/* A batch of curves on one grid (C3: 4096 x 2000 samples x 5e4 frequencies): curve b owns samples
 * [offsets[b], offsets[b+1]); the six sums land in rows of nf.  One parallel region over (curve, tile) - a
 * 2000-sample curve is far too little work to spread over 256 threads by itself. */
void oracle_gls_sums_f64_batch(const double *t, const double *hy, const double *h, const int64_t *offsets,
                               int64_t n_curves, const double *freq, int64_t nf, double *Sh, double *Ch, double *S,
                               double *C, double *S2, double *C2) {
    enum { TILE = 32, RUN = 4096 };
    const int64_t tiles = (nf + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t work = 0; work < n_curves * tiles; ++work) {
        const int64_t b = work / tiles, j0 = (work % tiles) * TILE;
        const int64_t nj = (nf - j0 < TILE) ? nf - j0 : TILE;
        const int64_t lo = offsets[b], n = offsets[b + 1] - offsets[b];
        long double acc[TILE][6];
        memset(acc, 0, sizeof acc);
        for (int64_t r = 0; r < n; r += RUN) {
            const int64_t m = (n - r < RUN) ? n - r : RUN;
            for (int64_t j = 0; j < nj; ++j)
                oracle_gls_sums_f64_acc(t + lo + r, hy + lo + r, h + lo + r, m, freq[j0 + j], acc[j]);
        }
        for (int64_t j = 0; j < nj; ++j) {
            const int64_t o = b * nf + j0 + j;
            Sh[o] = (double)acc[j][0]; Ch[o] = (double)acc[j][1];
            S[o] = (double)acc[j][2];  C[o] = (double)acc[j][3];
            S2[o] = (double)acc[j][4]; C2[o] = (double)acc[j][5];
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * The scans the reference only names (phase.py:11-15, TODO upstream: PARITY UNPINNED BY THE REFERENCE) in C, for
 * whole BASELINE-size period grids: literal restatements of scan_oracle.py's aov_theta / cond_entropy /
 * gl_log_s (the published formulas), checked against them in tests/test_oracle_golden.py.
 * ---------------------------------------------------------------------------------------------------------- */
double lgamma_r(double, int *);

/* scan_oracle.py:_phase_bins - searchsorted(arange(r + 1) / r, phi, "right") - 1, phi == 1.0 joins the last
 * bin, NaN belongs to none. */
static int phase_bin(double t, double period, int r) {
    const double phi = mod1(t / period);
    if (phi != phi) return -1;
    int g = (int)(phi * r);
    if (g > r - 1) g = r - 1;
    if (g < 0) g = 0;
    while (g > 0 && phi < (double)g / r) --g;
    while (g + 1 <= r - 1 && phi >= (double)(g + 1) / r) ++g;
    return g;
}

/* Schwarzenberg-Czerny 1989, eq. 1-3 (scan_oracle.py:aov_theta): two passes, bin means then residuals. */
void oracle_aov_scan(const double *t, const double *x, int64_t n, const double *periods, int64_t n_periods,
                     int n_bins, double *theta) {
#pragma omp parallel
    {
        int *k = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        double *sum = (double *)malloc(sizeof(double) * (size_t)n_bins);
        int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_bins);
#pragma omp for schedule(dynamic, 4)
        for (int64_t p = 0; p < n_periods; ++p) {
            int64_t ok = 0;
            long double total = 0.0L;
            for (int b = 0; b < n_bins; ++b) { sum[b] = 0.0; cnt[b] = 0; }
            for (int64_t i = 0; i < n; ++i) {
                k[i] = phase_bin(t[i], periods[p], n_bins);
                if (k[i] >= 0) { ++ok; total += x[i]; sum[k[i]] += x[i]; ++cnt[k[i]]; }
            }
            if (ok <= n_bins || n_bins < 2) { theta[p] = NAN; continue; }
            const double xbar = (double)(total / (long double)ok);
            double between = 0.0;
            for (int b = 0; b < n_bins; ++b)
                if (cnt[b]) {
                    sum[b] /= (double)cnt[b]; /* now the bin mean */
                    between += (double)cnt[b] * (sum[b] - xbar) * (sum[b] - xbar);
                }
            long double within = 0.0L;
            for (int64_t i = 0; i < n; ++i)
                if (k[i] >= 0) { const double d = x[i] - sum[k[i]]; within += d * d; }
            theta[p] = (between / (n_bins - 1)) / ((double)within / (double)(ok - n_bins));
        }
        free(k); free(sum); free(cnt);
    }
}

/* Graham et al. 2013, eq. 1 (scan_oracle.py:cond_entropy): H_c = sum p(m, phi) ln(p(phi) / p(m, phi)). */
void oracle_cond_entropy_scan(const double *t, const double *mag_bin, int64_t n, const double *periods,
                              int64_t n_periods, int n_phase, int n_mag, double *h) {
#pragma omp parallel
    {
        int64_t *cells = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_phase * n_mag));
#pragma omp for schedule(dynamic, 4)
        for (int64_t p = 0; p < n_periods; ++p) {
            memset(cells, 0, sizeof(int64_t) * (size_t)(n_phase * n_mag));
            int64_t ok = 0;
            for (int64_t i = 0; i < n; ++i) {
                const int k = phase_bin(t[i], periods[p], n_phase);
                if (k >= 0) { ++ok; ++cells[k * n_mag + (int)mag_bin[i]]; }
            }
            if (ok == 0) { h[p] = NAN; continue; }
            double acc = 0.0;
            for (int k = 0; k < n_phase; ++k) {
                int64_t row = 0;
                for (int m = 0; m < n_mag; ++m) row += cells[k * n_mag + m];
                for (int m = 0; m < n_mag; ++m) {
                    const int64_t c = cells[k * n_mag + m];
                    if (c > 0) acc += (double)c / (double)ok * log((double)row / (double)c);
                }
            }
            h[p] = acc;
        }
        free(cells);
    }
}

/* Gregory & Loredo 1992, eq. 5.13-5.14 averaged over the bin offset (scan_oracle.py:gl_log_s). */
void oracle_gl_scan(const double *t, int64_t n, const double *periods, int64_t n_periods, int m, int n_offsets,
                    double *log_s) {
    const int fine = m * n_offsets;
#pragma omp parallel
    {
        int64_t *counts = (int64_t *)malloc(sizeof(int64_t) * (size_t)fine);
        double *terms = (double *)malloc(sizeof(double) * (size_t)n_offsets);
        int sign;
#pragma omp for schedule(dynamic, 4)
        for (int64_t p = 0; p < n_periods; ++p) {
            memset(counts, 0, sizeof(int64_t) * (size_t)fine);
            int64_t ok = 0;
            for (int64_t i = 0; i < n; ++i) {
                const int k = phase_bin(t[i], periods[p], fine);
                if (k >= 0) { ++ok; ++counts[k]; }
            }
            if (ok == 0) { log_s[p] = NAN; continue; }
            double top = -INFINITY;
            for (int off = 0; off < n_offsets; ++off) {
                double lg = 0.0;
                for (int b = 0; b < m; ++b) {
                    int64_t per_bin = 0;
                    for (int q = 0; q < n_offsets; ++q) per_bin += counts[(b * n_offsets + q + off) % fine];
                    lg += lgamma_r((double)per_bin + 1.0, &sign);
                }
                terms[off] = (double)ok * log((double)m) + lg - lgamma_r((double)ok + 1.0, &sign);
                if (terms[off] > top) top = terms[off];
            }
            double s = 0.0;
            for (int off = 0; off < n_offsets; ++off) s += exp(terms[off] - top);
            log_s[p] = top + log(s / n_offsets);
        }
        free(counts); free(terms);
    }
}
