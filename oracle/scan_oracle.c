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
