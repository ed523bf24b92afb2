/* TEST INFRASTRUCTURE: checks on the CPU the division-free quotient of csrc/stringlength.hip
 * (fast::exact_quotient): y = RN(1/p); q0 = RN(t y); two fma corrections; against the IEEE t / p,
 * on random and adversarial operands (few-bit significands, exact multiples, significands next to
 * all-ones and next to a power of two).  Prints the number of mismatches of the two-step form (must
 * be 0) and of the one-step form (informational).   gcc -O2 -mfma -ffp-contract=off */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline double quot2(double t, double p, double y) {
    double q0 = t * y, r0 = fma(-p, q0, t), q1 = fma(r0, y, q0), r1 = fma(-p, q1, t);
    return fma(r1, y, q1);
}
static inline double quot1(double t, double p, double y) {
    double q0 = t * y, r0 = fma(-p, q0, t);
    return fma(r0, y, q0);
}
static uint64_t s = 88172645463325252ULL;
static inline uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double rd(int emin, int emax) {
    uint64_t m = rnd() & 0xFFFFFFFFFFFFFULL;
    int e = emin + (int)(rnd() % (uint64_t)(emax - emin + 1));
    uint64_t b = ((uint64_t)(e + 1023) << 52) | m;
    double d;
    memcpy(&d, &b, 8);
    return (rnd() & 1) ? d : -d;
}
static double with_frac(uint64_t frac, int e) {
    uint64_t b = ((uint64_t)(1023 + e) << 52) | frac;
    double d;
    memcpy(&d, &b, 8);
    return d;
}
int main(int argc, char **argv) {
    long n = argc > 1 ? atol(argv[1]) : 20000000L, bad2 = 0, bad1 = 0;
    for (long it = 0; it < n; ++it) {
        double p, t;
        switch (it & 7) {
            case 0: case 1: case 2: p = rd(-3, 20); t = rd(-10, 30); break;
            case 3: p = with_frac((rnd() & 0xFFF) << 40, (int)(rnd() % 8)); t = (double)(int64_t)(rnd() % 100000); break;
            case 4: p = rd(0, 8); t = p * (double)(rnd() % 4096); break;
            case 5: p = (double)(1 + rnd() % 1000) / (double)(1 + rnd() % 64); t = (double)(rnd() % 1000000) / (double)(1 + rnd() % 128); break;
            case 6: p = with_frac(0xFFFFFFFFFFFFFULL - (rnd() % 64 + 1), (int)(rnd() % 8)); t = rd(-5, 20); break;
            default: p = with_frac(rnd() % 64, (int)(rnd() % 8)); t = rd(-5, 20); break;
        }
        const double y = 1.0 / p, q = t / p;
        if (quot2(t, p, y) != q) ++bad2;
        if (quot1(t, p, y) != q) ++bad1;
    }
    printf("pairs %ld two_step_mismatches %ld one_step_mismatches %ld\n", n, bad2, bad1);
    return bad2 != 0;
}
