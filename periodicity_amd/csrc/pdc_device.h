// Device-side helpers shared by the kernels of libperiodicity_hip.so (not installed).  Kept apart from
// pdc_internal.h (host-side runtime declarations) so that the source hashes profiles/*_pmc_summary.json
// records for a kernel change only when device code does.
#pragma once
#include <hip/hip_runtime.h>

namespace pdc {

// sin(2 pi r), cos(2 pi r) for r given in CYCLES, |r| <= ~0.5 (any finite r works; accuracy is
// that of the reduction r - q/4, exact in fp64).  gfx950 has no fp64 transcendental unit, so this
// is a polynomial on [-pi/4, pi/4] (degree 13 / 14 minimax, the classic fdlibm kernel
// coefficients) plus a quadrant fix-up done with integer ops on the high word.
__device__ __forceinline__ void sincos_cycles(double r, double &s, double &c) {
    const double q = __builtin_rint(4.0 * r);
    const double z = __builtin_fma(-0.25, q, r);           // exact: |z| <= 1/8 cycle
    const double x = z * 6.283185307179586476925;          // radians, |x| <= pi/4
    const double x2 = x * x;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, x2, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, x2, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, x2, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, x2, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, x2, -1.66666666666666324348e-01);
    const double sx = __builtin_fma(x * x2, ps, x);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, x2, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, x2, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, x2, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, x2, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, x2, 4.16666666666666019037e-02);
    const double cx = __builtin_fma(x2 * x2, pc, __builtin_fma(-0.5, x2, 1.0));
    const int qi = (int)q;
    const bool swap = qi & 1;
    double s1 = swap ? cx : sx;
    double c1 = swap ? sx : cx;
    // quadrant 1: (c, -s)  2: (-s, -c)  3: (-c, s)
    const unsigned long long sflip = (unsigned long long)(qi & 2) << 62;
    const unsigned long long cflip = (unsigned long long)((qi + 1) & 2) << 62;
    s = __longlong_as_double(__double_as_longlong(s1) ^ sflip);
    c = __longlong_as_double(__double_as_longlong(c1) ^ cflip);
}

// The same function with its twelve polynomial coefficients handed in ({sin, cos} coefficient of each Horner step,
// highest degree first; sincos_coefficient(i) lists them) instead of written as literals.  A VOP3 instruction reads
// ONE scalar operand, so literal coefficients live in SGPR pairs - and the head of each chain in a VGPR pair - that
// the compiler hoists out of every loop: in gls_scan_kernel<16, ..., BAL> (192 accumulators, SGPRs full of record
// fields) those hoisted pairs were the 17 spilled VGPRs of round 5.  Kept in LDS and read where the table fill
// needs them, they are live for the fill only.  Same constants, same fmas, same bits.
__device__ __forceinline__ double2 sincos_coefficient(int i) {
    switch (i) {
        case 0: return make_double2(1.58969099521155010221e-10, -1.13596475577881948265e-11);
        case 1: return make_double2(-2.50507602534068634195e-08, 2.08757232129817482790e-09);
        case 2: return make_double2(2.75573137070700676789e-06, -2.75573143513906633035e-07);
        case 3: return make_double2(-1.98412698298579493134e-04, 2.48015872894767294178e-05);
        case 4: return make_double2(8.33333333332248946124e-03, -1.38888888888741095749e-03);
        default: return make_double2(-1.66666666666666324348e-01, 4.16666666666666019037e-02);
    }
}
__device__ __forceinline__ void sincos_cycles_k(double r, const double2 (&k)[6], double &s, double &c) {
    const double q = __builtin_rint(4.0 * r);
    const double z = __builtin_fma(-0.25, q, r);
    const double x = z * 6.283185307179586476925;
    const double x2 = x * x;
    double ps = k[0].x, pc = k[0].y;
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        ps = __builtin_fma(ps, x2, k[i].x);
        pc = __builtin_fma(pc, x2, k[i].y);
    }
    const double sx = __builtin_fma(x * x2, ps, x);
    const double cx = __builtin_fma(x2 * x2, pc, __builtin_fma(-0.5, x2, 1.0));
    const int qi = (int)q;
    const bool swap = qi & 1;
    const double s1 = swap ? cx : sx;
    const double c1 = swap ? sx : cx;
    const unsigned long long sflip = (unsigned long long)(qi & 2) << 62;
    const unsigned long long cflip = (unsigned long long)((qi + 1) & 2) << 62;
    s = __longlong_as_double(__double_as_longlong(s1) ^ sflip);
    c = __longlong_as_double(__double_as_longlong(c1) ^ cflip);
}

// Hot-loop variant for |r| <= 0.5 cycle (always true after frac_product): evaluate at a quarter of
// the angle, where |a| <= pi/4 needs no quadrant logic, then double the angle twice
// (sin 2a = 2 s c, cos 2a = 1 - 2 s^2).  All fp64 VALU, no integer/select instructions; the two
// doublings cost 8 ops and keep the absolute error at a few 1e-16.
__device__ __forceinline__ void sincos_cycles_half(double r, double &s, double &c) {
    const double x = r * 1.570796326794896619231;          // (2 pi r) / 4
    // Polynomials summed term by term over explicit powers of x^2 (not Horner): every step is an
    // accumulate-into-self fma with a constant source, which hipcc emits as one v_fmac_f64 with
    // no register copy (Horner needs the constant in the destination: a v_mov_b64 per step), and
    // the two chains are short and independent.
    const double x2 = x * x;
    const double x4 = x2 * x2;
    const double x6 = x4 * x2;
    const double x8 = x4 * x4;
    const double x10 = x8 * x2;
    double ps = __builtin_fma(8.33333333332248946124e-03, x2, -1.66666666666666324348e-01);
    ps = __builtin_fma(-1.98412698298579493134e-04, x4, ps);
    ps = __builtin_fma(2.75573137070700676789e-06, x6, ps);
    ps = __builtin_fma(-2.50507602534068634195e-08, x8, ps);
    ps = __builtin_fma(1.58969099521155010221e-10, x10, ps);
    const double s1 = __builtin_fma(x * x2, ps, x);
    double pc = __builtin_fma(-1.38888888888741095749e-03, x2, 4.16666666666666019037e-02);
    pc = __builtin_fma(2.48015872894767294178e-05, x4, pc);
    pc = __builtin_fma(-2.75573143513906633035e-07, x6, pc);
    pc = __builtin_fma(2.08757232129817482790e-09, x8, pc);
    pc = __builtin_fma(-1.13596475577881948265e-11, x10, pc);
    const double c1 = __builtin_fma(x4, pc, __builtin_fma(-0.5, x2, 1.0));
    // angle doubling, twice
    double sc = s1 * c1;
    const double c2 = __builtin_fma(-2.0 * s1, s1, 1.0);
    const double s2 = sc + sc;
    sc = s2 * c2;
    c = __builtin_fma(-2.0 * s2, s2, 1.0);
    s = sc + sc;
}

// frac(a*b) in (-0.5, 0.5] cycles with the product carried exactly (fma error term), so the
// phase stays accurate to ~1e-16 cycle however many whole cycles a*b spans.
__device__ __forceinline__ double frac_product(double a, double b) {
    const double hi = a * b;
    const double lo = __builtin_fma(a, b, -hi);
    return (hi - __builtin_rint(hi)) + lo;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Deterministic block-wide sum for 256-thread blocks; result valid in every thread.
__device__ __forceinline__ double block_sum_256(double v, double *lds4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// Deterministic block-wide sum for BLOCK-thread blocks (BLOCK a multiple of 64, <= 1024).
template <int BLOCK>
__device__ __forceinline__ double block_sum(double v, double *lds_waves) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds_waves[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) r += lds_waves[w];
    return r;
}

}  // namespace pdc
