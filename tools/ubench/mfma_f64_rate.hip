// Micro-benchmark: v_mfma_f64_16x16x4_f64 rate on gfx950, alone and interleaved with independent
// v_fma_f64 (does the matrix pipe run beside the vector fp64 pipe, or do they share the DP units?).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate ; run: ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

// MF mfma chains and VF fma chains per loop trip (8 trips unrolled)
template <int MF, int VF>
__global__ __launch_bounds__(256) void mix_kernel(double *out, double a, double b, int iters) {
    d4 acc[MF > 0 ? MF : 1];
    double x[VF > 0 ? VF : 1];
    for (int c = 0; c < (MF > 0 ? MF : 1); ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
    for (int c = 0; c < (VF > 0 ? VF : 1); ++c) x[c] = threadIdx.x * 1e-3 + c;
    const double av = a + threadIdx.x * 1e-6, bv = b - threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < MF; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < VF; ++c) x[c] = __builtin_fma(x[c], a, b);
        }
    }
    double s = 0;
    for (int c = 0; c < MF; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    for (int c = 0; c < VF; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MF, int VF>
void run(int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mix_kernel<MF, VF><<<blocks, 256>>>(out, 0.999999, 1e-9, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix_kernel<MF, VF><<<blocks, 256>>>(out, 0.999999, 1e-9, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double mfma = waves * iters * 4.0 * MF, vfma = waves * iters * 4.0 * VF;
    const double tf_m = mfma * 1024 * 2 / ms / 1e9, tf_v = vfma * 64 * 2 / ms / 1e9;
    const double per_simd = waves / 1024.0;   // waves per SIMD
    printf("mfma/trip=%d fma/trip=%2d waves/SIMD=%d: %8.3f ms  MFMA %6.1f TF + VALU %6.1f TF = %6.1f TF fp64;"
           " cycles per trip per SIMD @2.4GHz %.1f\n",
           MF, VF, blocks_per_cu, ms, tf_m, tf_v, tf_m + tf_v, ms * 1e-3 * 2.4e9 / (per_simd * iters * 4.0));
    hipFree(out);
}

int main() {
    for (int bpc : {1, 2, 4}) {
        run<1, 0>(bpc, 4000);
        run<2, 0>(bpc, 4000);
        run<4, 0>(bpc, 2000);
        run<0, 16>(bpc, 4000);
        run<1, 4>(bpc, 4000);
        run<1, 8>(bpc, 4000);
        run<1, 12>(bpc, 4000);
        run<1, 16>(bpc, 4000);
        run<2, 16>(bpc, 4000);
        run<2, 32>(bpc, 2000);
    }
    return 0;
}
