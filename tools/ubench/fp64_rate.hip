// Micro-benchmark: sustained v_fma_f64 issue rate on gfx950 (independent chains, no memory).
// Build: hipcc --offload-arch=gfx950 -O3 fp64_rate.hip -o fp64_rate ; run: ./fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ __launch_bounds__(256) void fma_kernel(double *out, double a, double b, int iters) {
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    fma_kernel<CHAINS><<<blocks, 256>>>(out, 0.999999, 1e-9, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fma_kernel<CHAINS><<<blocks, 256>>>(out, 0.999999, 1e-9, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fmas = (double)blocks * 256 * iters * 8.0 * CHAINS;
    const double tf = fmas * 2 / ms / 1e9;
    // wave-instructions per SIMD per second -> cycles per instruction at 2.4 GHz nominal
    const double wave_instr_per_simd = fmas / 64 / 1024;
    printf("chains=%2d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s fp64, %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n",
           CHAINS, blocks_per_cu, ms, tf, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
    hipFree(out);
}

int main() {
    for (int bpc : {1, 2, 4}) {
        run<1>(bpc, 20000);
        run<2>(bpc, 20000);
        run<4>(bpc, 20000);
        run<8>(bpc, 20000);
        run<16>(bpc, 10000);
    }
    return 0;
}
