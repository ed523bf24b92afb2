// Micro-benchmark: does the v_fma_f64 issue rate on gfx950 depend on where the operands live?
//   sss: acc = fma(acc, S, S)        one VGPR source (the accumulator), two scalar operands (as fp64_rate.hip)
//   vs : acc = fma(V, S, acc)        two VGPR sources, one SGPR     (sums fed by a wave-uniform weight)
//   vv : acc = fma(V1, V2, acc)      three VGPR sources             (what gls_scan_kernel issues today)
// 16 independent accumulators per lane, 2 waves per SIMD, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 fp64_operands.hip -o fp64_operands ; run: ./fp64_operands
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(double *out, const double *in, double a, double b, int iters) {
    double x[16], v1[4], v2[4];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = threadIdx.x * 1e-3 + c;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v1[c] = in[threadIdx.x + 64 * c];
        v2[c] = in[threadIdx.x + 64 * c + 256];
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (MODE == 0) x[c] = __builtin_fma(x[c], a, b);
                if (MODE == 1) x[c] = __builtin_fma(v1[c & 3], a, x[c]);
                if (MODE == 2) x[c] = __builtin_fma(v1[c & 3], v2[(c >> 2) & 3], x[c]);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, double *out, double *in) {
    const int blocks = 512, iters = 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, in, 0.999999, 1e-9, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, in, 0.999999, 1e-9, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fmas = (double)blocks * 256 * iters * 8.0 * 16;
    printf("%s: %.3f ms, %.1f TFLOP/s, %.3f ns per wave-instr per SIMD\n", name, ms, fmas * 2 / ms / 1e9,
           ms * 1e6 / (fmas / 64 / 1024));
}

int main() {
    double *out, *in;
    hipMalloc(&out, sizeof(double) * 512 * 256);
    hipMalloc(&in, sizeof(double) * 1024);
    double h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 1.0 + 1e-7 * ((i * 2654435761u) % 1000);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("sss (acc, scalar, scalar)", out, in);
    run<1>("vs  (vgpr, scalar, acc)  ", out, in);
    run<2>("vv  (vgpr, vgpr, acc)    ", out, in);
    return 0;
}
