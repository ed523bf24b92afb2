// Micro-benchmark: LDS atomic / scattered-store rates on gfx950 for the counting-sort phases of the
// StringLength kernel: 1024-thread workgroups, one per CU (LDS-limited), every lane updates a
// pseudo-random 32-bit counter out of NB.
// Build: hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip -o lds_atomic_rate ; run: ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kBlock = 1024;

// MODE 0: ds_add_u32 (no return)   1: ds_add_rtn_u32   2: ds_write_b32 scatter   3: ds_write_b16 scatter
// 4: ds_add_u32, conflict-free addresses (lane-consecutive)   5: ds_add_rtn conflict-free
template <int MODE, int NB>
__global__ __launch_bounds__(kBlock) void k(unsigned *out, int iters) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < NB; i += kBlock) lds[i] = 0u;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            const unsigned b = (MODE >= 4) ? ((threadIdx.x + u * 64 + it) % NB) : ((x >> 8) % NB);
            if (MODE == 0 || MODE == 4) atomicAdd(&lds[b], 1u);
            else if (MODE == 1 || MODE == 5) acc += atomicAdd(&lds[b], 1u);
            else if (MODE == 2) lds[b] = x;
            else reinterpret_cast<unsigned short *>(lds)[b * 2] = (unsigned short)x;
        }
    }
    __syncthreads();
    out[blockIdx.x * kBlock + threadIdx.x] = acc + lds[threadIdx.x % NB];
}

template <int MODE, int NB>
void run(const char *name, unsigned *d_out) {
    const int blocks = 256, iters = 2000;
    const size_t lds = 150 * 1024;
    hipFuncSetAttribute((const void *)k<MODE, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, NB><<<blocks, kBlock, lds>>>(d_out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NB><<<blocks, kBlock, lds>>>(d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_ops_per_cu = (double)iters * 8 * (kBlock / 64);
    const double ns = ms * 1e6 / wave_ops_per_cu;
    printf("%-46s NB=%6d: %8.3f ms  %6.1f ns per wave-op per CU = %6.1f cyc @2.4GHz\n", name, NB, ms, ns, ns * 2.4);
}

int main() {
    unsigned *d_out;
    hipMalloc(&d_out, 256 * kBlock * 4);
    run<0, 2048>("ds_add_u32 random", d_out);
    run<0, 8192>("ds_add_u32 random", d_out);
    run<0, 32768>("ds_add_u32 random", d_out);
    run<1, 2048>("ds_add_rtn_u32 random", d_out);
    run<1, 32768>("ds_add_rtn_u32 random", d_out);
    run<2, 2048>("ds_write_b32 random", d_out);
    run<2, 32768>("ds_write_b32 random", d_out);
    run<3, 16384>("ds_write_b16 random", d_out);
    run<4, 2048>("ds_add_u32 lane-consecutive", d_out);
    run<5, 2048>("ds_add_rtn_u32 lane-consecutive", d_out);
    return 0;
}
