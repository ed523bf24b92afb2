// Micro-benchmark: random gathers from a small L2-resident array on gfx950, the access pattern of a
// sort-by-phase (StringLength): one 1024-thread workgroup per CU (LDS-limited, as the scan kernel),
// indices from an LDS-resident random permutation, UNROLL gathers in flight per lane.
// Reports ns per wave-gather-instruction per CU and the implied cycles per gathered line.
// Build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate ; run: ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kBlock = 1024;
constexpr int kN = 50000;

typedef double Rec __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ double take(const T &v);
template <> __device__ __forceinline__ double take<double>(const double &v) { return v; }
template <> __device__ __forceinline__ double take<Rec>(const Rec &v) { return v.x + v.y; }
template <> __device__ __forceinline__ double take<float>(const float &v) { return (double)v; }

// MODE 0: random gather through the permutation; 1: coalesced (lane-consecutive) loads; 2: nontemporal gather
template <typename T, int UNROLL, int MODE>
__global__ __launch_bounds__(kBlock) void gather_kernel(const T *__restrict__ src, const unsigned short *perm,
                                                         double *out, int n, int reps) {
    extern __shared__ unsigned short order[];
    for (int i = threadIdx.x; i < n; i += kBlock) order[i] = perm[i];
    __syncthreads();
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int base = 0; base + kBlock * UNROLL <= n; base += kBlock * UNROLL) {
            unsigned idx[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int s = base + u * kBlock + threadIdx.x;
                idx[u] = MODE == 1 ? (unsigned)((s + r) % n) : (unsigned)order[(s + r * 7) % n];
            }
            T v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if constexpr (MODE == 2) v[u] = __builtin_nontemporal_load(&src[idx[u]]);
                else v[u] = src[idx[u]];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += take<T>(v[u]);
        }
    }
    out[blockIdx.x * kBlock + threadIdx.x] = acc;
}


template <typename T, int UNROLL, int MODE>
void run(const char *name, const T *d_src, const unsigned short *d_perm, double *d_out) {
    const int blocks = 256, reps = 40;
    const size_t lds = 150 * 1024;  // forces one workgroup per CU like the scan kernel
    hipFuncSetAttribute((const void *)gather_kernel<T, UNROLL, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    gather_kernel<T, UNROLL, MODE><<<blocks, kBlock, lds>>>(d_src, d_perm, d_out, kN, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather_kernel<T, UNROLL, MODE><<<blocks, kBlock, lds>>>(d_src, d_perm, d_out, kN, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_thread = (double)(kN / (kBlock * UNROLL)) * UNROLL * reps;  // gathers per lane
    const double wave_instr_per_cu = per_thread * (kBlock / 64);
    const double ns = ms * 1e6 / wave_instr_per_cu;
    printf("%-44s unroll=%d: %8.3f ms  %7.1f ns per wave-gather per CU = %6.1f cyc @2.4GHz (%.2f cyc per lane-access), %.1f Gaccess/s chip\n",
           name, UNROLL, ms, ns, ns * 2.4, ns * 2.4 / 64, per_thread * kBlock * blocks / ms / 1e6);
}

int main() {
    std::vector<unsigned short> perm(kN);
    for (int i = 0; i < kN; ++i) perm[i] = (unsigned short)i;
    srand(1);
    for (int i = kN - 1; i > 0; --i) {
        int j = rand() % (i + 1);
        std::swap(perm[i], perm[j]);
    }
    std::vector<double> t(kN);
    std::vector<Rec> rec(kN);
    std::vector<float> f(kN);
    for (int i = 0; i < kN; ++i) {
        t[i] = i * 1.0;
        rec[i] = Rec{(double)i, 0.5};
        f[i] = (float)i;
    }
    unsigned short *d_perm;
    double *d_t, *d_out;
    Rec *d_rec;
    float *d_f;
    hipMalloc(&d_perm, kN * 2);
    hipMalloc(&d_t, kN * 8);
    hipMalloc(&d_rec, kN * 16);
    hipMalloc(&d_f, kN * 4);
    hipMalloc(&d_out, 256 * kBlock * 8);
    hipMemcpy(d_perm, perm.data(), kN * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_t, t.data(), kN * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_rec, rec.data(), kN * 16, hipMemcpyHostToDevice);
    hipMemcpy(d_f, f.data(), kN * 4, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        run<double, 4, 0>("random gather 8 B from 400 KB", d_t, d_perm, d_out);
        run<double, 8, 0>("random gather 8 B from 400 KB", d_t, d_perm, d_out);
        run<Rec, 4, 0>("random gather 16 B from 800 KB (AoS t,m)", d_rec, d_perm, d_out);
        run<Rec, 8, 0>("random gather 16 B from 800 KB (AoS t,m)", d_rec, d_perm, d_out);
        run<float, 4, 0>("random gather 4 B from 200 KB", d_f, d_perm, d_out);
        run<float, 8, 0>("random gather 4 B from 200 KB", d_f, d_perm, d_out);
        run<double, 4, 2>("random gather 8 B, nontemporal", d_t, d_perm, d_out);
        run<Rec, 4, 2>("random gather 16 B, nontemporal", d_rec, d_perm, d_out);
        run<double, 4, 1>("coalesced 8 B", d_t, d_perm, d_out);
        run<Rec, 4, 1>("coalesced 16 B", d_rec, d_perm, d_out);
    }
    return 0;
}
