// microbench.hip -- per-instruction costs that size the kernels:
//   v_mfma_f64_16x16x4_f64 issue interval (independent / dependent accumulators),
//   one Box-Muller pair of the rmh-rng stream, fp64 FMA throughput,
//   each at 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include "../riemann_amd/csrc/rmh_mfma.hpp"
#include "../riemann_amd/csrc/rmh_icdf_table.h"
#include <cstdio>
#include <vector>
using namespace rmh;

template <int NACC>
__global__ void k_mfma(double *out, int iters)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k % NACC] = mfma(a, b, acc[k % NACC]);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (16.0 * iters);
}

__device__ double g_tab[kIcdfDoubles];

__global__ void k_pair(double *out, int iters)
{
    __shared__ double tab[kIcdfDoubles];
    icdf_stage(tab, g_tab);
    __syncthreads();
    Xo x;
    xo_init(x, 1, threadIdx.x + blockIdx.x * blockDim.x);
    double s = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const double z0 = normal_icdf(x, tab), z1 = normal_icdf(x, tab);
        s += z0 * z1;
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / iters;
}

__global__ void k_fma(double *out, int iters)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const double b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k % 8] = __builtin_fma(a[k % 8], b, c);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (32.0 * iters);
}

// MFMA and RNG in the same wave, interleaved by the compiler
__global__ void k_mix(double *out, int iters)
{
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{0, 0, 0, 0};
    __shared__ double tab[kIcdfDoubles];
    icdf_stage(tab, g_tab);
    __syncthreads();
    Xo x;
    xo_init(x, 1, threadIdx.x);
    double a = threadIdx.x * 1e-3, s = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const double z0 = normal_icdf(x, tab), z1 = normal_icdf(x, tab);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k % 4] = mfma(a, s, acc[k % 4]);
        s += z0 * z1;
    }
    long long t1 = clock64();
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / iters;
}

template <class K>
double run(K k, int threads, int iters, double *d)
{
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double v; hipMemcpy(&v, d, 8, hipMemcpyDeviceToHost);
    printf("  clk/item (wave 0) %9.1f   kernel %8.3f ms\n", v, ms);
    return v;
}

int main()
{
    double *d; hipMalloc(&d, 256 * 1024 * 8);
    {
        std::vector<double> pad(kIcdfDoubles, 0.0);
        for (int c = 0; c < 448; ++c)
            for (int k = 0; k < 11; ++k) pad[(c / 8) * kIcdfRowStride + (c % 8) * 12 + k] = rmh_icdf_table[c * 11 + k];
        hipMemcpyToSymbol(HIP_SYMBOL(g_tab), pad.data(), pad.size() * 8);
    }
    const int it = 20000;
    for (int threads : {256, 512}) {
        printf("== %d threads/WG, 1 WG/CU (%d wave(s)/SIMD)\n", threads, threads / 256);
        printf("mfma f64 16x16x4, 1 accumulator (dependent):\n"); run(k_mfma<1>, threads, it, d);
        printf("mfma f64 16x16x4, 2 accumulators:\n"); run(k_mfma<2>, threads, it, d);
        printf("mfma f64 16x16x4, 4 accumulators:\n"); run(k_mfma<4>, threads, it, d);
        printf("2 inverse-CDF normals (rmh-rng v2):\n"); run(k_pair, threads, it, d);
        printf("v_fma_f64, 8 independent chains:\n"); run(k_fma, threads, it, d);
        printf("1 pair + 16 mfma per iteration (same wave):\n"); run(k_mix, threads, it / 4, d);
    }
    return 0;
}
