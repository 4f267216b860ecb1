// microbench3.hip -- cost of the table-driven inverse-CDF normal (rmh-rng v2):
// random cells per lane vs identical cells in all lanes (LDS conflict share),
// 4 independent normals per loop trip (ILP), 1 and 2 waves per SIMD.
#include "../riemann_amd/csrc/rmh_mfma.hpp"
#include "../riemann_amd/csrc/rmh_icdf_table.h"
#include <cstdio>
#include <vector>
using namespace rmh;
__device__ double g_tab[kIcdfDoubles];

template <bool SAME, int MODE>
__global__ void k(double *out, int iters)
{
    __shared__ double tab[kIcdfDoubles];
    icdf_stage(tab, g_tab);
    __syncthreads();
    Xo x;
    xo_init(x, 1, SAME ? 7 : threadIdx.x + blockIdx.x * blockDim.x);
    double s = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { // full: xoshiro + icdf x4
            uint64_t w0 = xo_next(x), w1 = xo_next(x), w2 = xo_next(x), w3 = xo_next(x);
            s += normal_from_word(w0, tab) + normal_from_word(w1, tab) + normal_from_word(w2, tab) + normal_from_word(w3, tab);
        } else if (MODE == 1) { // xoshiro only x4
            s += (double)((xo_next(x) ^ xo_next(x) ^ xo_next(x) ^ xo_next(x)) >> 40);
        } else { // icdf only x4 (cheap word source)
            uint64_t w0 = x.s0 * 0x9E3779B97F4A7C15ULL + it;
            x.s0 = w0;
            s += normal_from_word(w0, tab) + normal_from_word(w0 >> 3, tab) + normal_from_word(w0 << 5, tab) + normal_from_word(~w0, tab);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
float run(K kern, int threads, int iters, double *d)
{
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *d; hipMalloc(&d, 256 * 1024 * 8);
    std::vector<double> pad(kIcdfDoubles, 0.0);
    for (int c = 0; c < 448; ++c)
        for (int k = 0; k < 11; ++k) pad[(c / 8) * kIcdfRowStride + (c % 8) * 12 + k] = rmh_icdf_table[c * 11 + k];
    hipMemcpyToSymbol(HIP_SYMBOL(g_tab), pad.data(), pad.size() * 8);
    const int it = 10000;
    for (int threads : {256, 512}) {
        const double ns = 1e6 / (it * 4.0) / (threads / 256);   // ns per normal per wave-slot
        printf("== %d threads (%d wave/SIMD): ns per normal per SIMD\n", threads, threads / 256);
        printf("full, random cells   %.1f\n", run(k<false, 0>, threads, it, d) * ns);
        printf("full, same cell      %.1f\n", run(k<true, 0>, threads, it, d) * ns);
        printf("xoshiro only         %.1f\n", run(k<false, 1>, threads, it, d) * ns);
        printf("icdf only, random    %.1f\n", run(k<false, 2>, threads, it, d) * ns);
        printf("icdf only, same      %.1f\n", run(k<true, 2>, threads, it, d) * ns);
    }
    return 0;
}
