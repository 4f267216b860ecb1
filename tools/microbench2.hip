// microbench2.hip -- does v_mfma_f64_16x16x4_f64 overlap with VALU work of the
// OTHER wavefront on the same SIMD?  512-thread workgroups: waves 0-3 run MFMAs,
// waves 4-7 run a VALU loop of the given flavour (f64 fma / f32 fma / int).
#include "../riemann_amd/csrc/rmh_mfma.hpp"
#include <cstdio>
using namespace rmh;

template <int KIND, bool DO_MFMA, bool DO_VALU>
__global__ void __launch_bounds__(512) k(double *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    double r = 0;
    if (wave < 4) {
        if (DO_MFMA) {
            double4_t acc[2] = {double4_t{0, 0, 0, 0}, double4_t{0, 0, 0, 0}};
            double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q & 1] = mfma(a, b, acc[q & 1]);
            }
            r = acc[0].x + acc[1].y;
        }
    } else if (DO_VALU) {
        if (KIND == 0) { // f64 fma
            double a[8];
            for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 128; ++q) a[q & 7] = __builtin_fma(a[q & 7], 1.0000001, 1e-9);
            }
            for (int i = 0; i < 8; ++i) r += a[i];
        } else if (KIND == 1) { // f32 fma
            float a[8];
            for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 256; ++q) a[q & 7] = __builtin_fmaf(a[q & 7], 1.0000001f, 1e-9f);
            }
            for (int i = 0; i < 8; ++i) r += a[i];
        } else if (KIND == 2) { // integer (xoshiro)
            Xo x;
            xo_init(x, 1, threadIdx.x);
            uint64_t s = 0;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 8; ++q) s ^= xo_next(x);
            }
            r = (double)s;
        } else { // integer + fma mix
            Xo x;
            xo_init(x, 1, threadIdx.x);
            for (int it = 0; it < iters; ++it) r = __builtin_fma(r, 0.5, (double)(xo_next(x) >> 40));
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <class K>
float run(K kern, int iters, double *d)
{
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *d; hipMalloc(&d, 256 * 512 * 8);
    const int it = 10000;
    const char *names[4] = {"f64 fma x128", "f32 fma x256", "xoshiro x8", "xoshiro + fma"};
    float m = run(k<0, true, false>, it, d);
    printf("mfma alone (16/iter): %.3f ms\n", m);
#define ROW(K) { float v = run(k<K, false, true>, it, d), b = run(k<K, true, true>, it, d); \
                 printf("%-16s alone %.3f ms, with mfma %.3f ms  (sum %.3f, max %.3f)\n", names[K], v, b, v + m, v > m ? v : m); }
    ROW(0) ROW(1) ROW(2) ROW(3)
    return 0;
}
