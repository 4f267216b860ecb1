// microbench4.hip -- issue cost of the vector instructions the kernels are priced in, one wave per
// SIMD and two (256 / 512-thread workgroups, one workgroup per CU): cycles per instruction for a
// stream of independent instructions of ONE kind (8 destination registers in rotation), read with
// clock64 by wave 0.  hipcc --offload-arch=gfx950 -O3 tools/microbench4.hip -o tools/microbench4
#include <hip/hip_runtime.h>
#include <cstdio>

#define RMH_BENCH(NAME, DECL, ASM)                                                     \
    __global__ void k_##NAME(double *out, int iters)                                   \
    {                                                                                  \
        DECL;                                                                          \
        long long t0 = clock64();                                                      \
        for (int it = 0; it < iters; ++it) {                                           \
            _Pragma("unroll") for (int k = 0; k < 64; ++k) { ASM; }                    \
        }                                                                              \
        long long t1 = clock64();                                                      \
        double s = 0;                                                                  \
        for (int i = 0; i < 8; ++i) s += d[i] + (double)u[i];                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (64.0 * iters); \
    }

#define REGS                                                                           \
    double d[8];                                                                       \
    unsigned u[8];                                                                     \
    unsigned long long q[8];                                                           \
    for (int i = 0; i < 8; ++i) {                                                      \
        d[i] = 1.0 + threadIdx.x * 1e-3 + i;                                           \
        u[i] = threadIdx.x * 7 + i;                                                    \
        q[i] = (unsigned long long)threadIdx.x * 0x9E3779B97F4A7C15ULL + i;            \
    }                                                                                  \
    double b = 1.0000001, c = 1e-9;                                                    \
    (void)q; (void)b; (void)c

RMH_BENCH(fma_f64, REGS, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k & 7]) : "v"(b), "v"(c)))
RMH_BENCH(add_f64, REGS, asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k & 7]) : "v"(c)))
RMH_BENCH(mul_f64, REGS, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k & 7]) : "v"(b)))
RMH_BENCH(min_f64, REGS, asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[k & 7]) : "v"(b)))
RMH_BENCH(ldexp_f64, REGS, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d[k & 7])))
RMH_BENCH(fract_f64, REGS, asm volatile("v_fract_f64_e32 %0, %0" : "+v"(d[k & 7])))
RMH_BENCH(floor_f64, REGS, asm volatile("v_floor_f64_e32 %0, %0" : "+v"(d[k & 7])))
RMH_BENCH(cvt_i32_f64, REGS, asm volatile("v_cvt_i32_f64_e32 %0, %1" : "=v"(u[k & 7]) : "v"(d[k & 7])))
RMH_BENCH(cvt_f64_u32, REGS, asm volatile("v_cvt_f64_u32_e32 %0, %1" : "=v"(d[k & 7]) : "v"(u[k & 7])))
RMH_BENCH(cmp_f64, REGS, asm volatile("v_cmp_lt_f64_e32 vcc, %0, %1" : : "v"(d[k & 7]), "v"(b) : "vcc"))
RMH_BENCH(cndmask, REGS, asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7]) : "vcc"))
RMH_BENCH(mov_b64, REGS, asm volatile("v_mov_b64_e32 %0, %1" : "=v"(d[k & 7]) : "v"(d[(k + 1) & 7])))
RMH_BENCH(mul_lo_u32, REGS, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7])))
RMH_BENCH(mul_u32_u24, REGS, asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7])))
RMH_BENCH(mad_u32_u24, REGS, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7])))
RMH_BENCH(add_u32, REGS, asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7])))
RMH_BENCH(xor_b32, REGS, asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7])))
RMH_BENCH(alignbit, REGS, asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7])))
RMH_BENCH(lshl_b64, REGS, asm volatile("v_lshlrev_b64 %0, 17, %0" : "+v"(q[k & 7])))
RMH_BENCH(lshl_add_u64, REGS, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[k & 7]) : "v"(q[(k + 1) & 7])))
RMH_BENCH(ffbh_u32, REGS, asm volatile("v_ffbh_u32_e32 %0, %0" : "+v"(u[k & 7])))
RMH_BENCH(bfe_u32, REGS, asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(u[k & 7])))
RMH_BENCH(lshl_add_u32, REGS, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7])))
RMH_BENCH(permlane16_swap, REGS, asm volatile("v_permlane16_swap_b32_e32 %0, %1" : "+v"(u[k & 7]), "+v"(u[(k + 4) & 7])))
RMH_BENCH(fmac_f64_dpp, REGS,
          asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[k & 7]) : "v"(b), "v"(c)))
// one MFMA then N f64 fmas, repeated: does VALU work hide under the matrix instruction?
template <int N>
__global__ void k_mfma_plus(double *out, int iters)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double d[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) {
        d[i] = 1.0 + threadIdx.x * 1e-3 + i;
        u[i] = threadIdx.x + i;
    }
    double b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc[k & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[k & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (N > 0 && N < 100) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j & 7]) : "v"(b), "v"(c));
                if (N >= 100) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(u[j & 7]) : "v"(u[(j + 1) & 7]));
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + u[i];
    for (int i = 0; i < 4; ++i) s += acc[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (16.0 * iters);
}

// NM MFMAs then NV vector instructions of one kind per iteration: at two waves per SIMD, does the
// partner's vector work hide under this wave's MFMAs?  KIND 0 v_xor_b32, 1 v_fma_f64, 2 v_mul_u32_u24,
// 3 v_lshlrev_b64, 4 v_add_co_u32 + v_addc_co_u32 pairs, 5 v_lshlrev_b32, 6 v_and_b32, 7 v_bfe_u32, 8 v_cndmask (sgpr mask)
template <int NM, int NV, int KIND>
__global__ void k_mix(double *out, int iters)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double d[8];
    unsigned u[8];
    unsigned long long q[8];
    for (int i = 0; i < 8; ++i) {
        d[i] = 1.0 + threadIdx.x * 1e-3 + i;
        u[i] = threadIdx.x + i;
        q[i] = threadIdx.x * 77ULL + i;
    }
    double b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NM; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[k & 3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (KIND == 0) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(u[j & 7]) : "v"(u[(j + 1) & 7]));
            if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j & 7]) : "v"(b), "v"(c));
            if (KIND == 2) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u[j & 7]) : "v"(u[(j + 1) & 7]));
            if (KIND == 3) asm volatile("v_lshlrev_b64 %0, 17, %0" : "+v"(q[j & 7]));
            if (KIND == 4) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, %1, %3, vcc"
                                        : "+v"(u[j & 7]), "+v"(u[(j + 4) & 7]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 5) & 7]) : "vcc");
            if (KIND == 5) asm volatile("v_lshlrev_b32_e32 %0, 3, %0" : "+v"(u[j & 7]));
            if (KIND == 6) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(u[j & 7]) : "v"(u[(j + 1) & 7]));
            if (KIND == 7) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(u[j & 7]));
            if (KIND == 8) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u[j & 7]) : "v"(u[(j + 1) & 7]));
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += d[i] + u[i] + (double)q[i];
    for (int i = 0; i < 4; ++i) s += acc[i].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

template <class K>
static void run(const char *name, K k, int threads, int iters, double *d)
{
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double v;
    hipMemcpy(&v, d, 8, hipMemcpyDeviceToHost);
    printf("%-22s %4d thr  clk/instr (wave 0) %7.2f   kernel %8.3f ms\n", name, threads, v, ms);
}

int main()
{
    double *d;
    hipMalloc(&d, 256 * 1024 * 8);
    const int it = 4000;
    for (int threads : {256, 512}) {
#define R(NAME) run(#NAME, k_##NAME, threads, it, d)
        R(fma_f64); R(add_f64); R(mul_f64); R(min_f64); R(ldexp_f64); R(fract_f64); R(floor_f64);
        R(cvt_i32_f64); R(cvt_f64_u32); R(cmp_f64); R(cndmask); R(mov_b64); R(mul_lo_u32); R(mul_u32_u24);
        R(mad_u32_u24); R(add_u32); R(xor_b32); R(alignbit); R(lshl_b64); R(lshl_add_u64);
        R(ffbh_u32); R(bfe_u32); R(lshl_add_u32); R(permlane16_swap); R(fmac_f64_dpp);
        run("8 mfma + 64 xor", k_mix<8, 64, 0>, threads, it, d);
        run("8 mfma + 64 fma_f64", k_mix<8, 64, 1>, threads, it, d);
        run("8 mfma + 64 mul_u24", k_mix<8, 64, 2>, threads, it, d);
        run("8 mfma + 64 lshl_b64", k_mix<8, 64, 3>, threads, it, d);
        run("8 mfma + 64 add64", k_mix<8, 64, 4>, threads, it, d);
        run("8 mfma + 64 lshl_b32", k_mix<8, 64, 5>, threads, it, d);
        run("8 mfma + 64 and_b32", k_mix<8, 64, 6>, threads, it, d);
        run("8 mfma + 64 bfe", k_mix<8, 64, 7>, threads, it, d);
        run("8 mfma + 64 cndmask", k_mix<8, 64, 8>, threads, it, d);
        run("0 mfma + 64 xor", k_mix<0, 64, 0>, threads, it, d);
        run("0 mfma + 64 add64", k_mix<0, 64, 4>, threads, it, d);
        run("0 mfma + 64 lshl_b32", k_mix<0, 64, 5>, threads, it, d);
        run("0 mfma + 64 cndmask", k_mix<0, 64, 8>, threads, it, d);
        run("8 mfma + 0", k_mix<8, 0, 0>, threads, it, d);
        run("mfma + 0 fma", k_mfma_plus<0>, threads, it, d);
        run("mfma + 4 fma_f64", k_mfma_plus<4>, threads, it, d);
        run("mfma + 8 fma_f64", k_mfma_plus<8>, threads, it, d);
        run("mfma + 8 xor_b32", k_mfma_plus<108>, threads, it, d);
        run("mfma + 16 xor_b32", k_mfma_plus<116>, threads, it, d);
    }
    return 0;
}
