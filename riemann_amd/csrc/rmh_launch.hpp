// rmh_launch.hpp -- launch helper and tile-count dispatch shared by the kernel translation units
#pragma once
#include "rmh_internal.hpp"

namespace rmh {

template <class K, class... A>
static hipError_t launch(K kernel, int64_t nblocks, int nw, size_t lds_bytes, hipStream_t st,
                         A... args)
{
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((unsigned)((nblocks + nw - 1) / nw)), block(64 * nw);
    hipLaunchKernelGGL(kernel, grid, block, lds_bytes, st, args...);
    return hipGetLastError();
}

// k_run_hmc<M, NW, ADAPT, MASS> with the dimension's k-step count as a compile-time constant where it is 4 NT
// (d = 16 NT exactly) or 4 (NT - 1) + 1 (d = 16 (NT - 1) + 1 .. 4): straight-line MFMA chains instead of wave-uniform
// tests in front of the last k-steps of every product (Gaussian MALA at d = 100: 0.067 -> 0.053 ms per iteration,
// d = 52: 0.037 -> 0.027; HMC L = 5 at d = 100: 0.177 -> 0.158).  Include after rmh_run.hpp.
// FULLOK: false for the mixture model (measured: its d = 16 NT variants run 11-15 % SLOWER than the generic kernel, its tail
// variants 10-19 % faster)
template <class M, int NW, bool ADAPT, bool MASS, bool FULLOK = true, class A>
static hipError_t launch_hmc_ks(const A &m, const HmcArgs &p, const ChainArgs &c, const RunArgs &r, size_t lds,
                                hipStream_t st)
{
    constexpr int NT = M::T::S / 4;
    if (!policy(RMH_POLICY_NO_TAIL)) {
        if constexpr (FULLOK)
            if (c.d == 16 * NT)
                return launch(k_run_hmc<M, NW, ADAPT, MASS, false, false, 4 * NT>, c.nblocks, NW, lds, st, m, p, c, r);
        if constexpr (NT >= 2) {
            if (c.ks == 4 * (NT - 1) + 1)
                return launch(k_run_hmc<M, NW, ADAPT, MASS, false, false, 4 * (NT - 1) + 1>, c.nblocks, NW, lds, st, m, p, c, r);
        }
    }
    return launch(k_run_hmc<M, NW, ADAPT, MASS>, c.nblocks, NW, lds, st, m, p, c, r);
}

// the same for k_run_rw<M, NW, ADAPT, PCN, false, LGLOBAL>: FULL at d = 16 NT, KSC = 4 (NT - 1) + 1 at d = 16 (NT - 1) + 1 .. 4
template <class M, int NW, bool ADAPT, bool PCN, bool LGLOBAL, bool FULLOK = true, class A>
static hipError_t launch_rw_ks(const A &m, const RwArgs &p, const ChainArgs &c, const RunArgs &r, size_t lds, hipStream_t st)
{
    constexpr int NT = M::T::S / 4;
    if (!policy(RMH_POLICY_NO_TAIL)) {
        if constexpr (FULLOK)
            if (c.d == 16 * NT)
                return launch(k_run_rw<M, NW, ADAPT, PCN, false, LGLOBAL, true>, c.nblocks, NW, lds, st, m, p, c, r);
        if constexpr (NT >= 2) {
            if (c.ks == 4 * (NT - 1) + 1)
                return launch(k_run_rw<M, NW, ADAPT, PCN, false, LGLOBAL, false, 4 * (NT - 1) + 1>, c.nblocks, NW, lds, st, m, p, c, r);
        }
    }
    return launch(k_run_rw<M, NW, ADAPT, PCN, false, LGLOBAL>, c.nblocks, NW, lds, st, m, p, c, r);
}

#define RMH_NT_DISPATCH(NT, ...)                     \
    switch (NT) {                                    \
    case 1: { constexpr int kNT = 1; __VA_ARGS__; }  \
    case 2: { constexpr int kNT = 2; __VA_ARGS__; }  \
    case 3: { constexpr int kNT = 3; __VA_ARGS__; }  \
    case 4: { constexpr int kNT = 4; __VA_ARGS__; }  \
    case 5: { constexpr int kNT = 5; __VA_ARGS__; }  \
    case 6: { constexpr int kNT = 6; __VA_ARGS__; }  \
    case 7: { constexpr int kNT = 7; __VA_ARGS__; }  \
    default: return hipErrorInvalidValue;            \
    }

} // namespace rmh
