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
