// Raw device -> pinned-host bandwidth of the box (what rmh_history_download can reach at best):
// hipMemcpyAsync of 256 MiB and 1 GiB blocks from device memory into hipHostMalloc memory.
//   hipcc --offload-arch=gfx950 -O2 tools/pcie_raw.hip -o tools/pcie_raw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

int main()
{
    const size_t big = 1ull << 30;
    void *d = nullptr, *h = nullptr;
    if (hipMalloc(&d, big) != hipSuccess || hipHostMalloc(&h, big, hipHostMallocDefault) != hipSuccess) return 1;
    (void)hipMemset(d, 1, big);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    for (size_t chunk : {256ull << 20, 1ull << 30}) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (size_t off = 0; off < big; off += chunk)
                (void)hipMemcpyAsync((char *)h + off, (char *)d + off, chunk, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("D2H pinned, %4zu MiB chunks: %.1f GB/s\n", chunk >> 20, big / dt / 1e9);
        }
    }
    // a 10 GiB pinned destination filled from one 256 MiB device block (the shape of a cfg2 history download)
    void *hb = nullptr;
    const size_t nb = 10ull << 30, ck = 256ull << 20;
    if (hipHostMalloc(&hb, nb, hipHostMallocDefault) == hipSuccess) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (size_t off = 0; off < nb; off += ck)
                (void)hipMemcpyAsync((char *)hb + off, d, ck, hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("D2H into a 10 GiB pinned buffer, 256 MiB chunks: %.1f GB/s\n", nb / dt / 1e9);
        }
    }
    return 0;
}
