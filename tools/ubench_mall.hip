// Micro-benchmark (dev tool, not part of the library): how fast does MI355X re-read a buffer that fits the 256 MB
// memory-side cache (MALL / Infinity Cache) compared with one that does not?  Decides whether the one-launch LSM
// sweep has to keep the current price row on chip between its regression pass and its update pass (8 B per path and
// date from HBM) or may simply read the row twice (the second read ~30 us after the first).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mall.hip -o tools/ubench_mall && tools/ubench_mall
// Patterns:
//   A  one buffer of B bytes read R times in a row                     (B = 16 MB .. 2 GB)
//   B  two 64 MB rows of a large matrix per pass, sliding by one row   (row j-1 fresh from HBM, row j read one pass ago)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, size_t n2, double* sink) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 v = a[i];
        acc += v.x + v.y;
    }
    if (acc == 123.456) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_read2(const double2* __restrict__ a, const double2* __restrict__ b, size_t n2,
                                               double* sink) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 v = a[i], w = b[i];
        acc += v.x + v.y + w.x * w.y;
    }
    if (acc == 123.456) sink[0] = acc;
}

int main() {
    const size_t total = (size_t)16 << 30;  // 16 GiB arena
    char* buf = nullptr;
    double* sink = nullptr;
    CK(hipMalloc((void**)&buf, total));
    CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    std::printf("pattern A: one buffer, read R times back to back (launch per read)\n");
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096}) {
        const size_t bytes = mb << 20, n2 = bytes / 16;
        const int R = 30;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const double2*)buf, n2, sink);
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const double2*)buf, n2, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("  %5zu MB  %8.1f GB/s  (%.1f us per read)\n", mb, (double)bytes * R / (ms * 1e-3) / 1e9, ms * 1e3 / R);
    }
    std::printf("pattern B: rows of 64 MB (8M doubles); pass j reads row j (read one pass ago) and row j-1 (fresh)\n");
    {
        const size_t row = (size_t)64 << 20, n2 = row / 16;
        const int rows = 200;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int j = rows - 1; j >= 1; --j)
                hipLaunchKernelGGL(k_read2, dim3(grid), dim3(256), 0, 0, (const double2*)(buf + (size_t)j * row),
                                   (const double2*)(buf + (size_t)(j - 1) * row), n2, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::printf("  %d passes: %.1f us per pass, %.1f GB/s counting both rows, %.1f GB/s counting the fresh row only\n",
                        rows - 1, ms * 1e3 / (rows - 1), 2.0 * row * (rows - 1) / (ms * 1e-3) / 1e9,
                        1.0 * row * (rows - 1) / (ms * 1e-3) / 1e9);
        }
        // the same with fresh rows only (no reuse): two different far-apart rows per pass
        CK(hipEventRecord(e0, 0));
        for (int j = rows - 1; j >= 1; j -= 2)
            hipLaunchKernelGGL(k_read2, dim3(grid), dim3(256), 0, 0, (const double2*)(buf + (size_t)j * row),
                               (const double2*)(buf + (size_t)(j - 1) * row), n2, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("  no reuse: %.1f us per pass, %.1f GB/s\n", ms * 1e3 / (rows / 2), 2.0 * row * (rows / 2) / (ms * 1e-3) / 1e9);
    }
    return 0;
}
