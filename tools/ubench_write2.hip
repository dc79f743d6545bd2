// Micro-benchmark (round 3): does the path-matrix store pattern gain from WIDER contiguous spans per row?  A workgroup
// of 256 threads holds 512 Q adjacent paths (Q two-path units per thread, 512 columns apart) and writes, per step, Q
// chunks of 4 KiB that together form 4 Q KiB of one row; rows are 80 MB apart.  WORK dependent FMAs per stored value
// stand for the generator's arithmetic.  Against: plain streaming writes of the same bytes.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_write2.hip -o tools/ubench_write2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef double v2d __attribute__((ext_vector_type(2)));

template <int Q, int WORK>
__global__ __launch_bounds__(256) void k_pattern(double* out, int64_t ld, int n_steps, double seed) {
    const int64_t base = (int64_t)blockIdx.x * (512 * Q) + 2 * threadIdx.x;
    double s[2 * Q];
#pragma unroll
    for (int p = 0; p < 2 * Q; ++p) s[p] = seed + (double)(base + p);
    double* row = out + base;
    for (int n = 0; n <= n_steps; ++n) {
#pragma unroll
        for (int p = 0; p < 2 * Q; ++p) {
#pragma unroll
            for (int w = 0; w < WORK; ++w) s[p] = __builtin_fma(s[p], 1.0000001, 1e-9);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            v2d v = {s[2 * q], s[2 * q + 1]};
            __builtin_nontemporal_store(v, (v2d*)(row + 512 * q));
        }
        row += ld;
    }
}

__global__ __launch_bounds__(256) void k_stream(double2* out, int64_t n2, double seed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256)
        __builtin_nontemporal_store(v2d{seed, seed + i}, (v2d*)(out + i));
}

template <typename F>
static float time_ms(F f, int reps = 8) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 6; ++r) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

template <int Q, int WORK>
static void run(double* d, int64_t n_paths, int64_t ld, int n_steps, double gb) {
    const unsigned g = (unsigned)(n_paths / (512 * Q));
    float ms = time_ms([&] { hipLaunchKernelGGL((k_pattern<Q, WORK>), dim3(g), dim3(256), 0, 0, d, ld, n_steps, 1.0); });
    printf("span %2d KiB per row and workgroup, %2d FMAs per value: %8.3f ms  %8.1f GB/s\n", 4 * Q, WORK, ms, gb / (ms * 1e-3));
}

int main() {
    const int64_t n_paths = 10'000'000 / 4096 * 4096, ld = n_paths; const int n_steps = 252;
    const size_t bytes = (size_t)ld * (n_steps + 1) * 8;
    double* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double gb = (double)n_paths * (n_steps + 1) * 8 / 1e9;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (double2*)d, (int64_t)(bytes / 16), 1.0); });
    printf("stream (nontemporal double2, grid 4096):              %8.3f ms  %8.1f GB/s\n", ms, gb / (ms * 1e-3));
    run<1, 0>(d, n_paths, ld, n_steps, gb);
    run<2, 0>(d, n_paths, ld, n_steps, gb);
    run<4, 0>(d, n_paths, ld, n_steps, gb);
    run<8, 0>(d, n_paths, ld, n_steps, gb);
    run<1, 40>(d, n_paths, ld, n_steps, gb);
    run<2, 40>(d, n_paths, ld, n_steps, gb);
    run<4, 40>(d, n_paths, ld, n_steps, gb);
    run<8, 40>(d, n_paths, ld, n_steps, gb);
    run<1, 25>(d, n_paths, ld, n_steps, gb);
    run<4, 25>(d, n_paths, ld, n_steps, gb);
    return 0;
}
