// Micro-benchmark: HBM write ceiling for the path-matrix store pattern ([step][path], one path per
// lane, each wave writing one 512-B row segment per step) against plain streaming writes.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_write.hip -o tools/ubench_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PPL, bool NT, int WORK>
__global__ __launch_bounds__(256) void k_pattern(double* out, int64_t ld, int n_steps, double seed) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PPL;
    double* col = out + i;
    double s[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) s[p] = seed + (double)(i + p);
    for (int n = 0; n <= n_steps; ++n) {
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
#pragma unroll
            for (int w = 0; w < WORK; ++w) s[p] = __builtin_fma(s[p], 1.0000001, 1e-9);
        }
        if (PPL == 1) {
            if (NT) __builtin_nontemporal_store(s[0], col); else *col = s[0];
        } else {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d v = {s[0], s[PPL - 1]};
            if (NT) __builtin_nontemporal_store(v, (v2d*)col); else *(v2d*)col = v;
        }
        col += ld;
    }
}

__global__ __launch_bounds__(256) void k_stream(double2* out, int64_t n2, double seed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256)
        out[i] = make_double2(seed, seed + i);
}

template <typename F>
static float time_ms(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int64_t n_paths = 10'000'000, ld = (n_paths + 63) / 64 * 64; const int n_steps = 252;
    const size_t bytes = (size_t)ld * (n_steps + 1) * 8;
    double* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double gb = (double)n_paths * (n_steps + 1) * 8 / 1e9;
    auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, gb / (ms * 1e-3)); };
    const unsigned g1 = (unsigned)((n_paths + 255) / 256), g2 = (unsigned)((n_paths / 2 + 255) / 256);
    rep("stream double2, grid 2048", time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (double2*)d, (int64_t)(bytes / 16), 1.0); }));
    rep("pattern 8B/lane, plain store, work 0", time_ms([&] { hipLaunchKernelGGL((k_pattern<1, false, 0>), dim3(g1), dim3(256), 0, 0, d, ld, n_steps, 1.0); }));
    rep("pattern 8B/lane, nontemporal, work 0", time_ms([&] { hipLaunchKernelGGL((k_pattern<1, true, 0>), dim3(g1), dim3(256), 0, 0, d, ld, n_steps, 1.0); }));
    rep("pattern 16B/lane, plain store, work 0", time_ms([&] { hipLaunchKernelGGL((k_pattern<2, false, 0>), dim3(g2), dim3(256), 0, 0, d, ld, n_steps, 1.0); }));
    rep("pattern 16B/lane, nontemporal, work 0", time_ms([&] { hipLaunchKernelGGL((k_pattern<2, true, 0>), dim3(g2), dim3(256), 0, 0, d, ld, n_steps, 1.0); }));
    rep("pattern 8B/lane, nontemporal, work 40 fma", time_ms([&] { hipLaunchKernelGGL((k_pattern<1, true, 40>), dim3(g1), dim3(256), 0, 0, d, ld, n_steps, 1.0); }));
    rep("pattern 8B/lane, plain, work 40 fma", time_ms([&] { hipLaunchKernelGGL((k_pattern<1, false, 40>), dim3(g1), dim3(256), 0, 0, d, ld, n_steps, 1.0); }));
    rep("pattern 8B/lane, nontemporal, work 80 fma", time_ms([&] { hipLaunchKernelGGL((k_pattern<1, true, 80>), dim3(g1), dim3(256), 0, 0, d, ld, n_steps, 1.0); }));
    rep("pattern 16B/lane, nontemporal, work 40 fma", time_ms([&] { hipLaunchKernelGGL((k_pattern<2, true, 40>), dim3(g2), dim3(256), 0, 0, d, ld, n_steps, 1.0); }));
    return 0;
}
