// Layout probe + issue-rate measurement for v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/, double* D /*16x16*/) {
    const int l = threadIdx.x;
    // hypothesis: A lane l holds A[l%16][l/16]; B lane l holds B[l/16][l%16]; D lane l reg v holds D[4*(l/16)+v][l%16]
    double a = A[(l % 16) * 4 + l / 16];
    double b = B[(l / 16) * 16 + l % 16];
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * (l / 16) + v) * 16 + l % 16] = c[v];
}

__global__ void k_rate(double* out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    v4d c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c7, 0, 0, 0);
    }
    v4d s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s[0] + s[1] + s[2] + s[3] == 1.2345) out[0] = 1;
}

// MFMA + independent VALU work in the same wave: do they overlap?
__global__ void k_rate_mixed(double* out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    v4d c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    double x0 = a, x1 = b, x2 = a + b, x3 = a - b;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        x0 = __builtin_fma(x0, 1.0000001, 1e-9); x1 = __builtin_fma(x1, 1.0000001, 1e-9);
        x2 = __builtin_fma(x2, 1.0000001, 1e-9); x3 = __builtin_fma(x3, 1.0000001, 1e-9);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        x0 = __builtin_fma(x0, 1.0000001, 1e-9); x1 = __builtin_fma(x1, 1.0000001, 1e-9);
        x2 = __builtin_fma(x2, 1.0000001, 1e-9); x3 = __builtin_fma(x3, 1.0000001, 1e-9);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        x0 = __builtin_fma(x0, 1.0000001, 1e-9); x1 = __builtin_fma(x1, 1.0000001, 1e-9);
        x2 = __builtin_fma(x2, 1.0000001, 1e-9); x3 = __builtin_fma(x3, 1.0000001, 1e-9);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        x0 = __builtin_fma(x0, 1.0000001, 1e-9); x1 = __builtin_fma(x1, 1.0000001, 1e-9);
        x2 = __builtin_fma(x2, 1.0000001, 1e-9); x3 = __builtin_fma(x3, 1.0000001, 1e-9);
    }
    v4d s = c0 + c1 + c2 + c3;
    if (s[0] + s[1] + s[2] + s[3] + x0 + x1 + x2 + x3 == 1.2345) out[0] = 1;
}

int main() {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.37; B[i] = 2 - i * 0.11; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("layout hypothesis max abs err: %.3e (%s)\n", err, err < 1e-9 ? "CONFIRMED" : "WRONG");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        const int iters = 20000;
        hipLaunchKernelGGL(k_rate, dim3(256), dim3(256 * wps), 0, 0, dD, 100); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_rate, dim3(256), dim3(256 * wps), 0, 0, dD, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mfma_per_simd = (double)wps * iters * 8;
        printf("mfma only, %d wave/SIMD: %.3f ms -> %.1f nominal cycles (2.4GHz) per MFMA per SIMD, %.1f TFLOP/s\n", wps, ms,
               ms * 1e-3 * 2.4e9 / mfma_per_simd, 1024.0 * mfma_per_simd * 2048 / (ms * 1e-3) / 1e12);
        hipEventRecord(e0); hipLaunchKernelGGL(k_rate_mixed, dim3(256), dim3(256 * wps), 0, 0, dD, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("mfma + 4 fp64 FMA each, %d wave/SIMD: %.3f ms -> %.1f nominal cycles per (MFMA + 4 FMA)\n", wps, ms,
               ms * 1e-3 * 2.4e9 / ((double)wps * iters * 4));
    }
    return 0;
}
