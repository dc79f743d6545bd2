// Does the gfx940+ hazard "VALU writes an SGPR -> a VALU instruction reads it within 2 wait states" bite the patterns this
// library has inside inline-asm statements (hipcc's hazard recogniser does not look into them)?
//   A: v_cmp (ballot) -> SGPR pair -> asm v_cndmask_b32 with that pair as its mask          (lsm_select, kernels_lsm.hip)
//   B: v_readfirstlane -> SGPR -> asm VALU reading it as a constant                         (a restored / rematerialised "s"(C))
//   C: asm global_store_dwordx4 -> the next VALU overwrites its data registers               (store_row, kernels_gbm.hip)
// Each pattern once as the compiler leaves it (the asm right behind the producer) and once with `s_nop 1` in between;
// wrong lanes are counted against a plain-C evaluation.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_hazard.hip -o tools/ubench_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <bool NOP>
__global__ __launch_bounds__(256) void k_a(const double* x, const int* a, const int* b, int* out, int n, int iters) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double xv = x[i];
    int av = a[i], bv = b[i], acc = 0;
    for (int k = 0; k < iters; ++k) {
        const double t = 0.5 + 0.001 * k;
        int r;
        if (NOP)
            asm volatile("v_cmp_lt_f64_e64 s[20:21], %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %4, %1, s[20:21]" : "=v"(r) : "v"(av), "v"(xv), "v"(t), "v"(bv) : "s20", "s21");
        else
            asm volatile("v_cmp_lt_f64_e64 s[20:21], %2, %3\n\tv_cndmask_b32_e64 %0, %4, %1, s[20:21]" : "=v"(r) : "v"(av), "v"(xv), "v"(t), "v"(bv) : "s20", "s21");
        acc += r ^ k;
        xv = xv * 1.0001;
        if (xv > 1.0) xv -= 0.7;
    }
    out[i] = acc;
}

template <bool NOP>
__global__ __launch_bounds__(256) void k_b(const int* a, int* out, int n, int iters) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int v = a[i], acc = 0;
    for (int k = 0; k < iters; ++k) {
        int r;
        v = v * 1664525 + 1013904223;
        if (NOP)
            asm volatile("v_readfirstlane_b32 s22, %1\n\ts_nop 1\n\tv_add_u32 %0, s22, %1" : "=v"(r) : "v"(v) : "s22");
        else
            asm volatile("v_readfirstlane_b32 s22, %1\n\tv_add_u32 %0, s22, %1" : "=v"(r) : "v"(v) : "s22");
        acc ^= r;
    }
    out[i] = acc;
}

template <int NOPS>
__global__ __launch_bounds__(256) void k_c(double* out, int64_t ld, int n_rows) {
    double* row = out + (int64_t)blockIdx.x * 512;
    const unsigned lane_bytes = threadIdx.x * 16u;
    double a = 1.0 + threadIdx.x, b = 2.0 + blockIdx.x;
    for (int j = 0; j < n_rows; ++j) {
        // {a, b} -> v[10:13], store them, then overwrite the data registers at once (zeros: what lands must still be a, b)
#define MCG_C_BODY(NOPSTR)                                                                                              \
    asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\ts_nop 4\n\t"         \
                 "global_store_dwordx4 %4, v[10:13], %5 nt\n\t" NOPSTR                                                  \
                 "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0"                          \
                 :                                                                                                      \
                 : "v"(__double2loint(a)), "v"(__double2hiint(a)), "v"(__double2loint(b)), "v"(__double2hiint(b)),      \
                   "v"(lane_bytes), "s"(row)                                                                            \
                 : "memory", "v10", "v11", "v12", "v13")
        if (NOPS == 0) MCG_C_BODY("");
        else if (NOPS == 1) MCG_C_BODY("s_nop 0\n\t");
        else MCG_C_BODY("s_nop 1\n\t");
#undef MCG_C_BODY
        a += 1.0;
        b += 1.0;
        row += ld;
    }
}

int main() {
    const int n = 1 << 20, iters = 200;
    std::vector<double> hx(n);
    std::vector<int> ha(n), hb(n), ho(n), hr(n);
    uint32_t s = 12345;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        hx[i] = (s >> 8) * (1.0 / 16777216.0);
        s = s * 1664525u + 1013904223u;
        ha[i] = (int)s;
        s = s * 1664525u + 1013904223u;
        hb[i] = (int)s;
    }
    double* dx; int *da, *db, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice);
    // A
    for (int i = 0; i < n; ++i) {
        double xv = hx[i]; int acc = 0;
        for (int k = 0; k < iters; ++k) { const double t = 0.5 + 0.001 * k; const int r = xv < t ? ha[i] : hb[i]; acc += r ^ k; xv = xv * 1.0001; if (xv > 1.0) xv -= 0.7; }
        hr[i] = acc;
    }
    for (int nop = 0; nop < 2; ++nop) {
        if (nop) hipLaunchKernelGGL(k_a<true>, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dout, n, iters);
        else hipLaunchKernelGGL(k_a<false>, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dout, n, iters);
        hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost);
        long bad = 0; for (int i = 0; i < n; ++i) bad += ho[i] != hr[i];
        printf("A v_cmp -> asm v_cndmask, %s: %ld of %d lanes wrong\n", nop ? "s_nop 1 between" : "back to back", bad, n);
    }
    // B: readfirstlane of v -> r = first-lane value + v: per wave of 64 the first ACTIVE lane
    for (int i = 0; i < n; ++i) hr[i] = 0;
    {
        std::vector<int> v(ha);
        for (int k = 0; k < iters; ++k) {
            for (int i = 0; i < n; ++i) v[i] = (int)((uint32_t)v[i] * 1664525u + 1013904223u);
            for (int w = 0; w < n; w += 64) for (int l = 0; l < 64; ++l) hr[w + l] ^= (int)((uint32_t)v[w] + (uint32_t)v[w + l]);
        }
    }
    for (int nop = 0; nop < 2; ++nop) {
        if (nop) hipLaunchKernelGGL(k_b<true>, dim3(n / 256), dim3(256), 0, 0, da, dout, n, iters);
        else hipLaunchKernelGGL(k_b<false>, dim3(n / 256), dim3(256), 0, 0, da, dout, n, iters);
        hipMemcpy(ho.data(), dout, n * 4, hipMemcpyDeviceToHost);
        long bad = 0; for (int i = 0; i < n; ++i) bad += ho[i] != hr[i];
        printf("B v_readfirstlane -> asm VALU reading the SGPR, %s: %ld of %d lanes wrong\n", nop ? "s_nop 1 between" : "back to back", bad, n);
    }
    // C
    const int64_t ld = 1 << 20; const int rows = 64; const int wgs = (int)(ld / 512);
    double* dm; hipMalloc(&dm, ld * rows * 8);
    std::vector<double> hm((size_t)ld * rows);
    for (int nops = 0; nops < 3; ++nops) {
        hipMemset(dm, 0, ld * rows * 8);
        if (nops == 0) hipLaunchKernelGGL(k_c<0>, dim3(wgs), dim3(256), 0, 0, dm, ld, rows);
        else if (nops == 1) hipLaunchKernelGGL(k_c<1>, dim3(wgs), dim3(256), 0, 0, dm, ld, rows);
        else hipLaunchKernelGGL(k_c<2>, dim3(wgs), dim3(256), 0, 0, dm, ld, rows);
        hipMemcpy(hm.data(), dm, ld * rows * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int j = 0; j < rows; ++j)
            for (int64_t c = 0; c < ld; c += 2) {
                const int wg = (int)(c / 512), t = (int)((c % 512) / 2);
                bad += hm[(size_t)j * ld + c] != 1.0 + t + j;
                bad += hm[(size_t)j * ld + c + 1] != 2.0 + wg + j;
            }
        printf("C asm global_store_dwordx4 -> VALU overwrites its data, %d wait state(s) between: %ld of %ld values wrong\n", nops, bad, (long)ld * rows);
    }
    return 0;
}
