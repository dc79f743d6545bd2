#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* av, const double* bv, double* raw) {
    const int l = threadIdx.x;
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[l], bv[l], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) raw[l * 4 + v] = c[v];
}
int main() {
    // operand registers: a[l], b[l] arbitrary distinct values; find which (la, lb) pairs contribute to each output
    std::vector<double> a(64), b(64), raw(256);
    double *da, *db, *dr; hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dr, 2048);
    // Use powers: a[l] = one-hot probing: run 64 experiments with a = e_la, b = all distinct primes-ish
    std::vector<int> arow(64), acol(64);
    // experiment 1: a one-hot at la, b[l] = l+1  -> outputs nonzero at (lane,v) with value = b[lb] for contributing lb
    printf("A lane -> (which D regs get it, paired with which B lanes)\n");
    for (int la = 0; la < 64; la += 1) {
        for (int l = 0; l < 64; ++l) { a[l] = (l == la) ? 1.0 : 0.0; b[l] = l + 1; }
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dr);
        hipMemcpy(raw.data(), dr, 2048, hipMemcpyDeviceToHost);
        if (la < 20 || la % 16 == 0) {
            printf("la=%2d:", la);
            int cnt = 0;
            for (int i = 0; i < 256 && cnt < 8; ++i) if (raw[i] != 0) { printf(" (lane %d,v%d)<-b%d", i / 4, i % 4, (int)raw[i] - 1); ++cnt; }
            int tot = 0; for (int i = 0; i < 256; ++i) tot += raw[i] != 0;
            printf("  ... total %d\n", tot);
        }
    }
    return 0;
}
