// Measurement aid (mcg_probe_write_ceiling): what THIS board writes, right now, with the path matrix's store pattern and
// no arithmetic at all -- the ceiling bench.py sets the GBM generator's achieved GB/s against, next to the 8 TB/s of the
// data sheet.  Boards of one pool differ by ~10 % on this power-limited kernel; without a ceiling
// measured in the same process a slow board and a regression look the same.
// The pattern is k_gbm_paths<.., PPL = 2>'s: a workgroup of 256 lanes owns 512 adjacent columns, every lane writes its
// two columns of a row with ONE nontemporal 16-byte store through a scalar row pointer, rows are `ld` doubles apart.
#include "mcg_internal.hpp"

namespace mcg {

__global__ __launch_bounds__(256) void k_probe_write(double* out, int64_t ld, int n_steps, double seed) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    double* row = out + (int64_t)blockIdx.x * 512;
    const unsigned lane_bytes = threadIdx.x * 16u;
    v2d v = {seed + (double)blockIdx.x, seed + (double)threadIdx.x};
    for (int n = 0; n <= n_steps; ++n) {
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(lane_bytes), "v"(v), "s"(row) : "memory");  // (the generator's statement, its hazard cover included)
        row += ld;
    }
}

int probe_write_ceiling(mcg_ctx* ctx, int64_t n_paths, int n_steps, int reps, double* gb_per_s, double* ms_per_launch) {
    const int64_t ld = (n_paths + 511) / 512 * 512;  // whole workgroups
    const size_t bytes = (size_t)ld * (size_t)(n_steps + 1) * sizeof(double);
    void* buf = nullptr;
    int rc = pool_alloc(ctx, bytes, &buf);
    if (rc) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    const dim3 grid((unsigned)(ld / 512)), block(256);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        hipLaunchKernelGGL(k_probe_write, grid, block, 0, ctx->stream, (double*)buf, ld, n_steps, 1.0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    for (int k = 0; k < reps && e == hipSuccess; ++k) {
        hipLaunchKernelGGL(k_probe_write, grid, block, 0, ctx->stream, (double*)buf, ld, n_steps, 2.0 + k);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) (void)hipStreamSynchronize(ctx->stream);
    pool_release(ctx, buf, bytes);
    if (e != hipSuccess) return fail(MCG_ERR_HIP, "write-ceiling probe failed: %s", hipGetErrorString(e));
    const double per = (double)ms / reps;
    if (ms_per_launch) *ms_per_launch = per;
    // counted like the generator's algorithmic bytes: 8 (n_steps + 1) per path of the REQUESTED count (the padding is written too)
    if (gb_per_s) *gb_per_s = 8.0 * (double)(n_steps + 1) * (double)n_paths / (per * 1e-3) / 1e9;
    return MCG_OK;
}

}  // namespace mcg
