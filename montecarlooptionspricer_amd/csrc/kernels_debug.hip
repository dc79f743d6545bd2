// Test hook behind mcg_debug_eval: runs one routine of fastmath.hpp elementwise so that tests can
// measure its error against mpmath / libm.  Not on any product path.
#include "devmath.hpp"
#include "fastmath.hpp"
#include "mcg_internal.hpp"

namespace mcg {

__global__ __launch_bounds__(256) void k_debug_eval(int fn, const double* x, double* y, int64_t n,
                                                    const double2* gtab) {
    __shared__ fm::Tables tabs;
    const fm::Tables* tab = &tabs;
    fm::load_tables(&tabs, gtab);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double a = 0.0, b = 0.0;
    double z[4] = {0.0, 0.0, 0.0, 0.0};
    switch (fn) {
        case 0: a = fm::scaled_exp(1.0, v); break;
        case 1: a = fm::neg2log(v, tab->log); break;
        case 2: a = fm::sqrt_pos(v); break;
        case 3: fm::sincos_table((uint32_t)v, tab->sincos, a, b); break;
        case 4: fm::normal_quad_fast(1u, 0u, (uint64_t)v, 0u, STREAM_PRICE, tab, z); break;
        case 6: a = fm::scaled_exp_small(1.0, v); break;
        case 7: a = fm::scaled_exp_small6(1.0, v); break;
        case 8: fm::exp2_pair(v, v + 96.0, tab, a, b); break;  // 2^(v/256), 2^((v+96)/256)
        case 9: fm::normal_quad_fast<true>(1u, 0u, (uint64_t)v, 0u, STREAM_PRICE, tab, z); break;
        default: normal_quad_ref(1u, 0u, (uint64_t)v, 0u, STREAM_PRICE, z); break;
    }
    if (fn < 4 || (fn > 5 && fn != 9)) {
        z[0] = a;
        z[1] = b;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) y[4 * i + e] = z[e];
}

}  // namespace mcg

extern "C" int mcg_debug_eval(mcg_ctx* ctx, int fn, const double* x, double* y, int64_t n) {
    using namespace mcg;
    if (!ctx || !x || !y || n < 0 || fn < 0 || fn > 9) return fail(MCG_ERR_INVALID, "bad arguments");
    if (n == 0) return MCG_OK;
    MCG_HIP(hipSetDevice(ctx->device));
    double *dx = nullptr, *dy = nullptr;
    MCG_HIP(hipMalloc((void**)&dx, (size_t)n * sizeof(double)));
    if (hipMalloc((void**)&dy, (size_t)n * 4 * sizeof(double)) != hipSuccess) {
        (void)hipFree(dx);
        return fail(MCG_ERR_OOM, "hipMalloc failed");
    }
    int rc = MCG_OK;
    if (hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = fail(MCG_ERR_HIP, "H2D failed");
    if (rc == MCG_OK) {
        hipLaunchKernelGGL(k_debug_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, fn, dx, dy, n,
                           (const double2*)ctx->log_tab);
        if (hipMemcpyAsync(y, dy, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = fail(MCG_ERR_HIP, "debug eval failed: %s", hipGetErrorString(hipGetLastError()));
    }
    (void)hipFree(dx);
    (void)hipFree(dy);
    return rc;
}
