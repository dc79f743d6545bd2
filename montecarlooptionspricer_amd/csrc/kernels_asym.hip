// AsymptoticAnalysis::PredictOptionPrice on a device-resident step-major path matrix (gfx950).
//
// Reference: /root/reference/src/models/AsymptoticAnalysisPricer.cpp:38-113 -- per path, the best
// discounted payoff over the dates t_j = j*dt <= maturity at which S is beyond the analytic short-time
// exercise boundary (:8-36); the price is the mean over paths.  The boundary b_j and the discount
// e^{-r t_j} do not depend on the path: the host evaluates them once with the same libm calls as the
// reference and the kernel reads them through scalar loads.  One path per lane, the matrix is read
// row by row (64 consecutive doubles per wavefront per date): a pure HBM-read stream,
// 8 * (dates) bytes per path, no MFMA.  Per-path results are bit-identical to the reference (one
// subtract, one multiply, comparisons); only the final summation order differs.
#include <cmath>

#include "devmath.hpp"
#include "mcg_internal.hpp"

namespace mcg {

__global__ __launch_bounds__(256) void k_asym_scan(const double* data, int64_t ld, int64_t n_paths, int n_dates,
                                                   const double* bnd, const double* disc, double K, int is_call,
                                                   double* partials) {
    __shared__ double red[2 * 4];
    const bool call = is_call != 0;
    double v[2] = {0.0, 0.0};  // sum of per-path bests, number of valid paths
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_paths; p += (int64_t)gridDim.x * 256) {
        double best = 0.0;
        const double* col = data + p;
        for (int j = 0; j < n_dates; ++j) {
            const double S = col[(int64_t)j * ld];
            if (isnan(S) || isinf(S)) continue;                      // :74
            const double b = bnd[j];
            const bool in = call ? (S > b) : (S < b);                 // :80-85 (false when b is NaN)
            if (in) {
                const double pay = payoff_of(call, S, K);
                if (isnan(pay) || isinf(pay)) continue;               // :89
                const double d = disc[j] * pay;                       // :90
                if (d > best) best = d;
            }
        }
        if (!isnan(best) && !isinf(best)) {                           // :103-106
            v[0] += best;
            v[1] += 1.0;
        }
    }
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) {
        partials[2 * (int64_t)blockIdx.x] = v[0];
        partials[2 * (int64_t)blockIdx.x + 1] = v[1];
    }
}

int run_asymptotic(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                   double sigma, double dividend, double* price) {
    std::vector<double> bnd, disc;
    host_asymptotic_tables(P->n_steps + 1, r, K, maturity, dt, is_call, sigma, dividend, bnd, disc);
    const int n_dates = (int)bnd.size();
    int grid = (int)std::min<int64_t>((P->n_paths + 255) / 256, (int64_t)ctx->n_cus * 8);
    if (grid < 1) grid = 1;
    int rc = ensure_cap(ctx, &ctx->weights, &ctx->weights_cap, (size_t)2 * std::max(n_dates, 1));
    if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, (size_t)2 * grid);
    if (rc) return rc;
    if (n_dates > 0) {
        MCG_HIP(hipMemcpyAsync(ctx->weights, bnd.data(), (size_t)n_dates * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
        MCG_HIP(hipMemcpyAsync(ctx->weights + n_dates, disc.data(), (size_t)n_dates * sizeof(double),
                               hipMemcpyHostToDevice, ctx->stream));
        MCG_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die at return
    }
    {
        TimedLaunch t(ctx, MCG_K_ASYM);
        hipLaunchKernelGGL(k_asym_scan, dim3(grid), dim3(256), 0, ctx->stream, P->data, P->ld, P->n_paths, n_dates,
                           ctx->weights, ctx->weights + n_dates, K, is_call, ctx->partials);
    }
    MCG_HIP(hipGetLastError());
    double s[3];
    rc = finish_sums(ctx, grid, P->n_paths, s);  // {sum of bests, valid paths, n}; all-reduced when sharded
    if (rc) return rc;
    *price = s[1] > 0.0 ? s[0] / s[1] : 0.0;  // :108
    return MCG_OK;
}

}  // namespace mcg
