// MartingaleOptimization::PredictOptionPrice on a device-resident step-major path matrix (gfx950).
//
// Reference: /root/reference/src/models/MartingaleOptimizationPricer.cpp:21-189.  Per iteration the
// reference (a) finds each path's best discounted payoff and its date ("primal", :77-98), (b) takes
// max_j (discPayoff_j - (M(S_j) - offset)) floored at 0 ("dual", :100-121) with the polynomial M of the
// PREVIOUS iteration, (c) refits M by least squares on 2N samples -- (S_stop, 0.5 discPayoff_stop) and
// (S_other, 0.2 discPayoff_other), j_other = (j_stop + M/2) mod M (:126-176) -- and sets
// offset = mean M(S_0) (:178-183).  The primal scan does not depend on M, so the stop dates, hence the
// fitted M, are the same in every iteration: iteration 1 has dual == primal (M = 0), iterations >= 2 all
// have the dual of the fitted M.  The device therefore runs: one primal+moments stream, the small
// solve (shared with LSM: normal equations in x = S/K - 1, csrc/kernels_lsm.hip), one pass over row 0 for
// the offset, one dual stream.  Two HBM-read streams of the matrix (16 B per path per date); no MFMA.
// Sharded use: the moments (+ primal sum), the offset sum and the dual sum go through ctx->allreduce.
#include "devmath.hpp"
#include "lsm_device.hpp"
#include "mcg_internal.hpp"

namespace mcg {

struct MoArgs {
    const double* data;
    int64_t ld;
    int64_t n;
    int n_cols;          // M: all columns (j_other wraps over M, :143)
    int n_dates;         // columns with j*dt <= maturity (the scans break at the first t > maturity)
    const double* disc;  // [n_cols] exp(-r min(j dt, maturity))  (PathDiscountFactor, header :46-51)
    double K, invK;
    int is_call;
    const double* coef;  // device: the coefficient block of the refit (lsm_device.hpp: LSM_C_*): M in y = (S/K - 1) - centre
    double* partials;
    double center;       // k_mo_primal: the regressor's centre (0 on the first pass, the samples' mean when re-fitted)
};

// primal scan + regression moments.  partials[NM + 1][grid] (moment-major): NM moments, then the primal sum.
template <int NB>
__global__ __launch_bounds__(256) void k_mo_primal(MoArgs a) {
    constexpr int NM = 3 * NB - 1;
    __shared__ double red[(NM + 1) * 4];
    const bool call = a.is_call != 0;
    double m[NM + 1];
#pragma unroll
    for (int q = 0; q <= NM; ++q) m[q] = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < a.n; p += (int64_t)gridDim.x * 256) {
        const double* col = a.data + p;
        double best = 0.0;
        int stop = 0;
        for (int j = 0; j < a.n_dates; ++j) {  // :80-95
            const double d = payoff_of(call, col[(int64_t)j * a.ld], a.K) * a.disc[j];
            if (d > best) {
                best = d;
                stop = j;
            }
        }
        m[NM] += best;
        // the two regression samples of this path, :131-147
        const int other = (stop + a.n_cols / 2) % a.n_cols;
        const double xs[2] = {col[(int64_t)stop * a.ld], col[(int64_t)other * a.ld]};
        const double ys[2] = {0.5 * (payoff_of(call, xs[0], a.K) * a.disc[stop]),
                              0.2 * (payoff_of(call, xs[1], a.K) * a.disc[other])};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const double x = fma(xs[s], a.invK, -1.0) - a.center;
            double pw = 1.0;
#pragma unroll
            for (int q = 0; q < 2 * NB - 1; ++q) {
                m[q] += pw;
                if (q < NB) m[2 * NB - 1 + q] = fma(pw, ys[s], m[2 * NB - 1 + q]);
                pw *= x;
            }
        }
    }
    block_sum<NM + 1, 4>(m, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q <= NM; ++q) a.partials[(int64_t)q * gridDim.x + blockIdx.x] = m[q];
    }
}

template <int NB>
__device__ __forceinline__ double mo_poly(const double (&c)[NB], double center, double S, double invK) {
    return lsm_continuation<NB>(c, center, fma(S, invK, -1.0));
}

// sum_i M(S_i0) -> partials[block][2] (second column unused), :178-183
template <int NB>
__global__ __launch_bounds__(256) void k_mo_offset(MoArgs a) {
    __shared__ double red[2 * 4];
    double c[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) c[q] = a.coef[q];
    const double center = a.coef[LSM_C_CENTER];
    double v[2] = {0.0, 0.0};
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < a.n; p += (int64_t)gridDim.x * 256)
        v[0] += mo_poly<NB>(c, center, a.data[p], a.invK);
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) {
        a.partials[2 * (int64_t)blockIdx.x] = v[0];
        a.partials[2 * (int64_t)blockIdx.x + 1] = 0.0;
    }
}

// dual scan, :100-121
template <int NB>
__global__ __launch_bounds__(256) void k_mo_dual(MoArgs a, double offset) {
    __shared__ double red[2 * 4];
    const bool call = a.is_call != 0;
    double c[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) c[q] = a.coef[q];
    const double center = a.coef[LSM_C_CENTER];
    double v[2] = {0.0, 0.0};
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < a.n; p += (int64_t)gridDim.x * 256) {
        const double* col = a.data + p;
        double best = 0.0;
        for (int j = 0; j < a.n_dates; ++j) {
            const double S = col[(int64_t)j * a.ld];
            const double cand = payoff_of(call, S, a.K) * a.disc[j] - (mo_poly<NB>(c, center, S, a.invK) - offset);
            if (cand > best) best = cand;
        }
        v[0] += best;
    }
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) {
        a.partials[2 * (int64_t)blockIdx.x] = v[0];
        a.partials[2 * (int64_t)blockIdx.x + 1] = 0.0;
    }
}

template <int NB>
static int run_mo_nb(mcg_ctx* ctx, const MoArgs& a0, int grid, int64_t n_local, int max_iterations, double* price,
                     double* lower, double* upper) {
    constexpr int NM = 3 * NB - 1;
    MoArgs a = a0;
    double* moments = ctx->scalars + SC_LSM_MSG;  // NM moments + primal sum (<= 48 doubles)
    double* coef = ctx->scalars + SC_COEF;
    {
        TimedLaunch t(ctx, MCG_K_MARTINGALE);
        hipLaunchKernelGGL(k_mo_primal<NB>, dim3(grid), dim3(256), 0, ctx->stream, a);
    }
    // reduce (+ all-reduce) + solve; fewer than p+1 samples leave M = 0 (:150-153).  The refit is Eigen's bdcSvd on raw
    // monomials like LSM's (:166): when the first pass asks for it (lsm_solve_nb: ill-conditioned samples, or an order
    // whose raw monomials Eigen truncates) the samples are re-accumulated about their mean and solved by
    // lsm_solve_centered.  The request is read on the host -- this driver synchronises below anyway -- so sharded runs
    // refine too (every rank sees the same all-reduced moments, hence the same request).
    int rc = lsm_reduce_allreduce_solve(ctx, grid, NM + 1, NB, (double)NB, a.K, 1, 0.0);
    if (rc) return rc;
    MCG_HIP(hipMemcpyAsync(ctx->h_scalars + SC_COEF, coef, LSM_COEF_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->h_scalars[SC_COEF + LSM_C_REFINE] != 0.0) {
        // (the primal sum of the first pass, behind the moments, is overwritten by an identical one)
        a.center = ctx->h_scalars[SC_COEF + LSM_C_HINT];
        {
            TimedLaunch t(ctx, MCG_K_MARTINGALE);
            hipLaunchKernelGGL(k_mo_primal<NB>, dim3(grid), dim3(256), 0, ctx->stream, a);
        }
        rc = lsm_reduce_allreduce_solve(ctx, grid, NM + 1, NB, (double)NB, a.K, 2, a.center);
        if (rc) return rc;
    }
    double s[3];
    {
        TimedLaunch t(ctx, MCG_K_MARTINGALE);
        hipLaunchKernelGGL(k_mo_offset<NB>, dim3(grid), dim3(256), 0, ctx->stream, a);
    }
    rc = finish_sums(ctx, grid, n_local, s);  // {sum M(S0), 0, N}; all-reduced when sharded
    if (rc) return rc;
    const double n_total = s[2];
    if (!(n_total >= 1.0)) return fail(MCG_ERR_EMPTY_PATHS, "MartingaleOptimization: Empty pricePaths.");
    const double offset = s[0] / n_total;
    {
        TimedLaunch t(ctx, MCG_K_MARTINGALE);
        hipLaunchKernelGGL(k_mo_dual<NB>, dim3(grid), dim3(256), 0, ctx->stream, a, offset);
    }
    rc = finish_sums(ctx, grid, n_local, s);
    if (rc) return rc;
    const double dual_fitted = s[0] / n_total;
    // primal sum sits behind the moments (already all-reduced there)
    MCG_HIP(hipMemcpyAsync(ctx->h_scalars + SC_LSM_MSG, moments, (NM + 1) * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    const double primal = ctx->h_scalars[SC_LSM_MSG + NM] / n_total;
    (void)coef;
    const double dual = max_iterations >= 2 ? dual_fitted : primal;  // iteration 1 runs with M = 0
    if (lower) *lower = primal;
    if (upper) *upper = dual;
    *price = 0.5 * (primal + dual);  // :63
    return MCG_OK;
}

int run_martingale(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                   int poly_order, int max_iterations, double* price, double* lower, double* upper) {
    const int n_cols = P->n_steps + 1;
    std::vector<double> disc((size_t)n_cols);
    int n_dates = 0;
    for (int j = 0; j < n_cols; ++j) {
        double t = j * dt;
        if (!(t > maturity)) {
            if (n_dates == j) n_dates = j + 1;  // dates form a prefix: the scans break at the first t > maturity
        } else {
            t = maturity;
        }
        disc[(size_t)j] = std::exp(-r * t);
    }
    int grid = (int)std::min<int64_t>((P->n_paths + 255) / 256, (int64_t)ctx->n_cus * 8);
    if (grid < 1) grid = 1;
    int rc = ensure_cap(ctx, &ctx->weights, &ctx->weights_cap, (size_t)n_cols);
    if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, (size_t)grid * 48);
    if (rc) return rc;
    MCG_HIP(hipMemcpyAsync(ctx->weights, disc.data(), (size_t)n_cols * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MCG_HIP(hipStreamSynchronize(ctx->stream));

    MoArgs a;
    a.data = P->data;
    a.ld = P->ld;
    a.n = P->n_paths;
    a.n_cols = n_cols;
    a.n_dates = n_dates;
    a.disc = ctx->weights;
    a.K = K;
    a.invK = 1.0 / K;
    a.is_call = is_call;
    a.coef = ctx->scalars + SC_COEF;
    a.partials = ctx->partials;
    a.center = 0.0;
    typedef int (*Runner)(mcg_ctx*, const MoArgs&, int, int64_t, int, double*, double*, double*);
    static const Runner runners[LSM_MAX_NB] = {run_mo_nb<1>,  run_mo_nb<2>,  run_mo_nb<3>,  run_mo_nb<4>,  run_mo_nb<5>,  run_mo_nb<6>,
                                               run_mo_nb<7>,  run_mo_nb<8>,  run_mo_nb<9>,  run_mo_nb<10>, run_mo_nb<11>, run_mo_nb<12>,
                                               run_mo_nb<13>, run_mo_nb<14>, run_mo_nb<15>, run_mo_nb<16>};
    return runners[poly_order](ctx, a, grid, P->n_paths, max_iterations, price, lower, upper);
}

}  // namespace mcg
