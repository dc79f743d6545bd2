// Longstaff-Schwartz backward sweep on a device-resident step-major path matrix (gfx950).
//
// Follows /root/reference/src/models/LSMPricer.cpp:19-102 (the value-iteration variant: the FITTED
// continuation value is carried backwards, :85) with three MI355X-first changes that leave the
// fitted values -- and therefore every V[i][j] -- unchanged up to rounding:
//   * only one value vector V (n_paths doubles) is live instead of Values[N][M] (:35);
//   * per exercise date ONE streaming kernel does "update V with date j's fit" and "accumulate the
//     regression moments of date j-1" (reads S_j, S_{j-1}, V; writes V: 32 B/path/date);
//   * the least-squares fit (:61-76, Eigen bdcSvd on raw monomials) is solved from the
//     (p+1)x(p+1) moment matrix of the SCALED regressor x = S/K - 1 (same polynomial space => same
//     fitted values; cond drops from ~1e8 to ~1e2), via a Jacobi eigen-decomposition pseudo-inverse
//     so rank-deficient dates (one ITM path, all paths equal at j=0) still give the projection.
//     The moments are the only cross-GPU exchange: 3p+2 doubles per date through ctx->allreduce.
// HBM-bound streaming; no MFMA (the "GEMM" A^T A is a (p+1)^2 moment accumulation, done in
// registers with wavefront-shuffle reductions).
#include "lsm_device.hpp"
#include "mcg_internal.hpp"

namespace mcg {

enum { UPD_INIT = 0, UPD_REGRESS = 1, UPD_DISCOUNT = 2 };

struct LsmArgs {
    const double* S_upd;  // row being updated (date j), or the last row for UPD_INIT
    const double* S_mom;  // row j-1 whose moments are accumulated (nullptr: none)
    double* V;
    int64_t n;
    double K, invK, disc;
    int is_call;
    int upd;
    const double* coef;   // device: coef[0..NB), coef[9] = number of ITM paths of date j (global)
    double* partials;     // [NM][gridDim.x] (moment-major: the reduce kernel reads contiguously)
    int rev;              // walk the grid-stride chunks from the last to the first (see run_lsm)
};

// NB = poly_order + 1 basis functions; NM = (2p+1) power sums + (p+1) cross sums = 3*NB - 1.
template <int NB>
__global__ __launch_bounds__(256) void k_lsm_sweep(LsmArgs a) {
    constexpr int NM = 3 * NB - 1;
    __shared__ double red[NM * 4];
    const bool call = a.is_call != 0;
    double c[NB];
    double n_itm = 0.0;
    if (a.upd == UPD_REGRESS) {
#pragma unroll
        for (int q = 0; q < NB; ++q) c[q] = a.coef[q];
        n_itm = a.coef[9];
    }
    double m[NM];
#pragma unroll
    for (int q = 0; q < NM; ++q) m[q] = 0.0;

    const int64_t chunk = (int64_t)gridDim.x * 256;
    const int64_t n_chunks = (a.n + chunk - 1) / chunk;
    for (int64_t k = 0; k < n_chunks; ++k) {
        const int64_t i = (a.rev ? n_chunks - 1 - k : k) * chunk + (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= a.n) continue;
        double v;
        if (a.upd == UPD_INIT) {
            v = payoff_of(call, a.S_upd[i], a.K);  // LSMPricer.cpp:37-40
        } else if (a.upd == UPD_DISCOUNT) {
            v = a.V[i] * a.disc;  // :43-49
        } else {
            const double s = a.S_upd[i];
            const double pay = payoff_of(call, s, a.K);
            const double vn = a.V[i] * a.disc;
            if (pay > 1e-14 && n_itm > 0.0) {  // :78-86
                const double x = fma(s, a.invK, -1.0);
                double cont = c[NB - 1];
#pragma unroll
                for (int q = NB - 2; q >= 0; --q) cont = fma(cont, x, c[q]);
                v = fmax(pay, cont);
            } else if (pay < 1e-14) {  // :89-94
                v = vn;
            } else {
                v = 0.0;  // payoff == 1e-14 exactly falls through both branches (:55 vs :91)
            }
        }
        a.V[i] = v;
        if (a.S_mom) {  // regression inputs of the next (earlier) date, :51-74
            const double s = a.S_mom[i];
            if (payoff_of(call, s, a.K) > 1e-14) {
                const double x = fma(s, a.invK, -1.0);
                const double y = v * a.disc;
                double pw = 1.0;
#pragma unroll
                for (int q = 0; q < 2 * NB - 1; ++q) {
                    m[q] += pw;
                    if (q < NB) m[2 * NB - 1 + q] = fma(pw, y, m[2 * NB - 1 + q]);
                    pw *= x;
                }
            }
        }
    }
    if (a.S_mom) {
        block_sum<NM, 4>(m, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < NM; ++q) a.partials[(int64_t)q * gridDim.x + blockIdx.x] = m[q];
        }
    }
}

// One block.  do_reduce: partials[nm][n_blocks] -> moments[nm] in a fixed order (wave w sums moments
// w, w+4, ...: lanes stride over the blocks, then a wavefront butterfly).  do_solve: moments -> coef.
// Single GPU: both in one launch.  Sharded: reduce, all-reduce of `moments`, then solve.
__global__ __launch_bounds__(256) void k_lsm_reduce_solve(const double* partials, int n_blocks, int nm, int nb,
                                                          double* moments, double* coef, int do_reduce, int do_solve,
                                                          double min_count) {
    __shared__ double sm[32];
    if (do_reduce) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int q = wave; q < nm; q += 4) {
            double s = 0.0;
#pragma unroll 8
            for (int b = lane; b < n_blocks; b += 64) s += partials[(int64_t)q * n_blocks + b];  // loads issue ahead of the adds
            s = wave_sum(s);
            if (lane == 0) {
                moments[q] = s;  // for the all-reduce / the host
                sm[q] = s;       // for the solve below (same block: hand over through LDS)
            }
        }
        __syncthreads();
    }
    if (do_solve && threadIdx.x == 0) lsm_solve_one(do_reduce ? sm : moments, nb, min_count, coef);
}

template <int NB, int PPT>
__global__ __launch_bounds__(256) void k_lsm_small(const double* data, int64_t ld, int n, int n_cols, double r_unused,
                                                   double K, double maturity, double dt, double disc, int is_call,
                                                   double* out3) {
    (void)r_unused;
    lsm_small_body<NB, PPT>(data, ld, n, n_cols, K, maturity, dt, disc, is_call, out3);
}

template <int NB>
static void launch_small_nb(mcg_ctx* ctx, const mcg_paths* P, double K, double maturity, double dt, double disc,
                            int is_call, double* out3) {
    if (P->n_paths <= 256)  // the production shape (250 paths): one path per thread
        hipLaunchKernelGGL((k_lsm_small<NB, 1>), dim3(1), dim3(256), 0, ctx->stream, P->data, P->ld, (int)P->n_paths,
                           P->n_steps + 1, 0.0, K, maturity, dt, disc, is_call, out3);
    else
        hipLaunchKernelGGL((k_lsm_small<NB, 4>), dim3(1), dim3(256), 0, ctx->stream, P->data, P->ld, (int)P->n_paths,
                           P->n_steps + 1, 0.0, K, maturity, dt, disc, is_call, out3);
}

// sum V, sum V^2 -> partials[grid][2]
__global__ __launch_bounds__(256) void k_lsm_final(const double* V, int64_t n, double* partials) {
    __shared__ double red[2 * 4];
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = V[i];
        v[0] += x;
        v[1] += x * x;
    }
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) {
        partials[2 * (int64_t)blockIdx.x] = v[0];
        partials[2 * (int64_t)blockIdx.x + 1] = v[1];
    }
}

// partials[grid][nm] -> moments (fixed order) -> optional all-reduce -> coefficients in ctx->scalars.
// Shared by the LSM sweep and the MartingaleOptimization refit.
int lsm_reduce_allreduce_solve(mcg_ctx* ctx, int grid, int nm, int nb, double min_count) {
    double* moments = ctx->scalars + SC_MOMENTS;
    double* coef = ctx->scalars + SC_COEF;
    if (ctx->allreduce) {
        {
            TimedLaunch t(ctx, MCG_K_LSM_SOLVE);
            hipLaunchKernelGGL(k_lsm_reduce_solve, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, grid, nm, nb,
                               moments, coef, 1, 0, min_count);
        }
        if (ctx->allreduce(ctx->allreduce_user, moments, nm, (void*)ctx->stream) != 0)
            return fail(MCG_ERR_COMM, "all-reduce of regression moments failed");
        TimedLaunch t(ctx, MCG_K_LSM_SOLVE);
        hipLaunchKernelGGL(k_lsm_reduce_solve, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, grid, nm, nb, moments,
                           coef, 0, 1, min_count);
    } else {
        TimedLaunch t(ctx, MCG_K_LSM_SOLVE);
        hipLaunchKernelGGL(k_lsm_reduce_solve, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, grid, nm, nb, moments,
                           coef, 1, 1, min_count);
    }
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}

template <int NB>
static void launch_sweep_nb(mcg_ctx* ctx, int grid, const LsmArgs& a) {
    hipLaunchKernelGGL(k_lsm_sweep<NB>, dim3(grid), dim3(256), 0, ctx->stream, a);
}

static void launch_sweep(mcg_ctx* ctx, int nb, int grid, const LsmArgs& a) {
    TimedLaunch t(ctx, MCG_K_LSM_SWEEP);
    switch (nb) {
        case 1: launch_sweep_nb<1>(ctx, grid, a); break;
        case 2: launch_sweep_nb<2>(ctx, grid, a); break;
        case 3: launch_sweep_nb<3>(ctx, grid, a); break;
        case 4: launch_sweep_nb<4>(ctx, grid, a); break;
        case 5: launch_sweep_nb<5>(ctx, grid, a); break;
        case 6: launch_sweep_nb<6>(ctx, grid, a); break;
        case 7: launch_sweep_nb<7>(ctx, grid, a); break;
        case 8: launch_sweep_nb<8>(ctx, grid, a); break;
        default: launch_sweep_nb<9>(ctx, grid, a); break;
    }
}

int run_lsm(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
            int poly_order, double* mean, double* std_err) {
    const int nb = poly_order + 1;
    const int nm = 3 * nb - 1;
    const int64_t N = P->n_paths;
    const int M = P->n_steps + 1;
    int grid = (int)std::min<int64_t>((N + 255) / 256, (int64_t)ctx->n_cus * 8);
    if (grid < 1) grid = 1;

    if (N >= 1 && N <= 1024 && !ctx->allreduce) {  // one launch for the whole sweep
        const double disc_s = std::exp(-r * dt);
        double* d3 = ctx->scalars + SC_SUMS;
        {
            TimedLaunch t(ctx, MCG_K_LSM_SWEEP);
            switch (nb) {
                case 1: launch_small_nb<1>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
                case 2: launch_small_nb<2>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
                case 3: launch_small_nb<3>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
                case 4: launch_small_nb<4>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
                case 5: launch_small_nb<5>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
                case 6: launch_small_nb<6>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
                case 7: launch_small_nb<7>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
                case 8: launch_small_nb<8>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
                default: launch_small_nb<9>(ctx, P, K, maturity, dt, disc_s, is_call, d3); break;
            }
        }
        MCG_HIP(hipGetLastError());
        MCG_HIP(hipMemcpyAsync(ctx->h_scalars + SC_SUMS, d3, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MCG_HIP(hipStreamSynchronize(ctx->stream));
        const double n = ctx->h_scalars[SC_SUMS + 2], m = ctx->h_scalars[SC_SUMS] / n;
        *mean = m;
        if (std_err) {
            const double var = n > 1.0 ? std::max(0.0, (ctx->h_scalars[SC_SUMS + 1] - n * m * m) / (n - 1.0)) : 0.0;
            *std_err = std::sqrt(var / n);
        }
        return MCG_OK;
    }
    int rc = ensure_cap(ctx, &ctx->lsm_v, &ctx->lsm_v_cap, (size_t)std::max<int64_t>(N, 1));
    if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, (size_t)grid * (size_t)std::max(nm, 2));
    if (rc) return rc;

    double* coef = ctx->scalars + SC_COEF;
    const double disc = std::exp(-r * dt);  // LSMPricer.cpp:46,:69,:92
    auto row = [&](int j) { return P->data + (int64_t)j * P->ld; };
    auto regress_at = [&](int j) {  // :43-44
        const double this_time = j * dt;
        return !(this_time > maturity);
    };

    LsmArgs a;
    a.V = ctx->lsm_v;
    a.n = N;
    a.K = K;
    a.invK = 1.0 / K;
    a.disc = disc;
    a.is_call = is_call;
    a.coef = coef;
    a.partials = ctx->partials;
    // Consecutive sweeps walk the paths in opposite directions: what sweep j touched last (the tail of V and of
    // row j-1, which sweep j-1 reads again) is what sweep j-1 touches first, while it is still in the 256 MB
    // memory-side cache.
    a.rev = 0;

    // terminal payoff, fused with the moments of date M-2
    a.upd = UPD_INIT;
    a.S_upd = row(M - 1);
    a.S_mom = (M >= 2 && regress_at(M - 2)) ? row(M - 2) : nullptr;
    launch_sweep(ctx, nb, grid, a);
    MCG_HIP(hipGetLastError());

    for (int j = M - 2; j >= 0; --j) {
        const bool reg = regress_at(j);
        if (reg) {
            int rcs = lsm_reduce_allreduce_solve(ctx, grid, nm, nb, 1.0);
            if (rcs) return rcs;
        }
        a.upd = reg ? UPD_REGRESS : UPD_DISCOUNT;
        a.S_upd = row(j);
        a.S_mom = (j >= 1 && regress_at(j - 1)) ? row(j - 1) : nullptr;
        a.rev ^= 1;
        launch_sweep(ctx, nb, grid, a);
    }
    MCG_HIP(hipGetLastError());

    {
        TimedLaunch t(ctx, MCG_K_LSM_SWEEP);
        hipLaunchKernelGGL(k_lsm_final, dim3(grid), dim3(256), 0, ctx->stream, ctx->lsm_v, N, ctx->partials);
    }
    MCG_HIP(hipGetLastError());
    double s[3];
    rc = finish_sums(ctx, grid, N, s);
    if (rc) return rc;
    const double n = s[2];
    if (!(n >= 1.0)) return fail(MCG_ERR_EMPTY_PATHS, "LSM::PredictOptionPrice: Empty pricePaths.");
    const double m = s[0] / n;  // :97-101
    *mean = m;
    if (std_err) {
        const double var = n > 1.0 ? std::max(0.0, (s[1] - n * m * m) / (n - 1.0)) : 0.0;
        *std_err = std::sqrt(var / n);
    }
    return MCG_OK;
}

}  // namespace mcg
