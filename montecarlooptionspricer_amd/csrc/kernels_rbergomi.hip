// rBergomi path generation for gfx950.
//
// Reference behaviour (per path, /root/reference/src/models/RoughVolatility.cpp:346-365):
//   Z -> X = sqrt(2H) eta Re(FFT^-(phi (.) Z)/Mz)   (:347-348, :264-292)
//   v_n = xi exp(X_n - 0.5 eta^2 t_n^{2H})          (:349, :294-309)
//   S_{n+1} = S_n exp((r - v_n/2) dt + sqrt(max(0,v_n)) dW_n),  dW_n ~ N(0, dt)  (:354-364)
//
// Device algorithm.  The law of X is reproduced by the real Volterra (circular-convolution) form
//   X_n = sum_{j<Mz} kappa_{(n-j) mod Mz} eps_j,  eps ~ iid N(0,1)       (host/volterra.cpp)
// which is a contraction over j -- the one place on this path where a matrix instruction fits:
//   X[n][path] = sum_j C[n][j] * eps[j][path],  C[n][j] = kappa_{(n-j) mod Mz}   (circulant).
// One wavefront owns 16 paths.  Per 4 values of j it issues one v_mfma_f64_16x16x4_f64 per 16-step
// tile: A = a 16x4 slice of C read from the LDS-staged, periodically extended weight vector (one
// ds_read_b64 per lane per MFMA, conflict-free), B = the 4x16 slice of eps that the lanes have just
// generated from Philox (each lane's Philox block feeds four consecutive MFMAs: no noise is ever stored), D = 16 steps x 16 paths of X
// accumulated in registers (16 tiles = 256 steps per pass; longer grids run in several passes and
// regenerate eps).  On gfx950 the fp64 MFMA runs at the fp64 VALU rate (64 cycles per instruction,
// measured in tools/ubench_mfma_f64.hip, and it does not overlap fp64 VALU work), so this kernel
// is bound by Mz*steps fp64 FMAs per path, not by HBM; what the MFMA buys is operand delivery:
// 1 LDS read per 1024 FMAs and no per-lane noise buffer.
//
// MFMA register layout on gfx950 (probed, tools/probe_mfma_layout.hip): A lane l = A[l%16][l/16],
// B lane l = B[l/16][l%16], D lane l reg v = D[4v + l/16][l%16].  Row i of A is given the time
// index n = 16t + 4(i%4) + i/4, so that D lane (g = l/16, c = l%16) reg v holds
// X[n = 16t + 4g + v][path c]: four CONSECUTIVE steps of one path per lane.
//
// Price stepping happens in that layout, in log space: each lane forms its four increments, a
// 4-element in-lane prefix plus a wavefront-shuffle scan over the four lane groups gives
// log S_n for all 16 steps of the tile, S_n = exp(.) is stored step-major (each store instruction
// writes four full 128-B lines).  Terminal payoff reduction as in the GBM kernel.
#include "mcg_internal.hpp"
#include "rbergomi_device.hpp"

namespace mcg {

template <bool PAYOFF>
__global__ __launch_bounds__(256) void k_rbergomi_paths(RbArgs a) {
    extern __shared__ double smem[];
    __shared__ fm::Tables tabs;
    bool lead;
    const double logS = rb_generate(a, (int64_t)blockIdx.x, smem, &tabs, lead);
    if (PAYOFF) {
        __shared__ double red[2 * 4];
        const double ST = fm::scaled_exp(1.0, logS);
        const double pay = lead ? payoff_of(a.is_call != 0, ST, a.K) : 0.0;
        double v[2] = {pay, pay * pay};
        block_sum<2, 4>(v, red);
        if (threadIdx.x == 0) {
            a.partials[2 * (int64_t)blockIdx.x] = v[0];
            a.partials[2 * (int64_t)blockIdx.x + 1] = v[1];
        }
    }
}

int launch_rbergomi(mcg_ctx* ctx, mcg_paths* P, uint64_t seed, double S0, double r, double xi, double H, double eta,
                    double dt, bool want_payoff, double K, int is_call) {
    if (!(S0 > 0.0)) return fail(MCG_ERR_INVALID, "rBergomi needs S0 > 0 (log-space stepping)");
    std::vector<double> kappa, comp;
    int rc = host_rbergomi_weights(H, eta, dt, P->n_steps, kappa, comp);
    if (rc) return rc;
    const int M = (int)kappa.size();
    const int64_t n_blocks = (P->n_paths + 63) / 64;
    if (n_blocks > 0x7fffffffLL) return fail(MCG_ERR_INVALID, "n_paths too large for one launch");

    rc = ensure_cap(ctx, &ctx->weights, &ctx->weights_cap, (size_t)M + (size_t)P->n_steps);
    if (rc) return rc;
    if (want_payoff) {
        rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, (size_t)(2 * n_blocks));
        if (rc) return rc;
    }
    MCG_HIP(hipMemcpyAsync(ctx->weights, kappa.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MCG_HIP(hipMemcpyAsync(ctx->weights + M, comp.data(), (size_t)P->n_steps * sizeof(double), hipMemcpyHostToDevice,
                           ctx->stream));
    // the host vectors die at return: make sure the copies have been consumed
    MCG_HIP(hipStreamSynchronize(ctx->stream));

    RbArgs a;
    a.out = P->data;
    a.ld = P->ld;
    a.n_paths = P->n_paths;
    a.n_steps = P->n_steps;
    a.M = M;
    a.path_begin = P->path_begin;
    a.k0 = (uint32_t)seed;
    a.k1 = (uint32_t)(seed >> 32);
    a.S0 = S0;
    a.logS0 = std::log(S0);
    a.r = r;
    a.xi = xi;
    a.dt = dt;
    a.sqdt = std::sqrt(dt);
    a.kappa = ctx->weights;
    a.comp = ctx->weights + M;
    a.log_tab = (const double2*)ctx->log_tab;
    a.K = K;
    a.is_call = is_call;
    a.partials = ctx->partials;
    const size_t smem = ((size_t)M + RB_PAD + (size_t)P->n_steps) * sizeof(double);
    {
        TimedLaunch t(ctx, MCG_K_RBERGOMI);
        if (want_payoff)
            hipLaunchKernelGGL(k_rbergomi_paths<true>, dim3((unsigned)n_blocks), dim3(256), smem, ctx->stream, a);
        else
            hipLaunchKernelGGL(k_rbergomi_paths<false>, dim3((unsigned)n_blocks), dim3(256), smem, ctx->stream, a);
    }
    MCG_HIP(hipGetLastError());
    if (want_payoff) {
        rc = finish_sums(ctx, n_blocks, P->n_paths, P->sums);
        if (rc) return rc;
        P->has_sums = true;
        P->sums_K = K;
        P->sums_is_call = is_call;
    }
    return MCG_OK;
}

}  // namespace mcg
