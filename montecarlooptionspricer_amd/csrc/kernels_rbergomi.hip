// rBergomi path generation for gfx950.
//
// Reference behaviour (per path, /root/reference/src/models/RoughVolatility.cpp:346-365):
//   Z -> X = sqrt(2H) eta Re(FFT^-(phi (.) Z)/Mz)   (:347-348, :264-292)
//   v_n = xi exp(X_n - 0.5 eta^2 t_n^{2H})          (:349, :294-309)
//   S_{n+1} = S_n exp((r - v_n/2) dt + sqrt(max(0,v_n)) dW_n),  dW_n ~ N(0, dt)  (:354-364)
// Device algorithm: the law of X is reproduced by the real Volterra (circular-convolution) form
//   X_n = sum_{j<Mz} kappa_{(n-j) mod Mz} eps_j,  eps ~ iid N(0,1)       (host/volterra.cpp)
// with kappa staged in LDS (periodically extended so a tile of TN outputs reads a contiguous
// window) and the compensator table next to it.  One path per lane; the per-lane noise eps lives
// in a step-major scratch slab (coalesced 512-B wave accesses), X tiles of TN steps are
// accumulated in registers, consumed immediately by the price stepping and never stored.
// Stores of S are step-major like the GBM kernel.  Terminal payoff reduction as in GBM.
#include "devmath.hpp"
#include "mcg_internal.hpp"

namespace mcg {

constexpr int RB_TN = 8;  // outputs per register tile

struct RbArgs {
    double* out;
    int64_t ld;
    int64_t n_paths;
    int n_steps;
    int M;  // Mz
    uint64_t path_begin;
    uint32_t k0, k1;
    double S0, r, xi, dt, sqdt;
    const double* kappa;  // [M]
    const double* comp;   // [n_steps]
    double* eps;          // scratch [M][lds]
    int64_t lds;
    double K;
    int is_call;
    double* partials;
};

template <bool PAYOFF>
__global__ __launch_bounds__(256) void k_rbergomi_paths(RbArgs a) {
    extern __shared__ double smem[];
    const int M = a.M;
    double* kext = smem;                    // [M + 2*TN], kext[i] = kappa[(i - TN) mod M]
    double* comp = smem + M + 2 * RB_TN;    // [n_steps]
    for (int i = threadIdx.x; i < M + 2 * RB_TN; i += 256) kext[i] = a.kappa[(i - RB_TN + M) & (M - 1)];
    for (int i = threadIdx.x; i < a.n_steps; i += 256) comp[i] = a.comp[i];
    __syncthreads();

    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.n_paths;
    const uint64_t id = a.path_begin + (uint64_t)i;

    // volatility-driver noise for this lane
    double* eps = a.eps + i;
    for (int b = 0; 2 * b < M; ++b) {
        double z0, z1;
        normal_pair(a.k0, a.k1, id, (uint32_t)b, STREAM_VOL, z0, z1);
        eps[(int64_t)(2 * b) * a.lds] = z0;
        if (2 * b + 1 < M) eps[(int64_t)(2 * b + 1) * a.lds] = z1;
    }

    double* col = a.out + i;
    double S = a.S0;
    if (live) __builtin_nontemporal_store(S, col);
    double zc[2] = {0.0, 0.0};
    for (int n0 = 0; n0 < a.n_steps; n0 += RB_TN) {
        double acc[RB_TN];
#pragma unroll
        for (int t = 0; t < RB_TN; ++t) acc[t] = 0.0;
        for (int j = 0; j < M; ++j) {
            const double e = eps[(int64_t)j * a.lds];
            const double* w = kext + (((n0 - j) & (M - 1)) + RB_TN);
#pragma unroll
            for (int t = 0; t < RB_TN; ++t) acc[t] = fma(w[t], e, acc[t]);
        }
#pragma unroll
        for (int t = 0; t < RB_TN; ++t) {
            const int n = n0 + t;
            if (n < a.n_steps) {
                const double v = a.xi * exp(acc[t] + comp[n]);
                if ((n & 1) == 0) normal_pair(a.k0, a.k1, id, (uint32_t)(n >> 1), STREAM_PRICE, zc[0], zc[1]);
                const double drift = (a.r - 0.5 * v) * a.dt;
                const double sd = sqrt(fmax(0.0, v)) * a.sqdt;
                S = S * exp(fma(sd, zc[n & 1], drift));
                col += a.ld;
                if (live) __builtin_nontemporal_store(S, col);
            }
        }
    }
    if (PAYOFF) {
        __shared__ double red[2 * 4];
        const double pay = live ? payoff_of(a.is_call != 0, S, a.K) : 0.0;
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
    std::vector<double> kappa, comp;
    int rc = host_rbergomi_weights(H, eta, dt, P->n_steps, kappa, comp);
    if (rc) return rc;
    const int M = (int)kappa.size();
    const int64_t n_blocks = (P->n_paths + 255) / 256;
    if (n_blocks > 0x7fffffffLL) return fail(MCG_ERR_INVALID, "n_paths too large for one launch");
    const int64_t lds = n_blocks * 256;

    rc = ensure_cap(ctx, &ctx->weights, &ctx->weights_cap, (size_t)M + (size_t)P->n_steps);
    if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->scratch, &ctx->scratch_cap, (size_t)M * (size_t)lds);
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
    a.r = r;
    a.xi = xi;
    a.dt = dt;
    a.sqdt = std::sqrt(dt);
    a.kappa = ctx->weights;
    a.comp = ctx->weights + M;
    a.eps = ctx->scratch;
    a.lds = lds;
    a.K = K;
    a.is_call = is_call;
    a.partials = ctx->partials;
    const size_t smem = ((size_t)M + 2 * RB_TN + (size_t)P->n_steps) * sizeof(double);
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
