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
#include "devmath.hpp"
#include "fastmath.hpp"
#include "mcg_internal.hpp"

namespace mcg {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int RB_NT = 16;   // 16-step tiles accumulated per pass (256 steps)
constexpr int RB_PAD = 32;  // periodic extension of the weight vector in LDS

struct RbArgs {
    double* out;
    int64_t ld;
    int64_t n_paths;
    int n_steps;
    int M;  // Mz
    uint64_t path_begin;
    uint32_t k0, k1;
    double S0, logS0, r, xi, dt, sqdt;
    const double* kappa;  // [M]
    const double* comp;   // [n_steps]
    const double2* log_tab;
    double K;
    int is_call;
    double* partials;
};

// One pass over NT consecutive 16-step tiles starting at step n_base: accumulate X by MFMA, then
// advance the price through those steps.  Tiles (or single steps) beyond n_steps are computed but
// neither stored nor added to the running log-price.
template <int NT>
__device__ __forceinline__ void rb_pass(const RbArgs& a, const double* kext, const double* comp, const fm::Tables* tab,
                                        int n_base, int g, int c, int a_off, uint64_t id, bool live, double* col,
                                        double& logS) {
    const int M = a.M;
    v4d acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};

    const int n_kq = (M + 15) >> 4;
    for (int kq = 0; kq < n_kq; ++kq) {
        // this lane's four noise values: j = 16kq + 4g + e, e = 0..3 (one Philox block, number 4kq + g)
        double eps[4];
        fm::normal_quad_fast(a.k0, a.k1, id, (uint32_t)(4 * kq + g), STREAM_VOL, tab, eps);
        const int j0 = 16 * kq + 4 * g;
        const int base = (n_base - 16 * kq) & (M - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double ev = (j0 + e < M) ? eps[e] : 0.0;  // only matters when Mz < 16
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double w = kext[((base + 16 * t) & (M - 1)) + a_off - e];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, ev, acc[t], 0, 0, 0);
            }
        }
    }

    // price stepping; lane (g, c) owns steps nl .. nl+3 of path c in every tile
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nl = n_base + 16 * t + 4 * g;
        double z[4];  // steps nl..nl+3 are exactly Philox block nl/4 of the price stream
        fm::normal_quad_fast(a.k0, a.k1, id, (uint32_t)(nl >> 2), STREAM_PRICE, tab, z);
        double pre[4];
        double run = 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = nl + v;
            const bool valid = n < a.n_steps;
            const double var = fm::scaled_exp(a.xi, acc[t][v] + (valid ? comp[n] : 0.0));
            const double sd = fm::sqrt_pos(fmax(var, 1e-300)) * a.sqdt;
            const double inc = fma(sd, z[v], (a.r - 0.5 * var) * a.dt);
            run += valid ? inc : 0.0;
            pre[v] = run;
        }
        // inclusive scan of the group totals over g = 0..3 (lanes c, c+16, c+32, c+48)
        double incl = run;
        const double up16 = __shfl_up(incl, 16, 64);
        if (g >= 1) incl += up16;
        const double up32 = __shfl_up(incl, 32, 64);
        if (g >= 2) incl += up32;
        const double lead = logS + (incl - run);
        const double tile_total = __shfl(incl, 48 + c, 64);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = nl + v;
            const double S = fm::scaled_exp(1.0, lead + pre[v]);
            if (live && n < a.n_steps) __builtin_nontemporal_store(S, col + (int64_t)(n + 1) * a.ld);
        }
        logS += tile_total;
    }
}

template <bool PAYOFF>
__global__ __launch_bounds__(256) void k_rbergomi_paths(RbArgs a) {
    extern __shared__ double smem[];
    __shared__ fm::Tables tabs;
    const fm::Tables* tab = &tabs;
    const int M = a.M;
    double* kext = smem;                 // [M + RB_PAD], kext[i] = kappa[(i - 16) mod M]
    double* comp = smem + M + RB_PAD;    // [n_steps]
    for (int i = threadIdx.x; i < M + RB_PAD; i += 256) kext[i] = a.kappa[(i - 16 + 16 * M) & (M - 1)];
    for (int i = threadIdx.x; i < a.n_steps; i += 256) comp[i] = a.comp[i];
    fm::load_tables(&tabs, a.log_tab);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p = (int64_t)blockIdx.x * 64 + wave * 16 + c;
    const bool live = p < a.n_paths;
    const uint64_t id = a.path_begin + (uint64_t)p;
    // A operand: this lane supplies row i = c, i.e. local step 4(c%4) + c/4, for k-slot g
    // (noise index j = 16kq + 4g + e): weight index = step - j, shifted by the 16-entry extension
    const int a_off = 4 * (c & 3) + (c >> 2) - 4 * g + 16;

    double* col = a.out + p;
    if (live && g == 0) __builtin_nontemporal_store(a.S0, col);
    double logS = a.logS0;

    int n_base = 0;
    for (; n_base + RB_NT * 16 <= a.n_steps; n_base += RB_NT * 16)
        rb_pass<RB_NT>(a, kext, comp, tab, n_base, g, c, a_off, id, live, col, logS);
    const int tiles_left = (a.n_steps - n_base + 15) >> 4;  // 0..RB_NT-1, wave-uniform
    if (tiles_left > 8) rb_pass<16>(a, kext, comp, tab, n_base, g, c, a_off, id, live, col, logS);
    else if (tiles_left > 4) rb_pass<8>(a, kext, comp, tab, n_base, g, c, a_off, id, live, col, logS);
    else if (tiles_left > 2) rb_pass<4>(a, kext, comp, tab, n_base, g, c, a_off, id, live, col, logS);
    else if (tiles_left > 0) rb_pass<2>(a, kext, comp, tab, n_base, g, c, a_off, id, live, col, logS);

    if (PAYOFF) {
        __shared__ double red[2 * 4];
        const double ST = fm::scaled_exp(1.0, logS);
        const double pay = (live && g == 0) ? payoff_of(a.is_call != 0, ST, a.K) : 0.0;
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
