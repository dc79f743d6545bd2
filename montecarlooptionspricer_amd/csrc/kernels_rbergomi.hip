// rBergomi path generation for gfx950: kernels and launcher.  The algorithm (spectral synthesis by an
// in-register FFT spread over the lanes of a wavefront, log-space price stepping, 16-byte pair stores)
// is documented in rbergomi_device.hpp.
//
// Roofline: like the GBM kernel it is issue-bound on fp64 VALU work (per path and step: one volatility
// normal, one price normal, two exponentials, log2(Mz) half-butterflies) while writing 8 (n_steps+1)
// bytes per path; DESIGN.md section 5 has the instruction budget and the measurements.
#include <algorithm>
#include <cstdlib>

#include "mcg_internal.hpp"
#include "rbergomi_device.hpp"

namespace mcg {

// One share's payoff sums {sum, sum of squares} -> partials[2 share ..]: per SHARE, not per workgroup, because which
// workgroup draws which share differs from run to run and the fixed-order sum over the partials (finish_sums) must not.
template <bool PAYOFF>
__device__ __forceinline__ void rb_finish(const RbArgs& a, int64_t share, double end_a, double end_b, bool live_a,
                                          bool live_b, bool lead) {
    if (PAYOFF) {
        __shared__ double red[2 * 4];  // (re-used share after share: the barriers inside a share separate the uses)
        const bool call = a.is_call != 0;
        const double pay_a = (lead && live_a) ? payoff_of(call, end_a, a.K) : 0.0;
        const double pay_b = (lead && live_b) ? payoff_of(call, end_b, a.K) : 0.0;
        double v[2] = {pay_a + pay_b, pay_a * pay_a + pay_b * pay_b};
        block_sum<2, 4>(v, red);
        if (threadIdx.x == 0) {
            a.partials[2 * share] = v[0];
            a.partials[2 * share + 1] = v[1];
        }
    }
}

#ifndef RB_WAVES
#define RB_WAVES 2
#endif
// Two waves per SIMD: the kernel uses ~205 VGPRs (64 of them the transform); held to the 168 of three waves it spills
// 18-31 dwords and is slower, and at ~90 % VALU busy there is little left for a third wave to fill (-DRB_WAVES=3 to try).
template <int LG, int LT, bool PAYOFF>
__global__ __launch_bounds__(256, RB_WAVES) void k_rbergomi_fft(RbArgs a) {
    extern __shared__ double smem[];
    __shared__ fm::Tables tabs;
    // Persistent workgroups: the LDS tables (amplitudes, compensator, twiddles, the normal generator's tables) are
    // staged once, then the workgroup takes every gridDim.x-th share of 4 x 64/G pairs.
    const RbLds L = rb_stage_lds(a, smem, &tabs, RB_HALF_LOG2E);
    // Shares are handed out dynamically (the first gridDim.x by blockIdx, the rest from a ticket counter): workgroups
    // do not all run at the same speed, and a fixed stride leaves the fast ones idle at the end.  Thread 0 draws a
    // ticket one trip before it publishes it (so the atomic's latency is never waited for) into one of two alternating
    // LDS slots; the barriers inside a share order a slot's write against the reads of the trip before.
    __shared__ long long next_share[2];
    long long share = blockIdx.x;
    unsigned long long drawn = 0;  // thread 0: the ticket of the trip after the current one
    if (threadIdx.x == 0) drawn = atomicAdd(a.ticket, 1ull);
    for (int trip = 0; share < a.n_blocks; ++trip) {
        if (threadIdx.x == 0) {
            next_share[trip & 1] = (long long)(drawn + gridDim.x);
            drawn = atomicAdd(a.ticket, 1ull);
        }
        double la, lb;
        bool va, vb, lead;
        // (opaque per trip: hoisting the ~40 lane-derived indices and addresses out of the loop costs registers this
        // kernel does not have -- it spilled)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        // (likewise the two dozen wave-uniform conditions and strides derived from n_steps and ld: hoisted, they overflow
        // the scalar registers and come back through v_readlane on every trip; recomputed, they are a few s_cmp each)
        RbArgs b = a;
        asm volatile("" : "+s"(b.n_steps), "+s"(b.ld));
        rb_fft_block<LG, LT>(b, L, share, tid, &tabs, la, lb, va, vb, lead);
        rb_finish<PAYOFF>(a, share, la, lb, va, vb, lead);
        share = next_share[trip & 1];  // written before this trip's first barrier
    }
}

template <bool PAYOFF>
__global__ __launch_bounds__(256) void k_rbergomi_small(RbArgs a) {
    extern __shared__ double smem[];
    __shared__ fm::Tables tabs;
    double la, lb;
    bool va, vb, lead;
    rb_generate_small(a, (int64_t)blockIdx.x, smem, &tabs, la, lb, va, vb, lead);
    rb_finish<PAYOFF>(a, (int64_t)blockIdx.x, la, lb, va, vb, lead);
}

// ---- transforms longer than a wavefront holds (Mz > 2048, i.e. more than 2048 steps: beyond eight years of trading days) --
// The reference has no size limit (RoughVolatility.cpp:337-344), so neither has the entry point; these sizes take the
// definition literally instead of the in-register FFT: (1) the spectrum Y_k = a_k (g_k + i h_k) of a chunk of pairs goes
// to a workspace, (2) x_n = sum_k Y_k e^{2 pi i k n / Mz} is summed directly, one thread per (pair, step) -- O(Mz) per
// value, the arithmetic of the oracle's restatement --, (3) one thread per path steps the price through its column.
// Same Philox draws, same law, same matrix as the FFT variants would give; slow (O(Mz steps) per pair) and rare.
__global__ __launch_bounds__(256) void k_rb_generic_spectrum(RbArgs a, int64_t pair0, int64_t n_pairs, double2* Y) {
    __shared__ fm::Tables tabs;
    fm::load_tables(&tabs, a.log_tab);
    __syncthreads();
    const int64_t half = a.M / 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_pairs * half) return;
    const int64_t pair = idx / half, b = idx % half;
    const uint64_t pair_id = (a.path_begin >> 1) + (uint64_t)(pair0 + pair);
    double z[4];
    fm::normal_quad_fast(a.k0, a.k1, pair_id, (uint32_t)b, STREAM_VOL, &tabs, z, a.amp[2 * b], a.amp[2 * b + 1]);
    Y[pair * a.M + 2 * b] = make_double2(z[0], z[1]);
    Y[pair * a.M + 2 * b + 1] = make_double2(z[2], z[3]);
}

__global__ __launch_bounds__(256) void k_rb_generic_twiddle(double2* tw, int M) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= M) return;
    double sn, cs;
    sincospi(2.0 * (double)q / (double)M, &sn, &cs);
    tw[q] = make_double2(cs, sn);
}

__global__ __launch_bounds__(256) void k_rb_generic_x(RbArgs a, int64_t n_pairs, const double2* Y, const double2* tw, double2* X) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_pairs * a.n_steps) return;
    const int64_t pair = idx / a.n_steps;
    const uint64_t n = (uint64_t)(idx % a.n_steps);
    const double2* y = Y + pair * a.M;
    const uint64_t mask = (uint64_t)a.M - 1;
    double re = 0.0, im = 0.0;
    for (int k = 0; k < a.M; ++k) {
        const double2 w = tw[((uint64_t)k * n) & mask];
        const double2 v = y[k];
        re = fma(v.x, w.x, fma(-v.y, w.y, re));
        im = fma(v.x, w.y, fma(v.y, w.x, im));
    }
    X[idx] = make_double2(re, im);
}

__global__ __launch_bounds__(256) void k_rb_generic_step(RbArgs a, int64_t pair0, int64_t n_pairs, const double2* X) {
    __shared__ fm::Tables tabs;
    fm::load_tables(&tabs, a.log_tab);
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;  // path within the chunk
    const int64_t col = 2 * pair0 + p;
    if (p >= 2 * n_pairs || col >= a.n_paths) return;
    const uint64_t id = a.path_begin + (uint64_t)col;
    const double2* x = X + (p >> 1) * a.n_steps;
    const bool second = (p & 1) != 0;
    const double sq_xi_dt = sqrt(a.xi) * a.sqdt;
    double S = a.S0;
    a.out[col] = S;
    double z[4] = {0.0, 0.0, 0.0, 0.0};
    for (int n = 0; n < a.n_steps; ++n) {
        if ((n & 3) == 0) fm::normal_quad_fast(a.k0, a.k1, id, (uint32_t)(n >> 2), STREAM_PRICE, &tabs, z);
        const double2 xv = x[n];
        const double e = fm::exp_full(0.5 * ((second ? xv.y : xv.x) + a.comp[n]));  // sqrt(v / xi)
        const double inc = fma(sq_xi_dt * e, z[n & 3], fma(-0.5 * a.xi * a.dt, e * e, a.r * a.dt));  // RoughVolatility.cpp:359-363
        S *= fm::exp_full(inc);
        a.out[(int64_t)(n + 1) * a.ld + col] = S;
    }
}

static int launch_rbergomi_generic(mcg_ctx* ctx, const RbArgs& a) {
    const int64_t n_pairs = (a.n_paths + 1) / 2;
    const size_t per_pair = ((size_t)a.M + (size_t)a.n_steps) * sizeof(double2);
    int64_t chunk = (int64_t)std::max<size_t>(1, ((size_t)1 << 30) / per_pair);
    chunk = std::min(chunk, n_pairs);
    const size_t ws_bytes = (size_t)chunk * per_pair + (size_t)a.M * sizeof(double2);
    void* ws = nullptr;
    int rc = pool_alloc(ctx, ws_bytes, &ws);
    if (rc) return rc;
    double2* tw = (double2*)ws;
    double2* Y = tw + a.M;
    double2* X = Y + (size_t)chunk * a.M;
    hipLaunchKernelGGL(k_rb_generic_twiddle, dim3((unsigned)((a.M + 255) / 256)), dim3(256), 0, ctx->stream, tw, a.M);
    for (int64_t p0 = 0; p0 < n_pairs; p0 += chunk) {
        const int64_t np = std::min(chunk, n_pairs - p0);
        const int64_t g1 = (np * (a.M / 2) + 255) / 256, g2 = (np * a.n_steps + 255) / 256, g3 = (2 * np + 255) / 256;
        if (g1 > 0x7fffffffLL || g2 > 0x7fffffffLL) {
            pool_release(ctx, ws, ws_bytes);
            return fail(MCG_ERR_INVALID, "rBergomi: n_steps too large");
        }
        hipLaunchKernelGGL(k_rb_generic_spectrum, dim3((unsigned)g1), dim3(256), 0, ctx->stream, a, p0, np, Y);
        hipLaunchKernelGGL(k_rb_generic_x, dim3((unsigned)g2), dim3(256), 0, ctx->stream, a, np, (const double2*)Y, (const double2*)tw, X);
        hipLaunchKernelGGL(k_rb_generic_step, dim3((unsigned)g3), dim3(256), 0, ctx->stream, a, p0, np, (const double2*)X);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the workspace goes back to the pool
    pool_release(ctx, ws, ws_bytes);
    if (e != hipSuccess) return fail(MCG_ERR_HIP, "rBergomi (long transform) failed: %s", hipGetErrorString(e));
    return MCG_OK;
}

// gfx950 allows a workgroup up to 160 KB of LDS, but beyond 64 KB of dynamic LDS the kernel has to opt in
#define MCG_RB_LAUNCH(LG)                                                                                      \
    do {                                                                                                       \
        if (smem > 48 * 1024)                                                                                  \
            (void)hipFuncSetAttribute((const void*)k_rbergomi_fft<LG, LT, PAYOFF>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                  \
        hipLaunchKernelGGL((k_rbergomi_fft<LG, LT, PAYOFF>), g, b, smem, ctx->stream, a);                      \
    } while (0)

template <int LT, bool PAYOFF>
static void launch_fft(mcg_ctx* ctx, const RbArgs& a, dim3 g, dim3 b, size_t smem) {
    switch (a.M >> (2 + LT)) {  // lanes per pair
        case 1: MCG_RB_LAUNCH(0); break;
        case 2: MCG_RB_LAUNCH(1); break;
        case 4: MCG_RB_LAUNCH(2); break;
        case 8: MCG_RB_LAUNCH(3); break;
        case 16: MCG_RB_LAUNCH(4); break;
        case 32: MCG_RB_LAUNCH(5); break;
        default: MCG_RB_LAUNCH(6); break;
    }
}
#undef MCG_RB_LAUNCH

// workgroups per CU of the persistent FFT kernels
static int rb_grid_per_cu() {
    static const int v = study_switch("MCG_RB_GRID_PER_CU", 0);
    return v > 0 ? v : RB_WAVES;
}

template <bool PAYOFF>
static void launch_variant(mcg_ctx* ctx, const RbArgs& a, unsigned grid, size_t smem) {
    const dim3 g(grid), b(256);
    if (a.M < 32) hipLaunchKernelGGL(k_rbergomi_small<PAYOFF>, g, b, smem, ctx->stream, a);
    else if (rb_log_tiles(a.M) == 2) launch_fft<2, PAYOFF>(ctx, a, g, b, smem);
    else {  // Mz = 2048: 64 lanes x 32 points
        if (smem > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)k_rbergomi_fft<6, 3, PAYOFF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((k_rbergomi_fft<6, 3, PAYOFF>), g, b, smem, ctx->stream, a);
    }
}

int launch_rbergomi(mcg_ctx* ctx, mcg_paths* P, uint64_t seed, double S0, double r, double xi, double H, double eta,
                    double dt, bool want_payoff, double K, int is_call) {
    if (!(S0 > 0.0)) return fail(MCG_ERR_INVALID, "rBergomi needs S0 > 0 (log-space stepping)");
    if (P->path_begin & 1) return fail(MCG_ERR_INVALID, "rBergomi path_begin must be even (paths are generated in pairs)");
    std::vector<double> amp, comp;
    int rc = host_rbergomi_spectrum(H, eta, dt, P->n_steps, amp, comp);
    if (rc) return rc;
    const int M = (int)amp.size();
    const bool generic = M > 2048;  // longer than the in-register transforms reach: launch_rbergomi_generic
    const int64_t n_pairs = (P->n_paths + 1) / 2;
    const int ppb = generic ? 128 : rb_pairs_per_block(M);
    const int64_t n_blocks = (n_pairs + ppb - 1) / ppb;
    if (n_blocks > 0x7fffffffLL) return fail(MCG_ERR_INVALID, "n_paths too large for one launch");
    // FFT variants: persistent workgroups, RB_WAVES per CU (what their registers admit), each striding over the shares
    const int64_t grid = M < 32 ? n_blocks : std::min<int64_t>(n_blocks, (int64_t)ctx->n_cus * rb_grid_per_cu());

    rc = ensure_cap(ctx, &ctx->weights, &ctx->weights_cap, (size_t)M + (size_t)P->n_steps);
    if (rc) return rc;
    if (want_payoff && !generic) {
        rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, (size_t)(2 * n_blocks));
        if (rc) return rc;
    }
    MCG_HIP(hipMemcpyAsync(ctx->weights, amp.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
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
    a.amp = ctx->weights;
    a.comp = ctx->weights + M;
    a.log_tab = (const double2*)ctx->log_tab;
    a.K = K;
    a.is_call = is_call;
    a.partials = ctx->partials;
    a.n_blocks = n_blocks;
    a.ticket = reinterpret_cast<unsigned long long*>(ctx->scalars + SC_TICKET);
    if (generic) {
        {
            TimedLaunch t(ctx, MCG_K_RBERGOMI);
            rc = launch_rbergomi_generic(ctx, a);
        }
        if (rc) return rc;
        if (want_payoff) {  // (not fused on this route: one more read of the last row)
            rc = launch_payoff_sums(ctx, P, K, is_call, P->sums);
            if (rc) return rc;
            P->has_sums = true;
            P->sums_K = K;
            P->sums_is_call = is_call;
        }
        return MCG_OK;
    }
    if (M >= 32) MCG_HIP(hipMemsetAsync(a.ticket, 0, sizeof(unsigned long long), ctx->stream));
    const size_t smem = rb_smem_bytes(M, P->n_steps);
    {
        TimedLaunch t(ctx, MCG_K_RBERGOMI);
        if (want_payoff) launch_variant<true>(ctx, a, (unsigned)grid, smem);
        else launch_variant<false>(ctx, a, (unsigned)grid, smem);
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
