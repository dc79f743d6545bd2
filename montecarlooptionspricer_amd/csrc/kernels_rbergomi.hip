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

// workgroups per CU of the persistent FFT kernels (MCG_RB_GRID_PER_CU overrides, for timing studies)
static int rb_grid_per_cu() {
    static const int v = [] {
        const char* e = std::getenv("MCG_RB_GRID_PER_CU");
        const int n = e ? std::atoi(e) : 0;
        return n > 0 ? n : RB_WAVES;
    }();
    return v;
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
    const int64_t n_pairs = (P->n_paths + 1) / 2;
    const int ppb = rb_pairs_per_block(M);
    const int64_t n_blocks = (n_pairs + ppb - 1) / ppb;
    if (n_blocks > 0x7fffffffLL) return fail(MCG_ERR_INVALID, "n_paths too large for one launch");
    // FFT variants: persistent workgroups, RB_WAVES per CU (what their registers admit), each striding over the shares
    const int64_t grid = M < 32 ? n_blocks : std::min<int64_t>(n_blocks, (int64_t)ctx->n_cus * rb_grid_per_cu());

    rc = ensure_cap(ctx, &ctx->weights, &ctx->weights_cap, (size_t)M + (size_t)P->n_steps);
    if (rc) return rc;
    if (want_payoff) {
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
