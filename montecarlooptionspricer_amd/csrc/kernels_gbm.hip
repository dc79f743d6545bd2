// GBM path generation for gfx950: one path per lane, Philox normals (four per block), S in a register,
// step-major stores so a wavefront writes 64 consecutive doubles (512 B, four full 128-B lines)
// per time step, and a wavefront-shuffle reduction of the terminal payoff.
//
// Replaces, on the device, the stepping loop of /root/reference/src/models/RoughVolatility.cpp:354-364
// with v == sigma^2 (GBM is contained in the reference as that special case) and
// PayoffFunction (include/core/common.h:8-14) on the last column.
//
// Roofline: HBM write, 8*(n_steps+1) bytes per path, reads ~0.  No MFMA: the step is elementwise.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "devmath.hpp"
#include "fastmath.hpp"
#include "mcg_internal.hpp"

#ifndef MCG_GBM_TABLES
#define MCG_GBM_TABLES 0
#endif

namespace mcg {

struct GbmArgs {
    double* out;       // [n_steps+1][ld]
    int64_t ld;
    int64_t n_paths;
    int n_steps;
    uint64_t path_begin;
    uint32_t k0, k1;   // Philox key = seed
    double S0, drift, vol;
    fm::LogScale ls;   // MODE >= 2: the logarithm's constants times vol^2 (fm::neg2log_scaled)
    double K;
    int is_call;
    double* partials;  // [gridDim.x][2]
    const double2* log_tab;  // fm::LOG_TAB_HOST on the device
    // mcg_generator_clock: the workgroups whose index is clk_first modulo 2^clk_shift stamp their life with the shader-cycle
    // and the 100 MHz counters into slot (index >> clk_shift) -- two scalar reads at each end of <= 64 workgroups per
    // launch; the start stamps wait in LDS, not in registers (the kernel has no scalar register to spare: held in four
    // of them for the kernel's life, the stamps pushed Horner constants out into spilled lanes)
    unsigned long long* clk;
    unsigned clk_first, clk_shift;
};

// MODE 0: any parameters.  MODE 1 (SMALL): the host has checked |drift| + vol * 7.55 <= 0.125, so every step's exponent
// fits fm::scaled_exp_small (no range reduction, two fewer polynomial terms).  MODE 2: SMALL and vol > 0 folded into
// the logarithm (fm::neg2log_scaled): one multiply less per pair.  MODE 3: MODE 2 with the bound at 0.1, where e^a needs
// one polynomial term less (fm::scaled_exp_small6).
// PPL: paths per lane.  1: a wavefront writes 512 B per step.  2: a lane carries two adjacent paths and writes them with
// one 16-byte store (1 KB per wavefront and step; half the store instructions, half the loop and addressing overhead,
// two independent chains per lane) -- tools/ubench_write.hip: with 40 FMAs of work per path-step the 16-byte pattern
// runs 5.24 TB/s against 5.02.  A workgroup then covers 512 columns; rows are padded to 256, so the upper two waves of
// the last workgroup may lie beyond the row and sit the generation out (wave-uniform).
template <bool PAYOFF, int MODE, int PPL>
__global__ __launch_bounds__(256) void k_gbm_paths(GbmArgs a) {
    constexpr bool SMALL = MODE >= 1;
    typedef double v2d __attribute__((ext_vector_type(2)));
    // MCG_GBM_TABLES (A/B builds): 0 = all three tables of the rBergomi kernels (34 KiB: 4 workgroups per CU), 1 = logarithm
    // + 1024-entry sin/cos (32 KiB: 5), 2 = logarithm + 512-entry sin/cos (24 KiB: 6; one more cosine term per pair)
#if MCG_GBM_TABLES == 0
    typedef fm::Tables GbmTables;
    __shared__ GbmTables tabs;
    if (MODE >= 2) fm::load_tables_scaled(&tabs, a.log_tab, a.vol * a.vol);
    else fm::load_tables(&tabs, a.log_tab);
#else
    typedef fm::NormalTables<MCG_GBM_TABLES == 1 ? 10 : 9> GbmTables;
    __shared__ GbmTables tabs;
    fm::load_normal_tables(&tabs, a.log_tab, MODE >= 2 ? a.vol * a.vol : 1.0);
#endif
    const GbmTables* tab = &tabs;
    __syncthreads();
#ifdef MCG_GBM_NO_STAMPS  // (A/B builds: the kernel without its clock stamps)
    const bool stamps = false;
#else
    const bool stamps = a.clk != nullptr && (blockIdx.x & ((1u << a.clk_shift) - 1u)) == a.clk_first;  // (wave-uniform)
#endif
    // (everything the end of the kernel needs -- whether this workgroup stamps, where to, the start stamps -- waits in LDS)
    __shared__ unsigned long long stamp0[3];
    if (threadIdx.x == 0) {
        stamp0[2] = 0;
        if (stamps && (blockIdx.x >> a.clk_shift) < (unsigned)GBM_CLK_SLOTS) {
            stamp0[2] = (unsigned long long)(a.clk + 2 * (blockIdx.x >> a.clk_shift));
            stamp0[0] = __builtin_amdgcn_s_memtime();
            stamp0[1] = __builtin_amdgcn_s_memrealtime();
        }
    }
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PPL;  // first column of this lane
    const bool in_row = PPL == 1 || i < a.ld;
    double S[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) S[p] = a.S0;
    if (in_row) {
        // The address of a store is (wave-uniform row pointer) + (lane offset): the row pointer advances on the scalar
        // unit and the lane offset never changes, so a step spends no vector instruction on addressing.
        double* row = a.out + (int64_t)blockIdx.x * (256 * PPL);
        const unsigned lane_bytes = threadIdx.x * (8u * PPL);
        // (hipcc picks the scalar-base form for the first store only and rebuilds a 64-bit vector address inside the
        // loop, hence the explicit instruction)
        auto store_row = [&](double* r) {
            if constexpr (PPL == 1) {
                asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(lane_bytes), "v"(S[0]), "s"(r) : "memory");
            } else {
                // The `s_nop 1` belongs to the store: on gfx940+ a store of more than 64 bits reads its data registers
                // AFTER issue, and a VALU instruction that overwrites them within the next two wait states corrupts what is
                // stored (ISA: required software-inserted wait states).  hipcc covers its own stores; an asm statement is
                // opaque to it, and S[] is overwritten by the next step's FMA right behind this store.  Round 4: a build
                // whose scheduling put that FMA first stored foreign low / high words in rows 0, 1 and 5 of the matrix
                // (1.2e-10 and percent-size errors in a quarter of the lanes, tools/diag_stamps.py) while the chain in
                // the registers stayed right; tools/check_asm_hazards.py finds the pattern in the assembly.
                const v2d v = {S[0], S[PPL - 1]};
                asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(lane_bytes), "v"(v), "s"(r) : "memory");
            }
        };
        store_row(row);
        // One Philox block feeds two Box-Muller pairs = four steps.  The main loop takes whole blocks (both pairs
        // written out, so no register copies select a pair); the tail runs pair by pair over the last <= 3 steps.
        PhiloxLane lane_rng[PPL];
#pragma unroll
        for (int p = 0; p < PPL; ++p) lane_rng[p] = philox_lane_setup(a.path_begin + (uint64_t)(i + p), STREAM_PRICE, a.k1);
        auto step = [&](const double (&e)[PPL]) {
#pragma unroll
            for (int p = 0; p < PPL; ++p)
                S[p] = MODE == 3 ? fm::scaled_exp_small6(S[p], e[p]) : SMALL ? fm::scaled_exp_small(S[p], e[p]) : fm::scaled_exp(S[p], e[p]);
            row += a.ld;
            store_row(row);
        };
        auto pair_exponents = [&](uint32_t wa, uint32_t wb, double& e0, double& e1) {  // drift + vol*z of two steps
            if (MODE >= 2) fm::box_muller_pair_affine_scaled(wa, wb, tab, a.ls, a.drift, e0, e1);
            else fm::box_muller_pair_affine(wa, wb, tab, a.vol, a.drift, e0, e1);
        };
        const int n_blocks = a.n_steps >> 2;
#pragma unroll 1
        for (int b = 0; b < n_blocks; ++b) {
            Philox4 w[PPL];
#pragma unroll
            for (int p = 0; p < PPL; ++p) w[p] = philox4x32_10_lane(lane_rng[p], (uint32_t)b, a.k0, a.k1);
            double e0[PPL], e1[PPL];
#pragma unroll
            for (int p = 0; p < PPL; ++p) pair_exponents(w[p].w0, w[p].w1, e0[p], e1[p]);
            step(e0);
            step(e1);
#pragma unroll
            for (int p = 0; p < PPL; ++p) pair_exponents(w[p].w2, w[p].w3, e0[p], e1[p]);
            step(e0);
            step(e1);
        }
        const int rest = a.n_steps & 3;
        if (rest) {  // wave-uniform
            Philox4 w[PPL];
#pragma unroll
            for (int p = 0; p < PPL; ++p) w[p] = philox4x32_10_lane(lane_rng[p], (uint32_t)n_blocks, a.k0, a.k1);
            double e0[PPL], e1[PPL];
#pragma unroll
            for (int p = 0; p < PPL; ++p) pair_exponents(w[p].w0, w[p].w1, e0[p], e1[p]);
            step(e0);
            if (rest >= 2) step(e1);
            if (rest == 3) {
#pragma unroll
                for (int p = 0; p < PPL; ++p) pair_exponents(w[p].w2, w[p].w3, e0[p], e1[p]);
                step(e0);
            }
        }
    }
    if (threadIdx.x == 0 && stamp0[2] != 0) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* dst = (unsigned long long*)stamp0[2];
        dst[0] = c1 - stamp0[0];
        dst[1] = r1 - stamp0[1];
    }
    if (PAYOFF) {
        __shared__ double red[2 * 4];
        double v[2] = {0.0, 0.0};
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            const double pay = (in_row && i + p < a.n_paths) ? payoff_of(a.is_call != 0, S[p], a.K) : 0.0;
            v[0] += pay;
            v[1] += pay * pay;
        }
        block_sum<2, 4>(v, red);
        if (threadIdx.x == 0) {
            a.partials[2 * (int64_t)blockIdx.x] = v[0];
            a.partials[2 * (int64_t)blockIdx.x + 1] = v[1];
        }
    }
}

// Terminal payoff over a stored matrix: reads the last row only (8 B per path).
__global__ __launch_bounds__(256) void k_payoff_sums(const double* last_row, int64_t n_paths, double K,
                                                     int is_call, double* partials) {
    __shared__ double red[2 * 4];
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_paths; i += (int64_t)gridDim.x * 256) {
        const double pay = payoff_of(is_call != 0, last_row[i], K);
        v[0] += pay;
        v[1] += pay * pay;
    }
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) {
        partials[2 * (int64_t)blockIdx.x] = v[0];
        partials[2 * (int64_t)blockIdx.x + 1] = v[1];
    }
}

// Fixed-order reduction of per-block partials {sum, sum of squares} (deterministic for a given count): one block for
// up to FIN_CHUNK pairs; beyond that (the fused payoff of a 10M-path generator leaves 39 063 pairs, an rBergomi one up
// to 250 000) a first launch sums chunks of FIN_CHUNK pairs, each in the same fixed order, and the single block finishes
// over the chunk sums -- 15.7 us of one block's strided reads become two launches of ~3 us.
constexpr int64_t FIN_CHUNK = 8192;

__device__ __forceinline__ void sum_pairs(const double* partials, int64_t n, double (&v)[2], double* red) {
    v[0] = 0.0;
    v[1] = 0.0;
    for (int64_t b = threadIdx.x; b < n; b += 1024) {
        v[0] += partials[2 * b];
        v[1] += partials[2 * b + 1];
    }
    block_sum<2, 16>(v, red);
}

__global__ __launch_bounds__(1024) void k_sum_chunks(const double* partials, int64_t n, double* chunk_out) {
    __shared__ double red[2 * 16];
    const int64_t begin = (int64_t)blockIdx.x * FIN_CHUNK;
    double v[2];
    sum_pairs(partials + 2 * begin, n - begin < FIN_CHUNK ? n - begin : FIN_CHUNK, v, red);
    if (threadIdx.x == 0) {
        chunk_out[2 * (int64_t)blockIdx.x] = v[0];
        chunk_out[2 * (int64_t)blockIdx.x + 1] = v[1];
    }
}

__global__ __launch_bounds__(1024) void k_finish_sums(const double* partials, int64_t n_blocks, double n_local,
                                                      double* out3) {
    __shared__ double red[2 * 16];
    double v[2];
    sum_pairs(partials, n_blocks, v, red);
    if (threadIdx.x == 0) {
        out3[0] = v[0];
        out3[1] = v[1];
        out3[2] = n_local;
    }
}

int finish_sums(mcg_ctx* ctx, int64_t n_blocks, int64_t n_local, double out3[3]) {
    double* d = ctx->scalars + SC_SUMS;
    const double* src = ctx->partials;
    int64_t n = n_blocks;
    if (n > FIN_CHUNK) {
        const int64_t n_chunks = (n + FIN_CHUNK - 1) / FIN_CHUNK;  // <= 2^31 / 8192
        int rc = ensure_cap(ctx, &ctx->fin_chunks, &ctx->fin_chunks_cap, (size_t)(2 * n_chunks));
        if (rc) return rc;
        TimedLaunch t(ctx, MCG_K_PAYOFF);
        hipLaunchKernelGGL(k_sum_chunks, dim3((unsigned)n_chunks), dim3(1024), 0, ctx->stream, src, n, ctx->fin_chunks);
        src = ctx->fin_chunks;
        n = n_chunks;
    }
    {
        TimedLaunch t(ctx, MCG_K_PAYOFF);
        hipLaunchKernelGGL(k_finish_sums, dim3(1), dim3(1024), 0, ctx->stream, src, n, (double)n_local, d);
    }
    MCG_HIP(hipGetLastError());
    if (ctx->allreduce) {
        if (ctx->allreduce(ctx->allreduce_user, d, 3, (void*)ctx->stream) != 0)
            return fail(MCG_ERR_COMM, "all-reduce of payoff sums failed");
    }
    MCG_HIP(hipMemcpyAsync(ctx->h_scalars + SC_SUMS, d, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    out3[0] = ctx->h_scalars[SC_SUMS];
    out3[1] = ctx->h_scalars[SC_SUMS + 1];
    out3[2] = ctx->h_scalars[SC_SUMS + 2];
    return MCG_OK;
}

template <bool PAYOFF, int PPL>
static void launch_gbm_mode(mcg_ctx* ctx, const GbmArgs& a, int mode, unsigned n_blocks) {
    const dim3 grid(n_blocks), block(256);
    if (mode == 3) hipLaunchKernelGGL((k_gbm_paths<PAYOFF, 3, PPL>), grid, block, 0, ctx->stream, a);
    else if (mode == 2) hipLaunchKernelGGL((k_gbm_paths<PAYOFF, 2, PPL>), grid, block, 0, ctx->stream, a);
    else if (mode == 1) hipLaunchKernelGGL((k_gbm_paths<PAYOFF, 1, PPL>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_gbm_paths<PAYOFF, 0, PPL>), grid, block, 0, ctx->stream, a);
}

// paths per lane: 2 once every CU has several workgroups of 512 paths to work on
static int gbm_paths_per_lane(const mcg_ctx* ctx, int64_t n_paths) {
    static const int forced = study_switch("MCG_GBM_PPL", 0);
    if (forced == 1 || forced == 2) return forced;
    return n_paths >= (int64_t)ctx->n_cus * 512 * 8 ? 2 : 1;
}

int launch_gbm(mcg_ctx* ctx, mcg_paths* P, uint64_t seed, double S0, double r, double sigma, double dt,
               bool want_payoff, double K, int is_call) {
    const int ppl = gbm_paths_per_lane(ctx, P->n_paths);
    const int64_t n_blocks = (P->n_paths + 256 * ppl - 1) / (256 * ppl);
    if (n_blocks > 0x7fffffffLL) return fail(MCG_ERR_INVALID, "n_paths too large for one launch");
    if ((P->ld & 255) != 0) return fail(MCG_ERR_INVALID, "path matrix rows must be padded to 256 columns");
    if (want_payoff) {
        int rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, (size_t)(2 * n_blocks));
        if (rc) return rc;
    }
    GbmArgs a;
    a.out = P->data;
    a.ld = P->ld;
    a.n_paths = P->n_paths;
    a.n_steps = P->n_steps;
    a.path_begin = P->path_begin;
    a.k0 = (uint32_t)seed;
    a.k1 = (uint32_t)(seed >> 32);
    a.S0 = S0;
    a.drift = (r - 0.5 * sigma * sigma) * dt;
    a.vol = sigma * std::sqrt(dt);
    a.K = K;
    a.is_call = is_call;
    a.partials = ctx->partials;
    a.log_tab = (const double2*)ctx->log_tab;
    // shader-clock stamps (armed launches only, mcg_generator_clock_arm): one workgroup in every 2^shift, the smallest shift
    // that needs at most GBM_CLK_SLOTS slots
    a.clk = nullptr;
    a.clk_first = 0;
    a.clk_shift = 0;
    ctx->clk_slots_used = 0;
    if (ctx->clk_armed && ctx->clk_stamps && n_blocks >= 64) {
        unsigned shift = 2;
        while (((n_blocks - 1) >> shift) >= GBM_CLK_SLOTS) ++shift;
        a.clk = ctx->clk_stamps;
        a.clk_shift = shift;
        a.clk_first = (1u << shift) / 2;   // (the middle of each stretch; the last stretch may end before it)
        ctx->clk_slots_used = (int)(((n_blocks - 1 - a.clk_first) >> shift) + 1);
        MCG_HIP(hipMemsetAsync(ctx->clk_stamps, 0, sizeof(unsigned long long) * 2 * GBM_CLK_SLOTS, ctx->stream));
    }
    {
        TimedLaunch t(ctx, MCG_K_GBM);
        const bool small = std::fabs(a.drift) + std::fabs(a.vol) * fm::MAX_ABS_NORMAL <= fm::SMALL_EXP_BOUND;
        const double reach = std::fabs(a.drift) + std::fabs(a.vol) * fm::MAX_ABS_NORMAL;
        // vol^2 (-2 ln u) must stay a normal positive double for modes 2 and 3
        const int mode = !small ? 0 : !(a.vol > 1e-100) ? 1 : reach <= fm::SMALL6_EXP_BOUND ? 3 : 2;
        a.ls = fm::make_log_scale(a.vol * a.vol);
        if (want_payoff) {
            if (ppl == 2) launch_gbm_mode<true, 2>(ctx, a, mode, (unsigned)n_blocks);
            else launch_gbm_mode<true, 1>(ctx, a, mode, (unsigned)n_blocks);
        } else {
            if (ppl == 2) launch_gbm_mode<false, 2>(ctx, a, mode, (unsigned)n_blocks);
            else launch_gbm_mode<false, 1>(ctx, a, mode, (unsigned)n_blocks);
        }
    }
    MCG_HIP(hipGetLastError());
    if (want_payoff) {
        int rc = finish_sums(ctx, n_blocks, P->n_paths, P->sums);
        if (rc) return rc;
        // sums[] now holds the (all-reduced, if a collective is installed) totals
        P->has_sums = true;
        P->sums_K = K;
        P->sums_is_call = is_call;
    }
    return MCG_OK;
}

// Median (and range) of the stamping workgroups' clocks of the last GBM launch: cycles / (100 MHz ticks) x 0.1 GHz.
int generator_clock(mcg_ctx* ctx, double* ghz_median, int* n_stamps, double* ghz_min, double* ghz_max) {
    if (ghz_median) *ghz_median = 0.0;
    if (n_stamps) *n_stamps = 0;
    if (ghz_min) *ghz_min = 0.0;
    if (ghz_max) *ghz_max = 0.0;
    if (!ctx->clk_stamps || ctx->clk_slots_used < 1) return MCG_OK;
    unsigned long long h[2 * GBM_CLK_SLOTS];
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    MCG_HIP(hipMemcpy(h, ctx->clk_stamps, sizeof h, hipMemcpyDeviceToHost));
    std::vector<double> g;
    for (int k = 0; k < ctx->clk_slots_used; ++k)
        if (h[2 * k + 1] > 100) g.push_back((double)h[2 * k] / (double)h[2 * k + 1] * 0.1);  // (a stamp shorter than 1 us says nothing)
    if (g.empty()) return MCG_OK;
    std::sort(g.begin(), g.end());
    if (ghz_median) *ghz_median = g[g.size() / 2];
    if (n_stamps) *n_stamps = (int)g.size();
    if (ghz_min) *ghz_min = g.front();
    if (ghz_max) *ghz_max = g.back();
    return MCG_OK;
}

int launch_payoff_sums(mcg_ctx* ctx, const mcg_paths* P, double K, int is_call, double out3[3]) {
    int64_t n_blocks = (P->n_paths + 255) / 256;
    if (n_blocks > 4096) n_blocks = 4096;
    int rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, (size_t)(2 * n_blocks));
    if (rc) return rc;
    {
        TimedLaunch t(ctx, MCG_K_PAYOFF);
        hipLaunchKernelGGL(k_payoff_sums, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream,
                           P->data + (int64_t)P->n_steps * P->ld, P->n_paths, K, is_call, ctx->partials);
    }
    MCG_HIP(hipGetLastError());
    return finish_sums(ctx, n_blocks, P->n_paths, out3);
}

}  // namespace mcg
