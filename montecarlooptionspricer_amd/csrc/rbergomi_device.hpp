// Device-side body of the rBergomi generator, shared by the single-contract kernel
// (kernels_rbergomi.hip) and the batched driver-row kernel (kernels_batch.hip).  See
// kernels_rbergomi.hip for the algorithm.
#pragma once
#include "devmath.hpp"
#include "fastmath.hpp"

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

// One workgroup's share of the job: 64 paths (4 waves x 16), block `block_index` of the launch.
// smem: (M + RB_PAD + n_steps) doubles of LDS; tabs: the math tables, loaded here.
// Returns the final log-price of this lane's path; `live_lead` tells whether this lane is the one lane
// of its path (lane group 0) that should contribute a payoff.
__device__ __forceinline__ double rb_generate(const RbArgs& a, int64_t block_index, double* smem, fm::Tables* tabs,
                                              bool& live_lead) {
    const int M = a.M;
    double* kext = smem;               // [M + RB_PAD], kext[i] = kappa[(i - 16) mod M]
    double* comp = smem + M + RB_PAD;  // [n_steps]
    for (int i = threadIdx.x; i < M + RB_PAD; i += 256) kext[i] = a.kappa[(i - 16 + 16 * M) & (M - 1)];
    for (int i = threadIdx.x; i < a.n_steps; i += 256) comp[i] = a.comp[i];
    fm::load_tables(tabs, a.log_tab);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t p = block_index * 64 + wave * 16 + c;
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
        rb_pass<RB_NT>(a, kext, comp, tabs, n_base, g, c, a_off, id, live, col, logS);
    const int tiles_left = (a.n_steps - n_base + 15) >> 4;  // 0..RB_NT-1, wave-uniform
    if (tiles_left > 8) rb_pass<16>(a, kext, comp, tabs, n_base, g, c, a_off, id, live, col, logS);
    else if (tiles_left > 4) rb_pass<8>(a, kext, comp, tabs, n_base, g, c, a_off, id, live, col, logS);
    else if (tiles_left > 2) rb_pass<4>(a, kext, comp, tabs, n_base, g, c, a_off, id, live, col, logS);
    else if (tiles_left > 0) rb_pass<2>(a, kext, comp, tabs, n_base, g, c, a_off, id, live, col, logS);
    live_lead = live && g == 0;
    return logS;
}

}  // namespace mcg
