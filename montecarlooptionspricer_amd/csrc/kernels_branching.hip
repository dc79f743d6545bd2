// BranchingProcesses::PredictOptionPrice on a device-resident step-major path matrix (gfx950).
//
// Reference: /root/reference/src/models/BranchingProcessPricer.cpp:12-134.
//   lower bound (:41-72): per path, the first exercise date (t <= maturity) with a positive discounted payoff;
//   upper bound (:74-134): per path, max over exercise dates of max(discounted payoff now, continuation), where
//     the continuation at date t resamples `numBranches` paths uniformly at random and averages their best
//     discounted payoff over all LATER columns k (t_k <= maturity), :104-121;
//   price = midpoint (:37).
// The inner O(T) rescan per (path, date, branch) collapses to a lookup: with
//   F[j][p] = max_{k >= j, t_k <= maturity} e^{-r t_k} Payoff(S[k][p])   (suffix maximum, floored at 0)
// the reference's bestFut * e^{-r t} equals F[t_idx+1][rp].  k_branch_suffix builds F with one backward
// stream (read S, write F: 16 B per element); k_branch_bounds then does the O(N T B) random gathers inside row
// t_idx+1 of F (L2/MALL-resident rows) with Philox-drawn path indices (stream 2, counter = (path, date, branch/4)).
// The reference draws from an unseeded shared mt19937, so parity of the upper bound is statistical; the lower
// bound is deterministic.  Sharded use resamples within the local shard (the matrix is never exchanged).
#include "devmath.hpp"
#include "mcg_internal.hpp"

namespace mcg {

enum : uint32_t { STREAM_BRANCH = 2u };

// F[j][p] for j = n_cols-1 .. 0.  disc[j] = e^{-r j dt}; columns j >= n_dates (t_j > maturity) contribute 0.
__global__ __launch_bounds__(256) void k_branch_suffix(const double* S, double* F, int64_t ld, int64_t n, int n_cols,
                                                       int n_dates, const double* disc, double K, int is_call) {
    const bool call = is_call != 0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        double run = 0.0;
        for (int j = n_cols - 1; j >= 0; --j) {
            if (j < n_dates) {
                const double d = disc[j] * payoff_of(call, S[(int64_t)j * ld + p], K);
                if (d > run) run = d;
            }
            F[(int64_t)j * ld + p] = run;
        }
    }
}

struct BranchArgs {
    const double* S;
    const double* F;
    int64_t ld, n;
    uint64_t path_begin;
    uint32_t k0, k1;
    const int* ex;        // [n_ex] exercise column indices with t <= maturity (leading part of the list)
    const double* disc;   // [n_cols]
    int n_ex, ex_last, num_branches;
    int n_cols;           // rows of S and F: the continuation of date t gathers from row t+1 < n_cols only
    double K;
    int is_call;
};

// The dates are the OUTER loop: every thread carries the bounds of its BR_PPT paths in registers and all resident
// threads work on one exercise date at about the same time, so the device's gathers of that moment fall into ONE row of
// F (8 MB at a million paths: L2 / MALL) instead of being spread over the whole matrix (round 2: paths outermost, each
// thread walking its own dates: 7.7 ms for the 5 10^8 gathers of a 1M x 50 matrix; 5.4 ms this way).  A launch covers
// gridDim.x * 256 * BR_PPT paths starting at p0; larger shards take several launches.
constexpr int BR_PPT = 4;

__global__ __launch_bounds__(256) void k_branch_bounds(BranchArgs a, int64_t p0, double* partials) {
    __shared__ double red[2 * 4];
    const bool call = a.is_call != 0;
    const uint32_t n32 = (uint32_t)a.n;
    const int quads = (a.num_branches + 3) >> 2;
    const double inv_b = a.num_branches > 0 ? 1.0 / (double)a.num_branches : 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (a.n <= 0) {  // an empty shard still takes part in the collective that follows
        if (threadIdx.x == 0) partials[2 * (int64_t)blockIdx.x] = partials[2 * (int64_t)blockIdx.x + 1] = 0.0;
        return;
    }
    int64_t p[BR_PPT];
    bool live[BR_PPT], have_lower[BR_PPT];
    double lower[BR_PPT], upper[BR_PPT];
    PhiloxLane rng[BR_PPT];  // block numbers below are wave-uniform
#pragma unroll
    for (int q = 0; q < BR_PPT; ++q) {
        p[q] = p0 + (int64_t)blockIdx.x * 256 + threadIdx.x + q * stride;
        live[q] = p[q] < a.n;
        if (!live[q]) p[q] = a.n - 1;  // (reads stay in range; the result is dropped)
        have_lower[q] = false;
        lower[q] = upper[q] = 0.0;
        rng[q] = philox_lane_setup(a.path_begin + (uint64_t)p[q], STREAM_BRANCH, a.k1);
    }
    for (int e = 0; e < a.n_ex; ++e) {
        const int t_idx = a.ex[e];
        const double* rowS = a.S + (int64_t)t_idx * a.ld;
        const double dsc = a.disc[t_idx];
        // :104-121.  A trailing exercise index at or beyond the last column (the reference tolerates one behind its
        // `t > maturity` break, :97-99) leaves its `k` loop (:110) empty for t_idx = n_cols-1: continuation 0.
        const bool branch = t_idx < a.ex_last && t_idx + 1 < a.n_cols && a.num_branches > 0;
        const double* rowF = a.F + (int64_t)(branch ? t_idx + 1 : t_idx) * a.ld;
#pragma unroll
        for (int q = 0; q < BR_PPT; ++q) {
            const double now = dsc * payoff_of(call, rowS[p[q]], a.K);
            if (!have_lower[q] && now > 0.0) {  // :62-65, first positive discounted payoff
                lower[q] = now;
                have_lower[q] = true;
            }
            double better = now;
            if (branch) {
                double sum = 0.0;
                for (int k = 0; k < quads; ++k) {
                    const Philox4 w = philox4x32_10_lane(rng[q], (uint32_t)(e * quads + k), a.k0, a.k1);
                    const uint32_t ws[4] = {w.w0, w.w1, w.w2, w.w3};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (4 * k + s < a.num_branches) sum += rowF[__umulhi(ws[s], n32)];  // uniform on [0, n)
                    }
                }
                const double cont = sum * inv_b;
                if (cont > better) better = cont;
            }
            if (better > upper[q]) upper[q] = better;
        }
    }
    double v[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < BR_PPT; ++q) {
        if (live[q]) {
            v[0] += lower[q];
            v[1] += upper[q];
        }
    }
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) {
        partials[2 * (int64_t)blockIdx.x] = v[0];
        partials[2 * (int64_t)blockIdx.x + 1] = v[1];
    }
}

int run_branching(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                  int num_branches, const int* exercise_times, int n_ex_in, uint64_t seed, double* price, double* lower,
                  double* upper) {
    const int n_cols = P->n_steps + 1;
    if (P->n_paths >= (int64_t)0xFFFFFFFFLL) return fail(MCG_ERR_INVALID, "BranchingProcesses: too many paths in one shard");
    std::vector<double> disc((size_t)n_cols);
    int n_dates = 0;
    for (int j = 0; j < n_cols; ++j) {
        const double t = j * dt;
        if (!(t > maturity) && n_dates == j) n_dates = j + 1;
        disc[(size_t)j] = std::exp(-r * t);
    }
    std::vector<int> ex;
    for (int e = 0; e < n_ex_in; ++e) {
        const int t_idx = exercise_times[e];
        if (t_idx * dt > maturity) break;  // :57-59, :97-99 (tested before the column is touched, like the reference)
        if (t_idx < 0 || t_idx >= n_cols) return fail(MCG_ERR_INVALID, "BranchingProcesses: exercise time %d outside [0,%d)", t_idx, n_cols);
        ex.push_back(t_idx);
    }
    const int ex_last = exercise_times[n_ex_in - 1];  // exerciseTimes.back(), :104

    int grid = (int)std::min<int64_t>((P->n_paths + 255) / 256, (int64_t)ctx->n_cus * 8);
    if (grid < 1) grid = 1;
    int rc = ensure_cap(ctx, &ctx->weights, &ctx->weights_cap, (size_t)n_cols + (ex.size() + 1) / 2 + 1);
    if (rc) return rc;
    // k_branch_bounds: launches of at most 8 workgroups per CU, BR_PPT paths per thread
    const int64_t per_wg = 256 * (int64_t)BR_PPT;
    const int bgrid = (int)std::max<int64_t>(1, std::min<int64_t>((P->n_paths + per_wg - 1) / per_wg, (int64_t)ctx->n_cus * 8));
    const int64_t per_launch = (int64_t)bgrid * per_wg;
    const int64_t n_launches = std::max<int64_t>(1, (P->n_paths + per_launch - 1) / per_launch);
    rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, (size_t)2 * (size_t)std::max<int64_t>(grid, bgrid * n_launches));
    if (rc) return rc;
    void* Fbuf = nullptr;
    rc = pool_alloc(ctx, P->bytes, &Fbuf);
    if (rc) return rc;
    int* d_ex = reinterpret_cast<int*>(ctx->weights + n_cols);
    hipError_t e1 = hipMemcpyAsync(ctx->weights, disc.data(), (size_t)n_cols * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    hipError_t e2 = ex.empty() ? hipSuccess
                               : hipMemcpyAsync(d_ex, ex.data(), ex.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    hipError_t e3 = hipStreamSynchronize(ctx->stream);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        pool_release(ctx, Fbuf, P->bytes);
        return fail(MCG_ERR_HIP, "BranchingProcesses: table upload failed");
    }
    {
        TimedLaunch t(ctx, MCG_K_BRANCHING);
        hipLaunchKernelGGL(k_branch_suffix, dim3(grid), dim3(256), 0, ctx->stream, P->data, (double*)Fbuf, P->ld, P->n_paths,
                           n_cols, n_dates, ctx->weights, K, is_call);
    }
    BranchArgs a;
    a.S = P->data;
    a.F = (const double*)Fbuf;
    a.ld = P->ld;
    a.n = P->n_paths;
    a.path_begin = P->path_begin;
    a.k0 = (uint32_t)seed;
    a.k1 = (uint32_t)(seed >> 32);
    a.ex = d_ex;
    a.disc = ctx->weights;
    a.n_ex = (int)ex.size();
    a.ex_last = ex_last;
    a.n_cols = n_cols;
    a.num_branches = num_branches;
    a.K = K;
    a.is_call = is_call;
    for (int64_t l = 0; l < n_launches; ++l) {
        TimedLaunch t(ctx, MCG_K_BRANCHING);
        hipLaunchKernelGGL(k_branch_bounds, dim3(bgrid), dim3(256), 0, ctx->stream, a, l * per_launch, ctx->partials + 2 * l * bgrid);
    }
    double s[3];
    rc = finish_sums(ctx, (int64_t)bgrid * n_launches, P->n_paths, s);  // {sum lower, sum upper, N}
    pool_release(ctx, Fbuf, P->bytes);
    if (rc) return rc;
    if (!(s[2] >= 1.0)) return fail(MCG_ERR_EMPTY_PATHS, "BranchingProcesses: Empty pricePaths.");
    const double lo = s[0] / s[2], up = s[1] / s[2];
    if (lower) *lower = lo;
    if (upper) *upper = up;
    *price = 0.5 * (lo + up);  // :37
    return MCG_OK;
}

}  // namespace mcg
