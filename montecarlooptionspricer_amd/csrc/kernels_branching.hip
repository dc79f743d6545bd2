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
// F instead of being spread over the whole matrix (round 2: paths outermost, each thread walking its own dates: 7.7 ms
// for the 5 10^8 gathers of a 1M x 50 matrix; 5.4 ms this way).  A launch covers gridDim.x * 256 * BR_PPT paths
// starting at p0; larger shards take several launches.
//
// Round 4 -- the row in SLICES.  The counters (profiles/r04_branching_counters_before.json) put numbers on where a gather is
// served: a row that fits an XCD's 4 MiB of L2 (250k paths: 2 MB) is gathered at 188 G/s with 94 % L2 hits; the 8 MB row
// of a million paths at 92 G/s with 34 %, two thirds of the gathers going out to the fabric as one 64-byte request
// each (61 G requests/s: the rate at which the memory side serves random sectors, whatever cache they hit); the 32 MB
// row of 4M paths at 55 G/s with 4.5 %.  So the row is walked in slices of 2 MB: a thread draws the indices of all its
// branches for the date ONCE, keeps them in registers (4 QUADS words per path) and then passes over the slices, gathering
// in pass s only the indices that fall into slice s -- every lane of the device is in the same couple of slices at a
// time, which the L2 of every XCD holds.  The loads stay unconditional (a lane whose index lies elsewhere reads the
// slice's first element -- one line for all of them -- and adds zero), so all of a pass's loads are in flight together.
// Rows beyond 16 MB take 8 slices of n/8 (more passes would cost more instructions than the misses they avoid).
constexpr int BR_PPT = 4;
constexpr int BR_SLICE_SHIFT = 18;  // 2^18 paths = 2 MB of a row of F
constexpr int BR_MAX_SLICES = 8;
constexpr int BR_DATE_MAX_SLICES = 4;  // per-date launches pay for rows of up to four slices (8 MB: a million paths), see run_branching
constexpr int BR_WGS_PER_CU = 4;    // resident workgroups per CU the register budget is held to: a launch's grid is one resident wave

struct BranchLane {  // one thread's BR_PPT paths
    int64_t p[BR_PPT];
    bool live[BR_PPT], have_lower[BR_PPT];
    double lower[BR_PPT], upper[BR_PPT];
    PhiloxLane rng[BR_PPT];  // block numbers are wave-uniform
};

__device__ __forceinline__ void branch_lane_setup(BranchLane& t, const BranchArgs& a, int64_t p0) {
    const int64_t stride = (int64_t)gridDim.x * 256;
#pragma unroll
    for (int q = 0; q < BR_PPT; ++q) {
        t.p[q] = p0 + (int64_t)blockIdx.x * 256 + threadIdx.x + q * stride;
        t.live[q] = t.p[q] < a.n;
        if (!t.live[q]) t.p[q] = a.n - 1;  // (reads stay in range; the result is dropped)
        t.have_lower[q] = false;
        t.lower[q] = t.upper[q] = 0.0;
        t.rng[q] = philox_lane_setup(a.path_begin + (uint64_t)t.p[q], STREAM_BRANCH, a.k1);
    }
}

__device__ __forceinline__ void branch_lane_finish(const BranchLane& t, double* red, double* partials) {
    double v[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < BR_PPT; ++q) {
        if (t.live[q]) {
            v[0] += t.lower[q];
            v[1] += t.upper[q];
        }
    }
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) {
        partials[2 * (int64_t)blockIdx.x] = v[0];
        partials[2 * (int64_t)blockIdx.x + 1] = v[1];
    }
}

// QUADS = Philox blocks per path and date = ceil(num_branches / 4) <= 3 (the driver's 10 branches: 3); more branches than
// twelve take k_branch_bounds_any, which gathers as it draws.
template <int QUADS>
__global__ __launch_bounds__(256, BR_WGS_PER_CU) void k_branch_bounds(BranchArgs a, int64_t p0, double* partials, int slice_shift, int n_slices) {
    __shared__ double red[2 * 4];
    const bool call = a.is_call != 0;
    const uint32_t n32 = (uint32_t)a.n;
    const double inv_b = a.num_branches > 0 ? 1.0 / (double)a.num_branches : 0.0;
    if (a.n <= 0) {  // an empty shard still takes part in the collective that follows
        if (threadIdx.x == 0) partials[2 * (int64_t)blockIdx.x] = partials[2 * (int64_t)blockIdx.x + 1] = 0.0;
        return;
    }
    BranchLane t;
    branch_lane_setup(t, a, p0);
    for (int e = 0; e < a.n_ex; ++e) {
        const int t_idx = a.ex[e];
        const double* rowS = a.S + (int64_t)t_idx * a.ld;
        const double dsc = a.disc[t_idx];
        // :104-121.  A trailing exercise index at or beyond the last column (the reference tolerates one behind its
        // `t > maturity` break, :97-99) leaves its `k` loop (:110) empty for t_idx = n_cols-1: continuation 0.
        const bool branch = t_idx < a.ex_last && t_idx + 1 < a.n_cols && a.num_branches > 0;
        const double* rowF = a.F + (int64_t)(branch ? t_idx + 1 : t_idx) * a.ld;
        double now[BR_PPT];
#pragma unroll
        for (int q = 0; q < BR_PPT; ++q) {
            now[q] = dsc * payoff_of(call, rowS[t.p[q]], a.K);
            if (!t.have_lower[q] && now[q] > 0.0) {  // :62-65, first positive discounted payoff
                t.lower[q] = now[q];
                t.have_lower[q] = true;
            }
        }
        double sum[BR_PPT];
#pragma unroll
        for (int q = 0; q < BR_PPT; ++q) sum[q] = 0.0;
        if (branch) {  // (wave-uniform)
            uint32_t idx[BR_PPT][4 * QUADS];  // uniform on [0, n); a branch beyond num_branches gets an index in no slice
#pragma unroll
            for (int q = 0; q < BR_PPT; ++q) {
#pragma unroll
                for (int k = 0; k < QUADS; ++k) {
                    const Philox4 w = philox4x32_10_lane(t.rng[q], (uint32_t)(e * QUADS + k), a.k0, a.k1);
                    const uint32_t ws[4] = {w.w0, w.w1, w.w2, w.w3};
#pragma unroll
                    for (int s = 0; s < 4; ++s) idx[q][4 * k + s] = 4 * k + s < a.num_branches ? __umulhi(ws[s], n32) : 0xFFFFFFFFu;
                }
            }
            for (int sl = 0; sl < n_slices; ++sl) {
                const uint32_t first = (uint32_t)sl << slice_shift;  // (< n: the slices cover [0, n))
#pragma unroll
                for (int q = 0; q < BR_PPT; ++q) {
#pragma unroll
                    for (int b = 0; b < 4 * QUADS; ++b) {
                        const bool in = (idx[q][b] >> slice_shift) == (uint32_t)sl;
                        const double v = rowF[in ? idx[q][b] : first];
                        sum[q] += in ? v : 0.0;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < BR_PPT; ++q) {
            double better = now[q];
            const double cont = sum[q] * inv_b;
            if (branch && cont > better) better = cont;
            if (better > t.upper[q]) t.upper[q] = better;
        }
    }
    branch_lane_finish(t, red, partials);
}

// ONE EXERCISE DATE per launch (rows of F beyond one slice): inside one launch of ~100 us the workgroups stay within a
// slice or two of each other, which a whole sweep's worth of drift in k_branch_bounds does not -- its 1M-path row was still
// gathered with 42 % L2 hits (profiles/r04_branching_counters_before.json).  The bounds of a path travel between the launches in
// `state` = {lower, upper} per path (lower > 0 <=> the first positive payoff has been seen): 32 B per path and date.
template <int QUADS>
__global__ __launch_bounds__(256, BR_WGS_PER_CU) void k_branch_date(BranchArgs a, int64_t p0, int e, double2* state, int first_date,
                                                                   int slice_shift, int n_slices) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const bool call = a.is_call != 0;
    const uint32_t n32 = (uint32_t)a.n;
    const double inv_b = 1.0 / (double)a.num_branches;
    const int t_idx = a.ex[e];
    const double* rowS = a.S + (int64_t)t_idx * a.ld;
    // :104-121.  No continuation at the list's last date nor at an index on the last column (k_branch_bounds): n_slices = 0
    const bool branch = n_slices > 0;
    const double* rowF = a.F + (int64_t)(branch ? t_idx + 1 : t_idx) * a.ld;
    const double dsc = a.disc[t_idx];
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t p[BR_PPT];
    bool live[BR_PPT];
    double now[BR_PPT], sum[BR_PPT];
    v2d st[BR_PPT];
    uint32_t idx[BR_PPT][4 * QUADS];
#pragma unroll
    for (int q = 0; q < BR_PPT; ++q) {
        p[q] = p0 + (int64_t)blockIdx.x * 256 + threadIdx.x + q * stride;
        live[q] = p[q] < a.n;
        if (!live[q]) p[q] = a.n - 1;
        st[q] = first_date ? v2d{0.0, 0.0} : __builtin_nontemporal_load(reinterpret_cast<const v2d*>(state) + p[q]);
        now[q] = dsc * payoff_of(call, __builtin_nontemporal_load(rowS + p[q]), a.K);
        sum[q] = 0.0;
        if (branch) {  // (wave-uniform)
            const PhiloxLane rng = philox_lane_setup(a.path_begin + (uint64_t)p[q], STREAM_BRANCH, a.k1);
#pragma unroll
            for (int k = 0; k < QUADS; ++k) {
                const Philox4 w = philox4x32_10_lane(rng, (uint32_t)(e * QUADS + k), a.k0, a.k1);
                const uint32_t ws[4] = {w.w0, w.w1, w.w2, w.w3};
#pragma unroll
                for (int s = 0; s < 4; ++s) idx[q][4 * k + s] = 4 * k + s < a.num_branches ? __umulhi(ws[s], n32) : 0xFFFFFFFFu;
            }
        }
    }
    for (int sl = 0; sl < n_slices; ++sl) {
        const uint32_t first = (uint32_t)sl << slice_shift;
#pragma unroll
        for (int q = 0; q < BR_PPT; ++q) {
#pragma unroll
            for (int b = 0; b < 4 * QUADS; ++b) {
                const bool in = (idx[q][b] >> slice_shift) == (uint32_t)sl;
                const double v = rowF[in ? idx[q][b] : first];
                sum[q] += in ? v : 0.0;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < BR_PPT; ++q) {
        if (!(st[q].x > 0.0) && now[q] > 0.0) st[q].x = now[q];  // :62-65, first positive discounted payoff
        double better = now[q];
        const double cont = sum[q] * inv_b;
        if (branch && cont > better) better = cont;
        if (better > st[q].y) st[q].y = better;
        if (live[q]) __builtin_nontemporal_store(st[q], reinterpret_cast<v2d*>(state) + p[q]);
    }
}

// ONE EXERCISE DATE per launch, rows of MANY slices (round 5; beyond four slices = more than a million paths): the walk
// over the slices with every index tested in every pass costs a load instruction per index and pass however few lanes
// take part (4M x 50, 16 slices: 44.7 ms that way, 29.6 in one launch with 8 slices).  Here every thread SORTS its
// 4 QUADS PPT indices by slice first -- a counting sort into its own column of an LDS array (bins[pos][thread]: no two
// threads share an entry, no atomics, no barrier), the sixteen counters packed six bits each into two 64-bit registers
// -- and then walks the slices in order, gathering from its bin exactly the indices that fall into the slice: a pass costs
// as many load instructions as the fullest bin among the wave's lanes (~9 of 40 entries at sixteen slices) instead of 40,
// and every lane of the device is still in the same slice or two of the row at a time.  The owner of a gathered value (which
// of the thread's four paths) rides in the top two bits of the stored index -- rows of up to 2^30 - 2 paths -- and the value
// is added to that path's sum in the thread's own LDS cell (ds_add_f64 on a private address: program order, deterministic;
// selecting among PPT register sums costs four compare-select-add groups per gathered value instead).
// The sum over a path's branches is formed in slice order -- like k_branch_bounds / k_branch_date, not in branch order.
// What bounds this kernel is how many paths are RESIDENT at a time: every generation of resident workgroups pulls every slice
// of the row into the L2 of every XCD once (4M x 50: 32 MB x 8 XCDs per generation and date), so the bins hold the
// num_branches indices a path really has (dynamic LDS, 40 B per path at the driver's 10 branches + 8 B of sum), four paths
// per thread: 48 KiB per workgroup, three per CU, 786 432 paths per generation (three paths per thread: 38 KiB, 589 824).
// Three or four paths per thread (run_branching picks per shape: the generation whose LAST launch is fuller -- a nearly
// empty generation still pulls the whole row through every L2).
inline size_t brb_lds_bytes(int num_branches, int ppt) { return (size_t)256 * ((size_t)num_branches * ppt * sizeof(uint32_t) + 4 * sizeof(double)); }
template <int QUADS, int PPT>
__global__ __launch_bounds__(256, 3) void k_branch_date_binned(BranchArgs a, int64_t p0, int e, double2* state, int first_date,
                                                               int slice_shift, int n_slices) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    static_assert(12 * PPT <= 63 && PPT <= 4, "bin positions are six-bit fields; the owner is two bits");
    extern __shared__ double brb_lds[];
    double (*sums)[256] = reinterpret_cast<double (*)[256]>(brb_lds);                // [owner][thread]
    uint32_t (*bins)[256] = reinterpret_cast<uint32_t (*)[256]>(brb_lds + 4 * 256);  // [pos][thread], pos < num_branches PPT
    const bool call = a.is_call != 0;
    const uint32_t n32 = (uint32_t)a.n;
    const double inv_b = 1.0 / (double)a.num_branches;
    const int t_idx = a.ex[e];
    const double* rowS = a.S + (int64_t)t_idx * a.ld;
    const bool branch = n_slices > 0;  // (no continuation at the list's last date nor at an index on the last column)
    const double* rowF = a.F + (int64_t)(branch ? t_idx + 1 : t_idx) * a.ld;
    const double dsc = a.disc[t_idx];
    const int64_t stride = (int64_t)gridDim.x * 256;
    const unsigned tid = threadIdx.x;
    int64_t p[PPT];
    bool live[PPT];
    double now[PPT];
    v2d st[PPT];
    // six-bit fields: slices 0..9 in lo, 10..15 in hi
    uint64_t f_lo = 0, f_hi = 0;
    auto field_shift = [](uint32_t sl) { return 6u * (sl >= 10u ? sl - 10u : sl); };
    auto bump = [&](uint32_t sl) -> uint32_t {  // returns the field's value, then adds one to it
        const bool up = sl >= 10u;
        const uint32_t sh = field_shift(sl);
        const uint32_t old = (uint32_t)((up ? f_hi : f_lo) >> sh) & 63u;
        const uint64_t one = 1ull << sh;
        f_lo += up ? 0ull : one;
        f_hi += up ? one : 0ull;
        return old;
    };
    uint32_t idx[PPT][4 * QUADS];
#pragma unroll
    for (int q = 0; q < 4; ++q) sums[q][tid] = 0.0;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        p[q] = p0 + (int64_t)blockIdx.x * 256 + threadIdx.x + q * stride;
        live[q] = p[q] < a.n;
        if (!live[q]) p[q] = a.n - 1;
        st[q] = first_date ? v2d{0.0, 0.0} : __builtin_nontemporal_load(reinterpret_cast<const v2d*>(state) + p[q]);
        now[q] = dsc * payoff_of(call, __builtin_nontemporal_load(rowS + p[q]), a.K);
        if (branch) {  // (wave-uniform)
            const PhiloxLane rng = philox_lane_setup(a.path_begin + (uint64_t)p[q], STREAM_BRANCH, a.k1);
#pragma unroll
            for (int k = 0; k < QUADS; ++k) {
                const Philox4 w = philox4x32_10_lane(rng, (uint32_t)(e * QUADS + k), a.k0, a.k1);
                const uint32_t ws[4] = {w.w0, w.w1, w.w2, w.w3};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    idx[q][4 * k + s] = __umulhi(ws[s], n32);
                    if (4 * k + s < a.num_branches) (void)bump(idx[q][4 * k + s] >> slice_shift);  // count
                }
            }
        }
    }
    if (branch) {
        // counts -> first positions (exclusive prefix over the slices)
        {
            uint64_t o_lo = 0, o_hi = 0;
            uint32_t run = 0;
#pragma unroll
            for (uint32_t sl = 0; sl < 16u; ++sl) {
                const uint32_t sh = field_shift(sl);
                const uint32_t c = (uint32_t)((sl >= 10u ? f_hi : f_lo) >> sh) & 63u;
                if (sl >= 10u) o_hi |= (uint64_t)run << sh;
                else o_lo |= (uint64_t)run << sh;
                run += c;
            }
            f_lo = o_lo;
            f_hi = o_hi;
        }
        // place: afterwards every field holds the END of its slice's bin
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
#pragma unroll
            for (int b = 0; b < 4 * QUADS; ++b) {
                if (b < a.num_branches) {
                    const uint32_t pos = bump(idx[q][b] >> slice_shift);
                    bins[pos][tid] = idx[q][b] | ((uint32_t)q << 30);
                }
            }
        }
        // walk the slices; within a slice four entries of the bin per trip (all four loads in flight together).  One thread's
        // LDS accesses execute in program order: its reads of bins[] see its own writes, its adds to sums[] line up.
        uint32_t pos = 0;
#pragma unroll 1
        for (uint32_t sl = 0; sl < (uint32_t)n_slices; ++sl) {
            const uint32_t end = (uint32_t)((sl >= 10u ? f_hi : f_lo) >> field_shift(sl)) & 63u;
            while (__builtin_amdgcn_ballot_w64(pos < end) != 0ull) {
                uint32_t w[4];
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = pos + u < end ? bins[pos + u][tid] : 0xFFFFFFFFu;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = w[u] != 0xFFFFFFFFu ? rowF[w[u] & 0x3FFFFFFFu] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u)  // (an empty entry adds 0.0 to the fourth path's cell)
                    __hip_atomic_fetch_add(&sums[w[u] >> 30][tid], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pos = pos + 4u < end ? pos + 4u : end;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (!(st[q].x > 0.0) && now[q] > 0.0) st[q].x = now[q];  // :62-65, first positive discounted payoff
        double better = now[q];
        const double cont = sums[q][tid] * inv_b;
        if (branch && cont > better) better = cont;
        if (better > st[q].y) st[q].y = better;
        if (live[q]) __builtin_nontemporal_store(st[q], reinterpret_cast<v2d*>(state) + p[q]);
    }
}

// ---- XCD-AFFINE per-date gathers (round 6): rows of F beyond ~14 MB (more than 1.8M paths) ------------------
// The binned kernel above pulls every slice of the row through the L2 of every XCD once per generation of resident paths
// (4M x 50: 200 MB beyond L2 per launch, 53 % hits, profiles/r05_branching_binned_counters.json).  Here the row is dealt out to
// the XCDs in sixteen equal chunks: workgroup w (XCD w & 7 -- workgroups are handed to the XCDs round-robin) draws ALL the
// indices of its tile's paths and gathers ONLY those whose chunk belongs to its XCD, so an XCD's L2 only ever sees its eighth
// of the row (92 % hits at 4M paths: a 4 MB share in a 4 MB L2); every tile is worked on by eight workgroups, one per XCD,
// each leaving its partial sum per path in a cell of its own (cells[xcd][path]); k_branch_date_xcd_finish adds the eight
// cells in XCD order (deterministic: the same bits every run) and updates the bounds.  Costs: the Philox draws eight times
// over -- the gather kernel is VALU-bound, 0.83 busy --, 64 B of cells written and 64 + 40 B read per path and date.
// 4M x 50: 21.8 -> 15.1 ms, 3M x 50: 16.0 -> 11.1, 8M x 20: 21.4 -> 16.9, 2M x 50: 7.9 -> 7.5; 1.2M x 50 LOSES (4.19 -> 4.61:
// its row fits two L2s), hence the threshold (gpurun_out/r6o_branch_xcd.log; counters: profiles/r06_branching_xcd_counters.json).
// Tried on the way: the finishing pass folded into the next date's gather launch (an eighth of the tile per workgroup, first
// wave): 15.7 -> 17.1 ms, dropped; chunks of 2^13 / 2^15 / 2^18 paths taken from the INDEX instead of the word's top bits:
// 17.3 / 16.6 / 15.7 ms, and unbalanced whenever n is not a multiple of eight chunks (gpurun_out/r6n_branch_xcd.log).
constexpr int BRX_PPT = 2;

// bounds of path p after exercise date e: its eight cells added in XCD order (the same bits every run), then :62-65 and :123-127
__device__ __forceinline__ void brx_finish_path(const BranchArgs& a, int e, const double* cells, int64_t ldc, double2* state, int first_date,
                                                int branch, int64_t p) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const bool call = a.is_call != 0;
    const int t_idx = a.ex[e];
    v2d st = first_date ? v2d{0.0, 0.0} : __builtin_nontemporal_load(reinterpret_cast<const v2d*>(state) + p);
    const double now = a.disc[t_idx] * payoff_of(call, __builtin_nontemporal_load(a.S + (int64_t)t_idx * a.ld + p), a.K);
    double better = now;
    if (branch) {
        double c = 0.0;
#pragma unroll
        for (int x = 0; x < 8; ++x) c += __builtin_nontemporal_load(cells + (int64_t)x * ldc + p);
        const double cont = c * (1.0 / (double)a.num_branches);
        if (cont > better) better = cont;
    }
    if (!(st.x > 0.0) && now > 0.0) st.x = now;
    if (better > st.y) st.y = better;
    __builtin_nontemporal_store(st, reinterpret_cast<v2d*>(state) + p);
}

// A tile = 512 paths; workgroup w works on tile w >> 3 for XCD w & 7.  Which XCD a draw belongs to is read off the TOP bits of its
// Philox word (index = umulhi(word, n): the word's top four bits cut the row into sixteen equal chunks, chunk c to XCD c & 7 --
// two chunks of n / 16 paths per XCD whatever n is), so a draw that is not this XCD's costs a bit-field extract, a compare and one
// add-with-carry into the thread's mask; the words go to LDS four at a time and only the few that are kept are turned into indices.
template <int QUADS>
__global__ __launch_bounds__(256) void k_branch_date_xcd(BranchArgs a, int e, double* cells, int64_t ldc) {
    __shared__ uint4 buf[QUADS * BRX_PPT][256];  // [Philox block of the thread][thread]: a thread only ever reads its own column
    constexpr int DRAWS = 4 * QUADS * BRX_PPT;
    const unsigned tid = threadIdx.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const int64_t tile = (int64_t)(blockIdx.x >> 3);
    const uint32_t n32 = (uint32_t)a.n;
    const int t_idx = a.ex[e];
    const double* rowF = a.F + (int64_t)(t_idx + 1) * a.ld;
    // draws that exist: branch b < num_branches of either path; draw i (in drawing order) sits at bit DRAWS - 1 - i of the mask
    uint32_t exists = 0;
#pragma unroll
    for (int i = 0; i < DRAWS; ++i) exists |= ((i % (4 * QUADS)) < a.num_branches ? 1u : 0u) << (DRAWS - 1 - i);
    uint32_t m = 0;
    int64_t p[BRX_PPT];
#pragma unroll
    for (int q = 0; q < BRX_PPT; ++q) {
        p[q] = tile * (256 * BRX_PPT) + tid + q * 256;
        const bool live = p[q] < a.n;
        const PhiloxLane rng = philox_lane_setup(a.path_begin + (uint64_t)(live ? p[q] : a.n - 1), STREAM_BRANCH, a.k1);
#pragma unroll
        for (int k = 0; k < QUADS; ++k) {
            const Philox4 w = philox4x32_10_lane(rng, (uint32_t)(e * QUADS + k), a.k0, a.k1);
            buf[q * QUADS + k][tid] = make_uint4(w.w0, w.w1, w.w2, w.w3);
            const uint32_t ws[4] = {w.w0, w.w1, w.w2, w.w3};
#pragma unroll
            for (int s = 0; s < 4; ++s) m = m + m + ((((ws[s] >> 28) & 7u) == xcd && live) ? 1u : 0u);
        }
    }
    m &= exists;
    double sum[BRX_PPT];
#pragma unroll
    for (int q = 0; q < BRX_PPT; ++q) sum[q] = 0.0;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(&buf[0][0]);
    auto word_of = [&](uint32_t bit) {  // the Philox word of the draw at mask bit `bit`
        const uint32_t i = (uint32_t)(DRAWS - 1) - bit;
        return words[((i >> 2) * 256u + tid) * 4u + (i & 3u)];
    };
    while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {  // two of the thread's own gathers per trip
        const bool h0 = m != 0u;
        const uint32_t b0 = h0 ? (uint32_t)__builtin_ctz(m) : 0u;
        m &= m - (h0 ? 1u : 0u);
        const bool h1 = m != 0u;
        const uint32_t b1 = h1 ? (uint32_t)__builtin_ctz(m) : 0u;
        m &= m - (h1 ? 1u : 0u);
        const uint32_t w0 = word_of(b0), w1 = word_of(b1);
        const double v0 = h0 ? rowF[__umulhi(w0, n32)] : 0.0;
        const double v1 = h1 ? rowF[__umulhi(w1, n32)] : 0.0;
        // (bits >= 4 QUADS belong to the thread's FIRST path: it was drawn first)
        if (b0 >= 4u * QUADS) sum[0] += v0;
        else sum[BRX_PPT - 1] += v0;
        if (b1 >= 4u * QUADS) sum[0] += v1;
        else sum[BRX_PPT - 1] += v1;
    }
#pragma unroll
    for (int q = 0; q < BRX_PPT; ++q)
        if (p[q] < a.n) __builtin_nontemporal_store(sum[q], cells + (int64_t)xcd * ldc + p[q]);
}

__global__ __launch_bounds__(256) void k_branch_date_xcd_finish(BranchArgs a, int e, const double* cells, int64_t ldc, double2* state, int first_date,
                                                                int branch) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < a.n; p += (int64_t)gridDim.x * 256)
        brx_finish_path(a, e, cells, ldc, state, first_date, branch, p);
}

// sum of {lower, upper} over the paths -> partials[grid][2]
__global__ __launch_bounds__(256) void k_branch_finish(const double2* state, int64_t n, double* partials) {
    __shared__ double red[2 * 4];
    double v[2] = {0.0, 0.0};
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        const double2 st = state[p];
        v[0] += st.x;
        v[1] += st.y;
    }
    block_sum<2, 4>(v, red);
    if (threadIdx.x == 0) {
        partials[2 * (int64_t)blockIdx.x] = v[0];
        partials[2 * (int64_t)blockIdx.x + 1] = v[1];
    }
}

// any number of branches: the indices are used as they are drawn (no slices)
__global__ __launch_bounds__(256) void k_branch_bounds_any(BranchArgs a, int64_t p0, double* partials) {
    __shared__ double red[2 * 4];
    const bool call = a.is_call != 0;
    const uint32_t n32 = (uint32_t)a.n;
    const int quads = (a.num_branches + 3) >> 2;
    const double inv_b = a.num_branches > 0 ? 1.0 / (double)a.num_branches : 0.0;
    if (a.n <= 0) {
        if (threadIdx.x == 0) partials[2 * (int64_t)blockIdx.x] = partials[2 * (int64_t)blockIdx.x + 1] = 0.0;
        return;
    }
    BranchLane t;
    branch_lane_setup(t, a, p0);
    for (int e = 0; e < a.n_ex; ++e) {
        const int t_idx = a.ex[e];
        const double* rowS = a.S + (int64_t)t_idx * a.ld;
        const double dsc = a.disc[t_idx];
        const bool branch = t_idx < a.ex_last && t_idx + 1 < a.n_cols && a.num_branches > 0;
        const double* rowF = a.F + (int64_t)(branch ? t_idx + 1 : t_idx) * a.ld;
#pragma unroll
        for (int q = 0; q < BR_PPT; ++q) {
            const double now = dsc * payoff_of(call, rowS[t.p[q]], a.K);
            if (!t.have_lower[q] && now > 0.0) {
                t.lower[q] = now;
                t.have_lower[q] = true;
            }
            double better = now;
            if (branch) {
                double sum = 0.0;
                for (int k = 0; k < quads; ++k) {
                    const Philox4 w = philox4x32_10_lane(t.rng[q], (uint32_t)(e * quads + k), a.k0, a.k1);
                    const uint32_t ws[4] = {w.w0, w.w1, w.w2, w.w3};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (4 * k + s < a.num_branches) sum += rowF[__umulhi(ws[s], n32)];  // uniform on [0, n)
                    }
                }
                const double cont = sum * inv_b;
                if (cont > better) better = cont;
            }
            if (better > t.upper[q]) t.upper[q] = better;
        }
    }
    branch_lane_finish(t, red, partials);
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
    // k_branch_bounds: launches of one resident wave of workgroups (all of them walk the dates -- and the slices of a
    // date's row -- together), BR_PPT paths per thread
    const int64_t per_wg = 256 * (int64_t)BR_PPT;
    const int bgrid = (int)std::max<int64_t>(1, std::min<int64_t>((P->n_paths + per_wg - 1) / per_wg, (int64_t)ctx->n_cus * BR_WGS_PER_CU));
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
    // slices of the gathered row: 2 MB each, at most BR_MAX_SLICES (then of n / BR_MAX_SLICES, rounded up to a power of two)
    int slice_shift = BR_SLICE_SHIFT;
    while ((((P->n_paths > 0 ? P->n_paths : 1) - 1) >> slice_shift) + 1 > BR_MAX_SLICES) ++slice_shift;
    const int n_slices = (int)((((P->n_paths > 0 ? P->n_paths : 1) - 1) >> slice_shift) + 1);
    const int quads = (num_branches + 3) / 4;
    int64_t n_partials = (int64_t)bgrid * n_launches;
    void* state = nullptr;
    size_t state_bytes = 0;
    // Rows of more than four slices (more than a million paths), up to 2^30 - 2 paths: per-date launches of the BINNED
    // kernel (k_branch_date_binned: every thread walks its indices sorted by slice), sixteen slices of >= 2 MB.
    int bshift = study_switch("MCG_BRANCH_BIN_SHIFT", BR_SLICE_SHIFT);  // (A/B builds: 17 = 1 MB slices while sixteen of them cover the row)
    while ((((P->n_paths > 0 ? P->n_paths : 1) - 1) >> bshift) + 1 > 16) ++bshift;
    const int b_slices = (int)((((P->n_paths > 0 ? P->n_paths : 1) - 1) >> bshift) + 1);
    const bool binned = b_slices > study_switch("MCG_BRANCH_BINNED_ABOVE", BR_DATE_MAX_SLICES) && quads >= 1 && quads <= 3 && !ex.empty() && P->n_paths < ((int64_t)1 << 30) - 1 &&
                        study_switch("MCG_BRANCH_BINNED", 1) != 0;
    constexpr int64_t BRX_MIN_PATHS = (int64_t)7 << 18;   // 1.835M paths = rows of 14 MB: below, the binned kernel is at least as fast
    const int xcd_switch = study_switch("MCG_BRANCH_XCD", -1);   // (A/B builds: 0 never, 1 always)
    if (binned && (xcd_switch < 0 ? P->n_paths >= BRX_MIN_PATHS : xcd_switch != 0)) {
        // XCD-affine gathers (k_branch_date_xcd): a gather launch and a finishing pass per exercise date
        const int64_t ldc = (P->n_paths + 511) / 512 * 512;
        void* cells = nullptr;
        const size_t cells_bytes = (size_t)8 * ldc * sizeof(double);
        state_bytes = (size_t)P->n_paths * sizeof(double2);
        rc = pool_alloc(ctx, state_bytes, &state);
        if (!rc) rc = pool_alloc(ctx, cells_bytes, &cells);
        if (rc) {
            pool_release(ctx, Fbuf, P->bytes);
            if (state) pool_release(ctx, state, state_bytes);
            return rc;
        }
        const int64_t tiles = (P->n_paths + 256 * BRX_PPT - 1) / (256 * BRX_PPT);
        {
            TimedLaunch t(ctx, MCG_K_BRANCHING, 2 * (int64_t)ex.size() + 1);
            for (int e = 0; e < (int)ex.size(); ++e) {
                const int t_idx = ex[(size_t)e];
                const bool branch = t_idx < ex_last && t_idx + 1 < n_cols;
                if (branch) {
                    const dim3 g((unsigned)(tiles * 8)), b(256);
                    if (quads == 1) hipLaunchKernelGGL(k_branch_date_xcd<1>, g, b, 0, ctx->stream, a, e, (double*)cells, ldc);
                    else if (quads == 2) hipLaunchKernelGGL(k_branch_date_xcd<2>, g, b, 0, ctx->stream, a, e, (double*)cells, ldc);
                    else hipLaunchKernelGGL(k_branch_date_xcd<3>, g, b, 0, ctx->stream, a, e, (double*)cells, ldc);
                }
                hipLaunchKernelGGL(k_branch_date_xcd_finish, dim3(grid), dim3(256), 0, ctx->stream, a, e, (const double*)cells, ldc, (double2*)state,
                                   e == 0, branch ? 1 : 0);
            }
            hipLaunchKernelGGL(k_branch_finish, dim3(grid), dim3(256), 0, ctx->stream, (const double2*)state, P->n_paths, ctx->partials);
        }
        n_partials = grid;
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            pool_release(ctx, Fbuf, P->bytes);
            pool_release(ctx, state, state_bytes);
            pool_release(ctx, cells, cells_bytes);
            return fail(MCG_ERR_HIP, "BranchingProcesses: kernel launch failed");
        }
        pool_release(ctx, cells, cells_bytes);
    } else if (binned) {
        // Paths per thread: a generation is three resident workgroups per CU of 256 x PPT paths, and a launch that is nearly
        // empty still walks -- and pulls through every XCD's L2 -- the whole row.  A/B on one board (round 5): 4M x 50
        // 21.4 ms at three paths per thread (6.8 generations) against 23.4 at four (5.1); 2M x 50 8.76 against 8.12
        // (3.4 / 2.5 generations).  So: whichever leaves the fuller last generation.
        const double gen4 = (double)P->n_paths / (double)((int64_t)ctx->n_cus * 3 * 256 * 4), gen3 = (double)P->n_paths / (double)((int64_t)ctx->n_cus * 3 * 256 * 3);
        const int ppt = std::ceil(gen3) - gen3 < std::ceil(gen4) - gen4 ? 3 : 4;
        const int64_t per_wg_b = 256 * (int64_t)ppt;
        const size_t lds_b = brb_lds_bytes(num_branches, ppt);  // <= 56 KiB (twelve branches, four paths); 48 KiB at the driver's ten
        const int wgs_per_cu = (int)std::min<size_t>(3, ((size_t)160 << 10) / lds_b);
        const int ggrid = (int)std::max<int64_t>(1, std::min<int64_t>((P->n_paths + per_wg_b - 1) / per_wg_b, (int64_t)ctx->n_cus * wgs_per_cu));
        typedef void (*BinnedKernel)(BranchArgs, int64_t, int, double2*, int, int, int);
        static const BinnedKernel kern[3][2] = {{k_branch_date_binned<1, 3>, k_branch_date_binned<1, 4>},
                                                {k_branch_date_binned<2, 3>, k_branch_date_binned<2, 4>},
                                                {k_branch_date_binned<3, 3>, k_branch_date_binned<3, 4>}};
        const BinnedKernel k = kern[quads - 1][ppt - 3];
        if (lds_b > ((size_t)48 << 10)) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        const int64_t per_launch_b = (int64_t)ggrid * per_wg_b;
        const int64_t n_launches_b = std::max<int64_t>(1, (P->n_paths + per_launch_b - 1) / per_launch_b);
        state_bytes = (size_t)P->n_paths * sizeof(double2);
        rc = pool_alloc(ctx, state_bytes, &state);
        if (rc) {
            pool_release(ctx, Fbuf, P->bytes);
            return rc;
        }
        {
            TimedLaunch t(ctx, MCG_K_BRANCHING, (int64_t)ex.size() * n_launches_b + 1);
            for (int e = 0; e < (int)ex.size(); ++e) {
                const int t_idx = ex[(size_t)e];
                const bool branch = t_idx < ex_last && t_idx + 1 < n_cols;
                const int ns = branch ? b_slices : 0;
                for (int64_t l = 0; l < n_launches_b; ++l) {
                    // (the last generation's grid covers the paths that are left: a workgroup without a live path would still
                    //  draw, sort and gather for the clamped ones)
                    const int64_t left = P->n_paths - l * per_launch_b;
                    const int g = (int)std::min<int64_t>(ggrid, (left + per_wg_b - 1) / per_wg_b);
                    hipLaunchKernelGGL(k, dim3(g), dim3(256), lds_b, ctx->stream, a, l * per_launch_b, e, (double2*)state, e == 0, bshift, ns);
                }
            }
            hipLaunchKernelGGL(k_branch_finish, dim3(grid), dim3(256), 0, ctx->stream, (const double2*)state, P->n_paths, ctx->partials);
        }
        n_partials = grid;
    } else if (n_slices > 1 && n_slices <= BR_DATE_MAX_SLICES && quads >= 1 && quads <= 3 && !ex.empty()) {
        // Rows of two to four slices: one launch per exercise date, bounds in `state`.  A/B on one board, 1M x 50: 5.55 ms
        // (one launch, no slices) -> 5.25 (one launch, slices: 42 % L2 hits) -> 2.83 (per-date launches).  A pass costs
        // ~40 cycles per wave-load however few lanes take part, so rows of many slices lose what the hits gain (4M x 50,
        // 16 slices: 44.7 ms per date-launches against 29.6 in one launch with 8 slices and 35.3 without): those keep the
        // one-launch kernel.
        state_bytes = (size_t)P->n_paths * sizeof(double2);
        rc = pool_alloc(ctx, state_bytes, &state);
        if (rc) {
            pool_release(ctx, Fbuf, P->bytes);
            return rc;
        }
        {
            TimedLaunch t(ctx, MCG_K_BRANCHING, (int64_t)ex.size() * n_launches + 1);
            for (int e = 0; e < (int)ex.size(); ++e) {
                const int t_idx = ex[(size_t)e];
                const bool branch = t_idx < ex_last && t_idx + 1 < n_cols;
                for (int64_t l = 0; l < n_launches; ++l) {
                    const int64_t left = P->n_paths - l * per_launch;
                    const dim3 g((unsigned)std::min<int64_t>(bgrid, (left + per_wg - 1) / per_wg)), b(256);
                    const int ns = branch ? n_slices : 0;
                    if (quads == 1) hipLaunchKernelGGL(k_branch_date<1>, g, b, 0, ctx->stream, a, l * per_launch, e, (double2*)state, e == 0, slice_shift, ns);
                    else if (quads == 2) hipLaunchKernelGGL(k_branch_date<2>, g, b, 0, ctx->stream, a, l * per_launch, e, (double2*)state, e == 0, slice_shift, ns);
                    else hipLaunchKernelGGL(k_branch_date<3>, g, b, 0, ctx->stream, a, l * per_launch, e, (double2*)state, e == 0, slice_shift, ns);
                }
            }
            hipLaunchKernelGGL(k_branch_finish, dim3(grid), dim3(256), 0, ctx->stream, (const double2*)state, P->n_paths, ctx->partials);
        }
        n_partials = grid;
    } else {
        for (int64_t l = 0; l < n_launches; ++l) {
            TimedLaunch t(ctx, MCG_K_BRANCHING);
            double* part = ctx->partials + 2 * l * bgrid;
            const dim3 g(bgrid), b(256);
            if (quads <= 1) hipLaunchKernelGGL(k_branch_bounds<1>, g, b, 0, ctx->stream, a, l * per_launch, part, slice_shift, n_slices);
            else if (quads == 2) hipLaunchKernelGGL(k_branch_bounds<2>, g, b, 0, ctx->stream, a, l * per_launch, part, slice_shift, n_slices);
            else if (quads == 3) hipLaunchKernelGGL(k_branch_bounds<3>, g, b, 0, ctx->stream, a, l * per_launch, part, slice_shift, n_slices);
            else hipLaunchKernelGGL(k_branch_bounds_any, g, b, 0, ctx->stream, a, l * per_launch, part);
        }
    }
    if (hipGetLastError() != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        pool_release(ctx, Fbuf, P->bytes);
        if (state) pool_release(ctx, state, state_bytes);
        return fail(MCG_ERR_HIP, "BranchingProcesses: kernel launch failed");
    }
    double s[3];
    rc = finish_sums(ctx, n_partials, P->n_paths, s);  // {sum lower, sum upper, N}; synchronises the stream
    pool_release(ctx, Fbuf, P->bytes);
    if (state) pool_release(ctx, state, state_bytes);
    if (rc) return rc;
    if (!(s[2] >= 1.0)) return fail(MCG_ERR_EMPTY_PATHS, "BranchingProcesses: Empty pricePaths.");
    const double lo = s[0] / s[2], up = s[1] / s[2];
    if (lower) *lower = lo;
    if (upper) *upper = up;
    *price = 0.5 * (lo + up);  // :37
    return MCG_OK;
}

}  // namespace mcg
