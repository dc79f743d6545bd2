// Longstaff-Schwartz backward sweep on a device-resident step-major path matrix (gfx950).
//
// Follows /root/reference/src/models/LSMPricer.cpp:19-102 (the value-iteration variant: the FITTED
// continuation value is carried backwards, :85) with three MI355X-first changes that leave the
// fitted values -- and therefore every V[i][j] -- unchanged up to rounding:
//   * only one value vector V (n_paths doubles) is live instead of Values[N][M] (:35);
//   * the date's prices and V stay on chip wherever they fit: the whole sweep is ONE launch of co-operating workgroups
//     up to 8.37M paths per GPU (k_lsm_coop / k_lsm_big: 8 or 16 B per path and date), else one streaming kernel per
//     exercise date (k_lsm_date: reads S_j, S_{j-1}, V; writes V: 32 B per path and date);
//   * the least-squares fit (:61-76, Eigen bdcSvd on raw monomials) is solved from the (p+1)x(p+1) moment matrix of the
//     SCALED regressor x = S/K - 1 (same polynomial space => same fitted values; cond drops from ~1e8 to ~1e2) where the
//     reference's solve is at full numerical rank, and re-fitted by the reference's own rank rule from the moments about
//     the mean regressor where it is not (lsm_device.hpp).  The moments are the only cross-GPU exchange: 3p+2 doubles
//     per date, inside the one launch (node mailbox) or between two launches (ctx->allreduce).
// HBM-bound streaming; no MFMA (the "GEMM" A^T A is a (p+1)^2 moment accumulation, done in
// registers with wavefront-shuffle reductions).
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <shared_mutex>
#include <type_traits>

#include "lsm_device.hpp"
#include "mcg_internal.hpp"

namespace mcg {

// Agent-scope (sc1) accesses: through to the device's coherence point one by one, no fence (see the one-launch sweeps below).
__device__ __forceinline__ void lsm_st_shared(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double lsm_ld_shared(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every double that travels between workgroups (or GPUs) announces itself: its slot holds a reserved NaN until the value
// arrives, and whoever consumes it puts the NaN back -- a value that has not arrived cannot be mistaken for one.
constexpr unsigned LSM_SENTINEL32 = 0xFFF85EA7u;  // both halves of the reserved NaN (hipMemsetD32 fills the buffers)
__device__ __forceinline__ bool lsm_is_sentinel(double v) {
    return (unsigned long long)__double_as_longlong(v) == (((unsigned long long)LSM_SENTINEL32 << 32) | LSM_SENTINEL32);
}
__device__ __forceinline__ double lsm_sentinel() {
    return __longlong_as_double((long long)((((unsigned long long)LSM_SENTINEL32) << 32) | LSM_SENTINEL32));
}

// ------------------------------------------------------------------------------------------------
// The per-date route: ONE kernel per exercise date and -- sharded -- ONE collective between two of them.
// (Sharded runs through RCCL or a callback, orders > 4, > 8.37M paths per GPU, the prices after a hand-shake time-out.)
//   head : every workgroup solves the date's regression from the 3p+2 moments in `msg` (~1 us on one thread; sharded,
//          `msg` has just been summed over the ranks, so every workgroup of every GPU obtains the same coefficients);
//   body : V <- update with date j's fit (LSMPricer.cpp:78-94), moments of date j-1 accumulated from S_{j-1} and the
//          new V (:51-74): reads S_j, S_{j-1}, V, writes V = 32 B per path and date;
//   tail : per-workgroup partial moments leave by write-through stores; the LAST workgroup to finish (a relaxed
//          ticket -- no fence: a fence would write the XCD's L2 back, V included) sums them in a fixed order into `msg`.
// The host queues launch, all-reduce, launch, all-reduce, ... without ever looking at a result; WHICH date a launch
// works on is decided on the device: a three-word state {date j, phase, centre} that every workgroup reads in its
// head and the last workgroup advances in its tail.  That is what lets a date the first solve does not trust
// (lsm_solve_nb's refinement request: few or nearly coincident in-the-money prices, orders >= 4) be re-fitted by the
// reference's rank rule like in every other execution shape, although the re-fit needs a second, data-dependent
// reduction: such a date simply takes TWO launches -- the first one only accumulates the moments of row j about the
// mean regressor (V untouched: 16 B per path), the all-reduce that follows sums those instead, the second one solves
// them with lsm_solve_centered and carries on.  Every rank takes the same decision from the same summed moments, so the
// ranks stay in step.  Launches beyond the last date return at once; the host queues a few spare ones and, should a
// sweep need more than it queued (it reads the state back once per batch), queues the rest.
constexpr int LSM_ST_J = 0, LSM_ST_PHASE = 1, LSM_ST_MU = 2, LSM_ST_FAULT = 3;  // state words (doubles)
constexpr int LSM_ST_WORDS = 4;
static_assert(SC_LSM_STATE + LSM_ST_WORDS <= SC_LSM_TICKET, "the state fits its place in the ctx's scalar workspace");
enum { LSM_PH_REGULAR = 0, LSM_PH_REFINED = 1, LSM_PH_INIT = 2 };

struct LsmDateArgs {
    const double* data;   // step-major matrix
    int64_t ld, n;
    double* V;
    double K, invK, disc, dt, maturity;
    int is_call;
    double* msg;          // in: the moments the state's phase says (summed over the ranks); out: those of the next launch
    double* state;        // {date j the next launch works on (< 0: done), phase, centre of a refinement}
    double* partials;     // [NM][gridDim.x] (moment-major: the reducing workgroup reads contiguously), then [NM][groups]:
    //                       every slot holds the reserved NaN except between its store and its consumption (lsm_date_tail)
    unsigned* ticket;     // workgroups done, per group of 64 and of the groups; the last ones reset them (lsm_date_tail)
    unsigned spin_limit;  // polls before a consumer gives a slot up (state[LSM_ST_FAULT], the sweep ends, run_lsm fails)
    // test hooks (mcg_debug_lsm_date_fault): at date hook_date workgroup hook_wg withholds its partial moments (mode 1: a
    // store that never lands) or sends them ~hook_delay x 4 us AFTER its ticket (mode 2: a store that lands late)
    int hook_mode, hook_date, hook_wg, hook_delay;
};

// A consumer's read of one slot: the value, once it is there (normally at the first look: the ticket that elected this
// workgroup was drawn after the producers' stores had been acknowledged); the slot is re-armed for the next launch.
__device__ __forceinline__ double lsm_consume_slot(double* p, unsigned limit, bool& fault) {
    double v = lsm_ld_shared(p);
    unsigned spins = 0;
    while (lsm_is_sentinel(v)) {
        if (++spins > limit) {
            fault = true;
            return 0.0;
        }
        __builtin_amdgcn_s_sleep(2);
        v = lsm_ld_shared(p);
    }
    lsm_st_shared(p, lsm_sentinel());
    return v;
}

// Tail of a launch.  HAVE: this launch produced partial moments m.  Returns true in the last workgroup to get here,
// after it has left the sum of all partials in msg (or zeros when there were none).  Two levels, so that neither a
// thousand workgroups finishing together queue up at ONE ticket nor one workgroup reads a thousand partials: workgroups
// form groups of LSM_DATE_GROUP (by blockIdx; the smaller the group, the fewer workgroups meet at one ticket within the
// same microsecond); the last of a group sums the group's partials (lane l the l-th member, then the wavefront
// butterfly) and draws the top ticket; the last of those sums the group sums.  The summation order is
// fixed by blockIdx alone, whatever the order of arrival.
#ifndef LSM_DATE_GROUP_N
#define LSM_DATE_GROUP_N 16  // A/B on one box, C5 shard, sweep span per pass: 64 -> 12.08 ms, 32 -> 11.14, 16 -> 10.88
#endif
constexpr int LSM_DATE_GROUP = LSM_DATE_GROUP_N;
constexpr int LSM_DATE_MAX_GROUPS = 2048 / LSM_DATE_GROUP;  // tickets: [0 .. MAX_GROUPS) the groups', [MAX_GROUPS] the top one
static_assert((LSM_DATE_MAX_GROUPS + 1 + 1) / 2 <= SCALARS_DOUBLES - SC_LSM_TICKET, "the tickets fit the ctx's scalar workspace");

template <int NM>
__device__ __forceinline__ bool lsm_date_tail(const LsmDateArgs& a, bool have, double (&m)[NM], double* red, unsigned* sm_last, int date) {
    const unsigned G = gridDim.x, n_groups = (G + LSM_DATE_GROUP - 1) / LSM_DATE_GROUP;
    const unsigned grp = blockIdx.x / LSM_DATE_GROUP, first = grp * LSM_DATE_GROUP;
    const unsigned members = min((unsigned)LSM_DATE_GROUP, G - first);
    double* gsum = a.partials + (int64_t)NM * G;  // [NM][n_groups]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    static_assert(LSM_DATE_GROUP <= 64, "one member per lane");
    if (have) block_sum<NM, 4>(m, red);
    if (threadIdx.x == 0) {
        const bool hooked = a.hook_mode != 0 && date == a.hook_date && (int)blockIdx.x == a.hook_wg;
        auto send = [&]() {
#pragma unroll
            for (int q = 0; q < NM; ++q) lsm_st_shared(a.partials + (int64_t)q * G + blockIdx.x, m[q]);
            __builtin_amdgcn_s_waitcnt(0);  // the write-through stores are acknowledged before this workgroup's ticket is drawn
        };
        if (have && !hooked) send();
        *sm_last = __hip_atomic_fetch_add(a.ticket + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1 ? 1u : 0u;
        if (have && hooked && a.hook_mode == 2) {  // (test: these partials land after the ticket)
            for (int d = 0; d < a.hook_delay; ++d) __builtin_amdgcn_s_sleep(127);
            send();
        }
    }
    __syncthreads();
    if (*sm_last == 0) return false;
    // last of its group: every member has drawn its ticket (and, long before, read msg and the state).  Its partials are
    // CONSUMED, not just read: a slot that still holds the reserved NaN has not arrived and is waited for (bounded).
    bool fault = false;
    if (have) {
        for (int q = wave; q < NM; q += 4) {
            double s = (unsigned)lane < members ? lsm_consume_slot(a.partials + (int64_t)q * G + first + lane, a.spin_limit, fault) : 0.0;
            s = wave_sum(s);
            if (lane == 0) lsm_st_shared(gsum + (int64_t)q * n_groups + grp, s);
        }
        if (__builtin_amdgcn_ballot_w64(fault) != 0ull && lane == 0) lsm_st_shared(a.state + LSM_ST_FAULT, 1.0);
        __builtin_amdgcn_s_waitcnt(0);  // every wave's group sums (and a fault) are acknowledged before the barrier lets thread 0 draw the top ticket
    }
    __syncthreads();  // (sm_last is rewritten below)
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.ticket + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        *sm_last = __hip_atomic_fetch_add(a.ticket + LSM_DATE_MAX_GROUPS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_groups - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (*sm_last == 0) return false;
    fault = false;
    for (int q = wave; q < NM; q += 4) {
        double s = 0.0;
        if (have) {
            for (unsigned l = lane; l < n_groups; l += 64) s += lsm_consume_slot(gsum + (int64_t)q * n_groups + l, a.spin_limit, fault);
            s = wave_sum(s);
        }
        if (lane == 0) a.msg[q] = s;
    }
    if (__builtin_amdgcn_ballot_w64(fault) != 0ull && lane == 0) lsm_st_shared(a.state + LSM_ST_FAULT, 1.0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();  // every wave's verdict is in before thread 0 looks at the flag (lsm_date_advance)
    if (threadIdx.x == 0) __hip_atomic_store(a.ticket + LSM_DATE_MAX_GROUPS, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// The last workgroup of a launch moves the state on -- or ends the sweep when some consumer gave a slot up (the sums in
// msg are then incomplete: run_lsm reads the flag with the state and fails with MCG_ERR_HIP; never a silently wrong price).
__device__ __forceinline__ void lsm_date_advance(const LsmDateArgs& a, int next_j, int next_phase, double mu) {
    if (lsm_ld_shared(a.state + LSM_ST_FAULT) != 0.0) {
        a.state[LSM_ST_J] = -1.0;
        return;
    }
    a.state[LSM_ST_J] = (double)next_j;
    a.state[LSM_ST_PHASE] = (double)next_phase;
    a.state[LSM_ST_MU] = mu;
}

// NB = poly_order + 1 basis functions; NM = (2p+1) power sums + (p+1) cross sums = 3*NB - 1.
// The grid is one resident wave of workgroups (run_lsm sizes it by the occupancy query).  A thread works on UNITS of two
// adjacent paths (16-byte loads and stores; rows and V are 16-byte aligned, the path count is padded to a unit) and walks
// its grid-stride units through a ring of LSM_DATE_DEPTH register slots: the loads of the next DEPTH units are always in
// flight, and the first DEPTH are issued BEFORE the head's solve, so HBM stays busy while thread 0 of every workgroup
// solves (a quarter of the launch's bytes are on their way by the time the coefficients exist).
#ifndef LSM_DATE_DEPTH_N
#define LSM_DATE_DEPTH_N 3
#endif
constexpr int LSM_DATE_DEPTH = LSM_DATE_DEPTH_N;

template <int NB>
__global__ __launch_bounds__(256) void k_lsm_date(LsmDateArgs a) {
    constexpr int NM = 3 * NB - 1;
    constexpr int D = LSM_DATE_DEPTH;
    __shared__ double red[NM * 4];
    __shared__ double sm_mom[48];
    __shared__ double sm_coef[LSM_COEF_STRIDE];
    __shared__ double sm_ws[lsm_ws_doubles(NB)];
    __shared__ unsigned sm_last;
    const int j = (int)a.state[LSM_ST_J];
    if (j < 0) return;  // the sweep is over: a spare launch
    const int phase = (int)a.state[LSM_ST_PHASE];
    const bool call = a.is_call != 0;
    const bool reg = phase != LSM_PH_INIT && !(j * a.dt > a.maturity);          // LSMPricer.cpp:43-44
    const bool have = j >= 1 && !((j - 1) * a.dt > a.maturity);                 // date j-1 regresses: its inputs are formed here (:51-74)
    const bool need_v = phase != LSM_PH_INIT;
    const double2* S_j = reinterpret_cast<const double2*>(a.data + (int64_t)j * a.ld);
    const double2* S_mom = reinterpret_cast<const double2*>(a.data + (int64_t)(have ? j - 1 : j) * a.ld);
    double2* V2 = reinterpret_cast<double2*>(a.V);
    const int64_t n_units = (a.n + 1) / 2;
    const int64_t chunk = (int64_t)gridDim.x * 256;
    const int64_t n_chunks = (n_units + chunk - 1) / chunk;
    const bool rev = (j & 1) != 0;  // consecutive dates walk the paths in opposite directions (see run_lsm)
    const int64_t lane_unit = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto unit_of = [&](int64_t k) { return k < n_chunks ? (rev ? n_chunks - 1 - k : k) * chunk + lane_unit : n_units; };
    int64_t u[D];
    double2 s[D], v[D], sm[D];
    auto fetch = [&](int d, int64_t k, bool with_mom) {
        u[d] = unit_of(k);
        s[d] = make_double2(0.0, 0.0);
        v[d] = make_double2(0.0, 0.0);
        sm[d] = make_double2(0.0, 0.0);
        if (u[d] < n_units) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(S_j + u[d]));  // row j is not needed again (-1.3 %)
            s[d] = make_double2(t.x, t.y);
            if (need_v) v[d] = V2[u[d]];  // (V and row j-1 come back at the next launch: ordinary loads and stores, the
            //                               memory-side cache keeps part of them; nontemporal there costs 2 %)
            if (with_mom) sm[d] = S_mom[u[d]];
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, d, have);  // (row j-1 is not needed by a refinement launch: wasted loads on those rare ones)
    double c[NB];
    double n_itm = 0.0, center = 0.0;
    double m[NM];
#pragma unroll
    for (int q = 0; q < NM; ++q) m[q] = 0.0;
    if (reg) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int t = 0; t < NM; ++t) sm_mom[t] = a.msg[t];
            if (phase == LSM_PH_REFINED) {
                lsm_solve_centered(sm_mom, NB, a.state[LSM_ST_MU], a.K, sm_coef, sm_ws);
            } else if constexpr (NB <= 9) {
                lsm_solve_nb<NB>(sm_mom, 1.0, a.K, sm_coef);
            } else {
                // Orders >= 9: raw monomials up to S^9 and beyond are truncated by the reference's rank rule on every
                // date, so there is no fast path to try -- every date with an in-the-money path is re-fitted.
                for (int t = 0; t < LSM_COEF_DOUBLES; ++t) sm_coef[t] = 0.0;
                sm_coef[LSM_C_COUNT] = sm_mom[0];
                if (sm_mom[0] > 0.0) {
                    sm_coef[LSM_C_REFINE] = 1.0;
                    sm_coef[LSM_C_HINT] = sm_mom[1] / sm_mom[0];
                }
            }
        }
        __syncthreads();
        if (phase == LSM_PH_REGULAR && sm_coef[LSM_C_REFINE] != 0.0) {  // (uniform over the grid and over the ranks)
            // The date is re-fitted about the mean of its regressor: this launch only forms those moments.
            const double mu = sm_coef[LSM_C_HINT];
            for (int64_t k0 = 0; k0 < n_chunks; k0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (u[d] < n_units) {
                        lsm_accumulate_centered<NB>(m, payoff_of(call, s[d].x, a.K) > 1e-14, s[d].x, v[d].x, a.invK, mu, a.disc);
                        lsm_accumulate_centered<NB>(m, 2 * u[d] + 1 < a.n && payoff_of(call, s[d].y, a.K) > 1e-14, s[d].y, v[d].y,
                                                    a.invK, mu, a.disc);
                    }
                    fetch(d, k0 + d + D, false);
                }
            }
            if (lsm_date_tail<NM>(a, true, m, red, &sm_last, j) && threadIdx.x == 0) lsm_date_advance(a, j, LSM_PH_REFINED, mu);
            return;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) c[q] = sm_coef[q];
        n_itm = sm_coef[LSM_C_COUNT];
        center = sm_coef[LSM_C_CENTER];
    }
    auto update = [&](double s_now, double v_old) {
        if (phase == LSM_PH_INIT) return payoff_of(call, s_now, a.K);  // LSMPricer.cpp:37-40
        if (!reg) return v_old * a.disc;                                // :43-49
        const double pay = payoff_of(call, s_now, a.K);
        if (pay > 1e-14 && n_itm > 0.0)                                 // :78-86
            return fmax(pay, lsm_continuation<NB>(c, center, fma(s_now, a.invK, -1.0)));
        if (pay < 1e-14) return v_old * a.disc;                         // :89-94
        return 0.0;  // payoff == 1e-14 exactly falls through both branches (:55 vs :91)
    };
    auto accumulate = [&](bool live, double s_prev, double v_new) {  // regression inputs of date j-1
        if (live && payoff_of(call, s_prev, a.K) > 1e-14) {
            const double x = fma(s_prev, a.invK, -1.0);
            const double y = v_new * a.disc;
            double pw = 1.0;
#pragma unroll
            for (int q = 0; q < 2 * NB - 1; ++q) {
                m[q] += pw;
                if (q < NB) m[2 * NB - 1 + q] = fma(pw, y, m[2 * NB - 1 + q]);
                pw *= x;
            }
        }
    };
    for (int64_t k0 = 0; k0 < n_chunks; k0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (u[d] < n_units) {
                // (the second path of the last unit may lie beyond the shard: computed from whatever the padding holds,
                // stored into V's slack, never counted)
                const double2 vn = make_double2(update(s[d].x, v[d].x), update(s[d].y, v[d].y));
                V2[u[d]] = vn;
                if (have) {
                    accumulate(true, sm[d].x, vn.x);
                    accumulate(2 * u[d] + 1 < a.n, sm[d].y, vn.y);
                }
            }
            fetch(d, k0 + d + D, have);
        }
    }
    if (lsm_date_tail<NM>(a, have, m, red, &sm_last, j) && threadIdx.x == 0) lsm_date_advance(a, j - 1, LSM_PH_REGULAR, 0.0);
}

// What the solve needs to know about MartingaleOptimization's refit (its samples are not a row of the matrix, its driver
// is on the host anyway):
//   request_only: solve the first pass with the refinement test on and leave the request in the coefficient block;
//   centered:     the moments are already about `mu`: solve them with lsm_solve_centered.
struct LsmRefine {
    double K;
    int request_only, centered;
    double mu;
};

// One block.  do_reduce: partials[nm][n_blocks] -> moments[nm] in a fixed order (wave w sums moments
// w, w+4, ...: lanes stride over the blocks, then a wavefront butterfly).  do_solve: moments -> coef.
// Single GPU: both in one launch.  Sharded: reduce, all-reduce of `moments`, then solve.
__global__ __launch_bounds__(256) void k_lsm_reduce_solve(const double* partials, int n_blocks, int nm, int nb,
                                                          double* moments, double* coef, int do_reduce, int do_solve,
                                                          double min_count, LsmRefine rf) {
    __shared__ double sm[48];
    __shared__ double sm_c[LSM_COEF_STRIDE];
    __shared__ double sm_ws[lsm_ws_doubles(LSM_MAX_NB)];
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
    if (!do_solve) return;
    if (threadIdx.x == 0) {
        const double* mc = do_reduce ? sm : moments;
        if (rf.centered) {
            if (mc[0] >= min_count) {
                lsm_solve_centered(mc, nb, rf.mu, rf.K, sm_c, sm_ws);
            } else {  // too few samples: coefficients stay 0 (as in lsm_solve_nb)
                for (int t = 0; t < LSM_COEF_DOUBLES; ++t) sm_c[t] = 0.0;
                sm_c[LSM_C_COUNT] = mc[0];
            }
        } else if (nb <= 9) {
            lsm_solve_one(mc, nb, min_count, rf.request_only ? rf.K : 0.0, sm_c);
            if (!rf.request_only) sm_c[LSM_C_REFINE] = 0.0;
        } else {  // orders >= 9: the reference's rank rule truncates the raw monomials anyway -- always the centred re-fit
            for (int t = 0; t < LSM_COEF_DOUBLES; ++t) sm_c[t] = 0.0;
            sm_c[LSM_C_COUNT] = mc[0];
            if (rf.request_only && mc[0] >= min_count && mc[0] > 0.0) {
                sm_c[LSM_C_REFINE] = 1.0;
                sm_c[LSM_C_HINT] = mc[1] / mc[0];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < LSM_COEF_DOUBLES) coef[threadIdx.x] = sm_c[threadIdx.x];
}

template <int NB, int PPT>
__global__ __launch_bounds__(256) void k_lsm_small(const double* data, int64_t ld, int n, int n_cols, double r_unused,
                                                   double K, double maturity, double dt, double disc, int is_call,
                                                   double* out3) {
    (void)r_unused;
    lsm_small_body<NB, PPT>(data, ld, n, n_cols, K, maturity, dt, disc, is_call, out3);
}

// At most 256 paths (the production shape is 250): a single wavefront, no workgroup barrier on the per-date path.
template <int NB>
__global__ __launch_bounds__(64) void k_lsm_small_wave(const double* data, int64_t ld, int n, int n_cols, double K,
                                                       double maturity, double dt, double disc, int is_call, double* out3) {
    __shared__ double ws[lsm_ws_doubles(NB) + LSM_COEF_DOUBLES];
    double sum_v, sum_v2;
    lsm_wave_body<NB>(data, ld, n, n_cols, K, maturity, dt, disc, is_call, ws, sum_v, sum_v2);
    if (threadIdx.x == 0) {
        out3[0] = sum_v;
        out3[1] = sum_v2;
        out3[2] = (double)n;
    }
}

template <int NB>
static void launch_small_nb(mcg_ctx* ctx, const mcg_paths* P, double K, double maturity, double dt, double disc,
                            int is_call, double* out3) {
    if (P->n_paths <= 256)
        hipLaunchKernelGGL(k_lsm_small_wave<NB>, dim3(1), dim3(64), 0, ctx->stream, P->data, P->ld, (int)P->n_paths,
                           P->n_steps + 1, K, maturity, dt, disc, is_call, out3);
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

// ------------------------------------------------------------------------------------------------
// The whole sweep in ONE launch of co-resident, co-operating workgroups (single GPU, poly_order <= 4).  Every thread keeps the values V of
// its PPT paths in registers for the entire sweep, so the value vector never touches HBM, and -- when they fit
// (KEEP) -- also the date's prices between the regression pass and the update pass, so that the path matrix is read
// exactly once: 8 B per path and date instead of the 32 B of the per-date kernels (16 B without KEEP).
//
// Per regression date the workgroups exchange 3p+2 moments one way and p+2 coefficients the other way.  On MI355X
// (8 XCDs, one L2 each) the textbook device-wide barrier -- an atomic counter between two agent-scope fences -- costs
// 14 us at 256 workgroups and 63 us at 1024, because every fence writes back and invalidates the XCD's L2 and those
// operations serialise inside the XCD (tools/ubench_gridsync.hip: 1.5-2.4 us without the fences).  The exchange here
// uses no fence, no read-modify-write and no flag: every exchanged double is written and read with agent-scope (sc1)
// stores and loads, which go through to the device coherence point one by one, and every slot announces itself --
// it holds a reserved NaN pattern until its value arrives:
//   every workgroup : partial moments -> partials[set][t][b]   (set = 0, 1 alternating; 2 for a refinement round)
//   workgroup 0     : each lane polls its own slots (all loads of a round in flight together) until none is the
//                     sentinel; fixed-order reduction; solve; coefficients -> coef[set][0..12]
//   every workgroup : lanes 0..9 poll one coefficient each
// Two one-way trips per date.  Slots are recycled two dates later: workgroup 0 puts the sentinel back into the
// partials right after reading them and into the earlier coefficient blocks once the NEXT date's partials have all arrived (every
// workgroup has then used those coefficients); it waits for the acknowledgement of these stores (s_waitcnt) before
// it publishes anything newer.  The path matrix is read-only and V never leaves the registers, so nothing else
// needs coherence.  The grid is sized by the occupancy query (minus a margin) so that all workgroups are co-resident,
// one such kernel runs at a time per process, and every spin is bounded and raises a flag instead of hanging.
// Sharded runs keep the per-date kernels: their all-reduce is issued from the host between two launches.
constexpr int LSM_AREA_REFINE = 2;                // slot set of refinement rounds (regular rounds alternate between sets 0 and 1)
constexpr int LSM_COOP_RETRY_AFTER = 8;           // prices through the per-date kernels after a time-out before the one-launch sweep is tried again
constexpr int LSM_MAX_DEVICES = 64;                // per-device locks of the one-launch sweeps
constexpr int LSM_COOP_MAX_GRID = 512;            // 8 slots per lane and moment in workgroup 0, two moments in flight

struct LsmCoopArgs {
    const double* data;
    int64_t ld, n;
    int n_cols;
    double K, invK, maturity, dt, disc;
    int is_call;
    double* partials;  // [3][NM][gridDim.x], sentinel-filled: slot sets 0 / 1 of the regular rounds, 2 of refinement rounds
    double* coef;      // [3][LSM_COEF_STRIDE], sentinel-filled: the coefficient blocks (lsm_device.hpp: LSM_C_*)
    unsigned* timeout; // set when a spin gives up
    unsigned spin_limit; // polling rounds before a spin gives up (LSM_SPIN_LIMIT; mcg_debug_lsm_hooks in tests)
    unsigned poll_delay; // test hook: see lsm_poll_coefficients
    int u_full;          // k_lsm_big: units 0 .. u_full-1 lie inside the shard for EVERY thread (no masking needed)
    double* out;       // [2 * gridDim.x]: per-block {sum V, sum V^2}
    // node-level exchange (mcg_comm_init_shm): mailbox[round][rank][SHM_ROW_DOUBLES], nullptr on a single GPU.
    //   host mailbox (mb_push = 0): ONE copy in device-mapped host memory; every rank writes its row there and polls
    //     the others' (each access crosses PCIe);
    //   peer-memory mailbox (mb_push = 1): one copy per GPU in its own HBM; mbox = this rank's, mb_peer[r] = rank r's
    //     as mapped here (hipIpcOpenMemHandle: peer memory over xGMI); a rank pushes its row into every copy and polls
    //     only its own.
    double* mbox;
    int mb_ranks, mb_rank, mb_push;
    double* mb_peer[SHM_MAX_RANKS];
};

// -DMCG_LSM_TRACE (timing studies only): workgroup 0 stamps the phases of the first 32 exchanges of k_lsm_coop with the
// 100 MHz wall clock; run_lsm_coop prints the differences to stderr.
#ifdef MCG_LSM_TRACE
__device__ unsigned long long g_lsm_trace[32 * 8];
#define LSM_TRACE(round_, slot_)                                                             \
    do {                                                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (round_) < 32) g_lsm_trace[(round_) * 8 + (slot_)] = wall_clock64(); \
    } while (0)
#else
#define LSM_TRACE(round_, slot_) do { } while (0)
#endif

constexpr unsigned LSM_SPIN_LIMIT = 1u << 20;  // rounds of ~1.5 us; a co-resident grid needs a handful
constexpr unsigned LSM_DATE_SPIN_LIMIT = 1u << 16;  // k_lsm_date's consumers: ~0.2 s for a store that was acknowledged before the ticket

// The per-date exchange of the one-launch sweeps (protocol: see the comment above), in its three parts.
// G = number of workgroups that contribute partial moments, b = this workgroup's slot among them.

// every contributing workgroup: block-reduce the per-thread moments and send them off
template <int NB>
__device__ __forceinline__ void lsm_publish_partials(const LsmCoopArgs& a, double (&m)[3 * NB - 1], unsigned G, unsigned b, int area,
                                                     double* red) {
    constexpr int NM = 3 * NB - 1;
    block_sum<NM, 4>(m, red);
    double* part = a.partials + (int64_t)area * NM * G;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < NM; ++t) lsm_st_shared(part + (int64_t)t * G + b, m[t]);
    }
}
// ... only the moments T0 .. T0 + NT - 1 of the set (k_lsm_coop sends the power sums of a date, which do not depend on V, a
// round ahead of its cross sums); red: NT * 4 doubles of its own
template <int NB, int T0, int NT>
__device__ __forceinline__ void lsm_publish_some(const LsmCoopArgs& a, double (&v)[NT], unsigned G, unsigned b, int area, double* red) {
    constexpr int NM = 3 * NB - 1;
    static_assert(T0 >= 0 && T0 + NT <= NM, "a range of the set's moments");
    block_sum<NT, 4>(v, red);
    double* part = a.partials + (int64_t)area * NM * G;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) lsm_st_shared(part + (int64_t)(T0 + t) * G + b, v[t]);
    }
}

// the reducing workgroup: poll all G partials, fixed-order sum, solve, publish the coefficients (and leave them in
// sm_coef for its own threads).  The same fixed-order reduction as k_lsm_reduce_solve: wave w sums moments w, w+4, ...;
// lane l the workgroups l, l+64, ...  Two of the wave's moments per round: all their slots are polled together, so the
// usual date costs one round trip to the coherence point, not one per moment.
// System-scope access to the node mailbox (host memory mapped into every GPU of the node: each access crosses PCIe; or
// the GPUs' own HBM, the peers' copies mapped over xGMI).
__device__ __forceinline__ void lsm_st_node(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double lsm_ld_node(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// Sum the local moments sm_mom[0..NM) over the GPUs of the node, inside the kernel: lane t of wave 0 publishes moment t in
// this rank's row of the round's mailbox slot -- in the one shared copy, or (peer-memory mailbox) in every GPU's copy --
// and polls the same entry of every rank's row in the copy it reads (all loads of a poll in flight together: one PCIe
// round trip, or local HBM) until none holds the reserved NaN; the sum runs in rank order, so every GPU obtains the same
// bits.  Rows are written once per sweep; the host re-armed them before the launch (shm_arm_mailbox).
template <int NM>
__device__ __forceinline__ void lsm_node_allreduce(const LsmCoopArgs& a, int round, bool& gave_up, double* sm_mom) {
    static_assert(NM <= SHM_ROW_DOUBLES, "a rank's mailbox row holds the moments of orders <= 4");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // sm_mom is complete
    if (wave == 0 && lane < NM) {
        const size_t slot_off = (size_t)round * SHM_MAX_RANKS * SHM_ROW_DOUBLES;
        double* slot = a.mbox + slot_off;
        if (a.mb_push) {
#pragma unroll
            for (int r = 0; r < SHM_MAX_RANKS; ++r)
                if (r < a.mb_ranks) lsm_st_node(a.mb_peer[r] + slot_off + a.mb_rank * SHM_ROW_DOUBLES + lane, sm_mom[lane]);
        } else {
            lsm_st_node(slot + a.mb_rank * SHM_ROW_DOUBLES + lane, sm_mom[lane]);
        }
        double v[SHM_MAX_RANKS];
        unsigned spins = 0;
        bool missing = !gave_up;
        do {
#pragma unroll
            for (int r = 0; r < SHM_MAX_RANKS; ++r) v[r] = r < a.mb_ranks ? lsm_ld_node(slot + r * SHM_ROW_DOUBLES + lane) : 0.0;
            if (!missing) break;
            missing = false;
#pragma unroll
            for (int r = 0; r < SHM_MAX_RANKS; ++r) missing = missing || lsm_is_sentinel(v[r]);
            if (missing) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > a.spin_limit) {
                    __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = true;
                    break;
                }
            }
        } while (missing);
        double tot = 0.0;
#pragma unroll
        for (int r = 0; r < SHM_MAX_RANKS; ++r) tot += v[r];  // (entries beyond the communicator are +0)
        sm_mom[lane] = tot;
    }
    __syncthreads();
}

// centered: this round carries the moments of a refinement pass about `mu` (lsm_solve_nb asked for it on the previous
// round of the same date); ws = LDS workspace of lsm_solve_centered; round = exchange counter of this sweep (the slot
// of the node mailbox when the sweep is sharded over the GPUs of a node).
template <int NB>
__device__ __forceinline__ void lsm_reduce_solve_publish(const LsmCoopArgs& a, unsigned G, int area, bool& gave_up, double* sm_mom,
                                                         double* sm_coef, bool centered, double mu, double* ws, int round) {
    constexpr int NM = 3 * NB - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* part = a.partials + (int64_t)area * NM * G;
    double* coef_now = a.coef + LSM_COEF_STRIDE * area;
    for (int t0 = wave; t0 < NM; t0 += 8) {
        const int t1 = t0 + 4;
        const bool two = t1 < NM;
        double* slot0 = part + (int64_t)t0 * G;
        double* slot1 = part + (int64_t)(two ? t1 : t0) * G;
        constexpr int K = LSM_COOP_MAX_GRID / 64;
        double v0[K], v1[K];
        unsigned spins = 0;
        bool missing = !gave_up;  // after one timeout nothing is waited for any more: the run is void
        while (missing) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const unsigned b = lane + 64u * k;
                v0[k] = b < G ? lsm_ld_shared(slot0 + b) : 0.0;
                v1[k] = b < G ? lsm_ld_shared(slot1 + b) : 0.0;
            }
            missing = false;
#pragma unroll
            for (int k = 0; k < K; ++k) missing = missing || lsm_is_sentinel(v0[k]) || lsm_is_sentinel(v1[k]);
            if (missing) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > a.spin_limit) {
                    __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = true;
                    break;
                }
            }
        }
        double sum0 = 0.0, sum1 = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned b = lane + 64u * k;
            if (b < G) {
                sum0 += v0[k];
                sum1 += v1[k];
                lsm_st_shared(slot0 + b, lsm_sentinel());  // recycled two dates from now
                if (two) lsm_st_shared(slot1 + b, lsm_sentinel());
            }
        }
        sum0 = wave_sum(sum0);
        sum1 = wave_sum(sum1);
        if (lane == 0) {
            sm_mom[t0] = sum0;
            if (two) sm_mom[t1] = sum1;
        }
    }
    LSM_TRACE(round, 3);  // workgroup 0: this wave's moments are in
    if (a.mbox) lsm_node_allreduce<NM>(a, round, gave_up, sm_mom);  // (uniform) local -> node-wide moments
    __syncthreads();
    LSM_TRACE(round, 4);  // everybody's
    // ALL partials of a regular round are in -- behind the barrier: every wave's moments, the cross sums included, which
    // a workgroup sends only after it has read the previous coefficient block (k_lsm_coop sends a date's power sums a
    // round ahead, so the moments wave 0 polls prove nothing on their own).  Every workgroup is therefore past every
    // earlier coefficient block -- the previous regular round's and, if that date was re-fitted, the refinement
    // round's -- and they can be re-armed.  (A refinement round recycles nothing: the blocks it could recycle are the
    // ones the next regular round takes care of.)
    if (area != LSM_AREA_REFINE && threadIdx.x < 2 * LSM_COEF_DOUBLES) {
        const int blk = threadIdx.x < LSM_COEF_DOUBLES ? (area ^ 1) : LSM_AREA_REFINE;
        lsm_st_shared(a.coef + LSM_COEF_STRIDE * blk + threadIdx.x % LSM_COEF_DOUBLES, lsm_sentinel());
    }
    if (threadIdx.x == 0) {
        if (centered) lsm_solve_centered(sm_mom, NB, mu, a.K, sm_coef, ws);
        else lsm_solve_nb<NB>(sm_mom, 1.0, a.K, sm_coef);
    }
    LSM_TRACE(round, 5);  // solved
    __builtin_amdgcn_s_waitcnt(0);  // the recycling stores are acknowledged before anything newer goes out
    __syncthreads();
    if (threadIdx.x < LSM_COEF_DOUBLES) lsm_st_shared(coef_now + threadIdx.x, sm_coef[threadIdx.x]);
}

// every other workgroup: the first LSM_COEF_DOUBLES lanes poll one entry of the coefficient block each into sm_coef
__device__ __forceinline__ void lsm_poll_coefficients(const LsmCoopArgs& a, int area, bool& gave_up, double* sm_coef) {
    double* coef_now = a.coef + LSM_COEF_STRIDE * area;
    // test hook (mcg_debug_lsm_hooks): workgroups other than 0 arrive late at their poll, ~4 us per unit -- the
    // re-arming of a coefficient block must not depend on how quickly it was read
    for (unsigned d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(127);
    if (threadIdx.x < LSM_COEF_DOUBLES) {
        double cv = lsm_ld_shared(coef_now + threadIdx.x);
        unsigned spins = 0;
        while (lsm_is_sentinel(cv) && !gave_up) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > a.spin_limit) {
                __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
                break;
            }
            cv = lsm_ld_shared(coef_now + threadIdx.x);
        }
        sm_coef[threadIdx.x] = cv;
    }
    __syncthreads();
}

// k_lsm_coop: workgroup 0 contributes AND reduces.  after_publish() runs right after this workgroup's partial moments
// have left: the place to put loads in flight that should overlap the ~10 us the moments and coefficients travel.
template <int NB, class F>
__device__ __forceinline__ void lsm_exchange(const LsmCoopArgs& a, double (&m)[3 * NB - 1], int area, bool& gave_up,
                                             double* red, double* sm_mom, double* sm_coef, bool centered, double mu, double* ws,
                                             int round, F&& after_publish) {
    LSM_TRACE(round, 1);  // moments accumulated
    lsm_publish_partials<NB>(a, m, gridDim.x, blockIdx.x, area, red);
    after_publish();
    LSM_TRACE(round, 2);  // published, next row requested
    if (blockIdx.x == 0) lsm_reduce_solve_publish<NB>(a, gridDim.x, area, gave_up, sm_mom, sm_coef, centered, mu, ws, round);
    else lsm_poll_coefficients(a, area, gave_up, sm_coef);
}

// Branch-free per-path bodies of k_lsm_big (no exec-masked regions: the unrolled unit loop must stay one basic block so
// that the loads of the next pipeline stage can be scheduled across it).  (Tried in k_lsm_coop too: at one wave per SIMD
// and 16 paths per thread the branchy bodies are twice as fast there.)
// PayoffFunction (include/core/common.h:8-14) as max(sg s + nsK, 0) with (sg, nsK) = (1, -K) for a call and (-1, K) for a
// put: one FMA, same rounding as s - K / K - s.  A path that is not in the money enters the sums with weight w = 0 (one
// select on the high word of 1.0), i.e. adds exact zeros; in the money the products are the same x^t the other kernels
// form, bit for bit.
struct LsmPay {
    double sg, nsK;
};
__device__ __forceinline__ double lsm_pay(const LsmPay& p, double s) { return fmax(fma(p.sg, s, p.nsK), 0.0); }

template <int NB>
__device__ __forceinline__ void lsm_accumulate(double (&m)[3 * NB - 1], const LsmPay& p, double s, double v, double invK,
                                               double disc) {
    const double w = lsm_pay(p, s) > 1e-14 ? 1.0 : 0.0;  // regression inputs, LSMPricer.cpp:51-74
    const double x = fma(s, invK, -1.0);
    const double y = v * disc;
    double pw = w;
#pragma unroll
    for (int t = 0; t < 2 * NB - 1; ++t) {
        m[t] += pw;
        if (t < NB) m[2 * NB - 1 + t] = fma(pw, y, m[2 * NB - 1 + t]);
        if (t + 1 < 2 * NB - 1) pw *= x;
    }
}

// per lane: mask bit set ? a : b
// The mask comes straight from a v_cmp (ballot), i.e. from a VALU instruction that writes an SGPR pair; on gfx940+ a VALU
// instruction must not read such a pair within 2 wait states, and hipcc does not guard the operands of an asm statement
// (tools/check_asm_hazards.py).  tools/ubench_hazard.hip could not make this pattern fail on the hardware (2 10^8 selects
// back to back: none wrong), but the ISA asks for the wait states, so the statement carries them: both halves in ONE
// statement behind one `s_nop 1` (A/B on the C5 sweep: nothing measurable).
__device__ __forceinline__ double lsm_select(unsigned long long mask, double a, double b) {
    int lo, hi;
    asm("s_nop 1\n\tv_cndmask_b32_e64 %0, %2, %3, %6\n\tv_cndmask_b32_e64 %1, %4, %5, %6"
        : "=&v"(lo), "=v"(hi)   // (lo is written while the second instruction's inputs are still to be read; hi is written last)
        : "v"(__double2loint(b)), "v"(__double2loint(a)), "v"(__double2hiint(b)), "v"(__double2hiint(a)), "s"(mask));
    return __hiloint2double(hi, lo);
}

template <int NB>
__device__ __forceinline__ double lsm_update(const LsmPay& p, double s, double v_old, const double (&c)[NB], double center,
                                             unsigned long long any_itm /* all ones / zero: wave-uniform */,
                                             double invK, double disc) {  // LSMPricer.cpp:78-94
    const double pay = lsm_pay(p, s);
    const double cont = lsm_continuation<NB>(c, center, fma(s, invK, -1.0));
    // payoff == 1e-14 exactly falls through both of the reference's branches (:55 vs :91) and keeps 0
    // The two selects are written as v_cndmask on a ballot mask: from `c ? a : b` hipcc builds a divergent branch
    // around the polynomial here, i.e. four extra basic blocks and two exec-mask round trips per path.
    const double v = lsm_select(__builtin_amdgcn_ballot_w64(pay < 1e-14), v_old, 0.0) * disc;
    return lsm_select(__builtin_amdgcn_ballot_w64(pay > 1e-14) & any_itm, fmax(pay, cont), v);
}

// Second launch bound = workgroups per CU the register budget must allow.
template <int NB, int PPT, bool KEEP>
__global__ __launch_bounds__(256, (PPT >= 8 ? 2 : 3)) void k_lsm_coop(LsmCoopArgs a) {
    constexpr int NM = 3 * NB - 1;
    __shared__ double red[NM * 4];
    __shared__ double sm_mom[32];
    __shared__ double sm_coef[LSM_COEF_STRIDE];
    __shared__ double sm_ws[lsm_ws_doubles(NB)];
    const bool call = a.is_call != 0;
    const unsigned G = gridDim.x;
    // Path q of this thread is column first + q * stride.  `first` is the only per-lane part of an address: rows and
    // the q * stride offsets are wave-uniform and stay in scalar registers (a per-lane 64-bit address per path would
    // cost as many VGPRs as V itself).
    const unsigned first = blockIdx.x * 256u + threadIdx.x;
    int64_t stride = (int64_t)G * 256;
    // columns first + q * stride exist for q < n_live (one per-lane integer instead of PPT lane masks)
    const int n_live = (int64_t)first < a.n ? (int)std::min<int64_t>(PPT, (a.n - 1 - first) / stride + 1) : 0;
    int area = 0, round = 0;           // area: slot set of the regular round (0 / 1); round: exchanges so far (workgroup 0 uses it)
    bool gave_up = a.spin_limit == 0;  // this thread has hit the spin limit once (limit 0, tests: from the start)
    if (gave_up && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double V[PPT];
    {
        const double* last = a.data + (int64_t)(a.n_cols - 1) * a.ld;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const double* rq = last + (int64_t)q * stride;
            V[q] = q < n_live ? payoff_of(call, rq[first], a.K) : 0.0;  // LSMPricer.cpp:37-40
        }
    }
    // While the moments travel to workgroup 0 and the coefficients back (~10 us), the next date's row is already
    // on its way from HBM into s_nxt -- when the registers allow a second row (PPT <= 16).
    constexpr bool PREFETCH = PPT <= 16;
    static_assert(KEEP, "the date's prices stay in registers between the regression pass and the update");
    auto load_row = [&](int j, double (&dst)[PPT]) {
        const double* row = a.data + (int64_t)j * a.ld;
        // The PPT offsets q * stride are recomputed on the scalar unit for every row: as loop invariants hipcc
        // hoists all of them, and 2 x PPT scalar registers do not exist.
        asm volatile("" : "+s"(stride));
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const double* rq = row + (int64_t)q * stride;
            dst[q] = q < n_live ? rq[first] : 0.0;
        }
    };
    int j = a.n_cols - 2;
    for (; j >= 0 && j * a.dt > a.maturity; --j) {  // LSMPricer.cpp:43-49, uniform over the grid
#pragma unroll
        for (int q = 0; q < PPT; ++q) V[q] *= a.disc;
    }
    double s_j[PPT], s_nxt[PREFETCH ? PPT : 1];
    if (j >= 0) load_row(j, s_j);
    // A date's 2 NB - 1 power sums (count, sum x, sum x^2, ...) depend on its prices only, its NB cross sums on V as
    // well.  With the next row prefetched (PREFETCH) a workgroup therefore forms and sends the NEXT date's power sums
    // while this date's coefficients are on their way -- into the other slot set, which workgroup 0 reads a round later
    // -- and only the cross sums (and their block reduction) remain between receiving the coefficients and sending the
    // next partial moments.  early: the current date's power sums have gone out already.
    constexpr int NP = 2 * NB - 1;
    __shared__ double red_pow[NP * 4];
    __shared__ double red_cross[NB * 4];
    auto power_sums = [&](const double (&row)[PPT], double (&mp)[NP]) {
#pragma unroll
        for (int t = 0; t < NP; ++t) mp[t] = 0.0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const double s = row[q];
            if (q < n_live && payoff_of(call, s, a.K) > 1e-14) {
                const double x = fma(s, a.invK, -1.0);
                double pw = 1.0;
#pragma unroll
                for (int t = 0; t < NP; ++t) {
                    mp[t] += pw;
                    pw *= x;
                }
            }
        }
    };
    bool early = false;
    for (; j >= 0; --j) {  // every remaining date regresses (this_time <= maturity from here on)
        LSM_TRACE(round, 0);
        double mc[NB];  // cross sums: sum x^t y, y = discounted value one date on (regression inputs, :51-74)
#pragma unroll
        for (int t = 0; t < NB; ++t) mc[t] = 0.0;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const double s = s_j[q];
            if (q < n_live && payoff_of(call, s, a.K) > 1e-14) {
                const double x = fma(s, a.invK, -1.0);
                const double y = V[q] * a.disc;
                double pw = 1.0;
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    mc[t] = fma(pw, y, mc[t]);
                    pw *= x;
                }
            }
        }
        LSM_TRACE(round, 1);  // moments accumulated
        if (!early) {  // (uniform) first regression date, or no prefetch: the power sums go out with the cross sums
            double mp[NP];
            power_sums(s_j, mp);
            lsm_publish_some<NB, 0, NP>(a, mp, G, blockIdx.x, area, red_pow);
        }
        lsm_publish_some<NB, NP, NB>(a, mc, G, blockIdx.x, area, red_cross);
        bool next_early = false;
        if constexpr (PREFETCH) {
            if (j >= 1) load_row(j - 1, s_nxt);
            next_early = j >= 1;
        }
        LSM_TRACE(round, 2);  // published, next row requested
        auto send_next_power_sums = [&]() {
            if constexpr (PREFETCH) {
                if (next_early) {  // (uniform)
                    double mp[NP];
                    power_sums(s_nxt, mp);
                    lsm_publish_some<NB, 0, NP>(a, mp, G, blockIdx.x, area ^ 1, red_pow);
                }
            }
        };
        if (blockIdx.x == 0) {  // workgroup 0 contributes AND reduces: its own next power sums wait until the coefficients are out
            lsm_reduce_solve_publish<NB>(a, G, area, gave_up, sm_mom, sm_coef, false, 0.0, sm_ws, round);
            send_next_power_sums();
            __syncthreads();  // (red_pow's reader, thread 0, is done before anybody can come back to it)
        } else {
            send_next_power_sums();
            lsm_poll_coefficients(a, area, gave_up, sm_coef);
        }
        ++round;
        if (__builtin_amdgcn_readfirstlane(sm_coef[LSM_C_REFINE] != 0.0 ? 1 : 0)) {  // (the same LDS word in every lane)
            // Grid-uniform (every workgroup holds the same coefficient block): the date is re-fitted about the mean of its
            // regressor -- the prices and V are still in registers -- through one more exchange (lsm_solve_nb).
            const double mu = sm_coef[LSM_C_HINT];
            __syncthreads();  // everyone has read the block before the next exchange rewrites sm_coef
            double m[NM];
#pragma unroll
            for (int q = 0; q < NM; ++q) m[q] = 0.0;
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                lsm_accumulate_centered<NB>(m, q < n_live && payoff_of(call, s_j[q], a.K) > 1e-14, s_j[q], V[q], a.invK, mu, a.disc);
            lsm_exchange<NB>(a, m, LSM_AREA_REFINE, gave_up, red, sm_mom, sm_coef, true, mu, sm_ws, round++, []() {});
        }
        double c[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) c[t] = sm_coef[t];
        const double n_itm = sm_coef[LSM_C_COUNT], center = sm_coef[LSM_C_CENTER];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {  // :78-94
            const double s = s_j[q];
            const double pay = payoff_of(call, s, a.K);
            const double vn = V[q] * a.disc;
            double v;
            if (pay > 1e-14 && n_itm > 0.0) {
                v = fmax(pay, lsm_continuation<NB>(c, center, fma(s, a.invK, -1.0)));
            } else if (pay < 1e-14) {
                v = vn;
            } else {
                v = 0.0;
            }
            V[q] = q < n_live ? v : 0.0;
        }
        LSM_TRACE(round - 1, 6);  // V updated
        if (j >= 1) {
            if constexpr (PREFETCH) {
#pragma unroll
                for (int q = 0; q < PPT; ++q) s_j[q] = s_nxt[q];
            } else {
                load_row(j - 1, s_j);
            }
        }
        LSM_TRACE(round - 1, 7);  // next row in registers
        area ^= 1;
        early = next_early;
    }
    double f[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        if (q < n_live) {
            f[0] += V[q];
            f[1] += V[q] * V[q];
        }
    }
    __syncthreads();
    block_sum<2, 4>(f, red);
    if (threadIdx.x == 0) {
        a.out[2 * (int64_t)blockIdx.x] = f[0];
        a.out[2 * (int64_t)blockIdx.x + 1] = f[1];
    }
}

// ------------------------------------------------------------------------------------------------
// The one-launch sweep for shards beyond the register-resident variant above: up to 64 paths per thread, i.e. 8.39M
// paths on 512 co-resident workgroups (2 per CU) -- BASELINE.json's C5 shard (8M x 252) in ONE launch.
// V lives in registers (two adjacent paths per 16-byte unit, NU units per thread: 128 of the 256 VGPRs at NU = 32),
// which leaves no registers to stage loads in.  The path matrix therefore streams through an LDS ring filled by
// LDS-DMA (global_load_lds_dwordx4: HBM -> LDS with no VGPR destination, 1 KiB per wave-instruction into the wave's
// own slice of a slot, so no barrier guards the ring): 16 slots x 4 KiB = 64 KiB per workgroup, i.e. the loads of
// eight units (S_j and S_{j-1}) are always in flight while one unit is being processed, and the first eight units
// of the NEXT date are fetched while the moments and coefficients travel between the workgroups.
// Per date j the loop is the per-date kernels' fused form -- update V with S_j, accumulate date j-1's moments from
// S_{j-1} -- so every row is read twice, 16 B per path and date against their 32 (V never touches memory).  Reading
// each row once would need it on chip between the two passes; tools/ubench_mall.hip shows the memory-side cache does
// not provide that (a 64 MB row re-read 20 us later comes at HBM speed), and with V filling half the register file
// the other half plus LDS cannot hold a row either.
// Branch-free bodies (no exec-masked regions: the unrolled unit loop must stay one basic block so that the loads of
// the next pipeline stage can be scheduled across it), written for the instruction count -- at 64 paths per thread
// this loop, not HBM, is what a date costs.  PayoffFunction (include/core/common.h:8-14) as max(sg s + nsK, 0) with
// (sg, nsK) = (1, -K) for a call and (-1, K) for a put: one FMA, same rounding as s - K / K - s.  A path that is not
// in the money enters the sums with weight w = 0 (one select on the high word of 1.0), i.e. adds exact zeros; in the
// money the products are the same x^t the other kernels form, bit for bit.
// f(integral_constant<int, I>) for I = 0 .. N-1: a loop whose index is a compile-time constant in the body
// (immediate operands of inline asm, slot numbers)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int LSM_RING_SLOTS = 16;  // x 256 lanes x 16 B = 64 KiB of dynamic LDS per workgroup

// s_waitcnt with only the vector-memory counter set (gfx9 encoding: vmcnt in bits 3:0 and 15:14)
__device__ __forceinline__ constexpr int lsm_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }

// The reducing workgroup of k_lsm_big: one exchange per regression date, the same dates in the same order as the
// workers count them (LSMPricer.cpp:42-49).
template <int NB>
__device__ __forceinline__ void lsm_reduce_loop(const LsmCoopArgs& a, unsigned G, double* sm_mom, double* sm_coef, double* ws) {
    bool gave_up = a.spin_limit == 0;
    if (gave_up && threadIdx.x == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int j = a.n_cols - 2;
    for (; j >= 0 && j * a.dt > a.maturity; --j) {
    }
    int area = 0, round = 0;
    for (; j >= 0; --j) {
        lsm_reduce_solve_publish<NB>(a, G, area, gave_up, sm_mom, sm_coef, false, 0.0, ws, round++);
        __syncthreads();
        const bool refine = __builtin_amdgcn_readfirstlane(sm_coef[LSM_C_REFINE] != 0.0 ? 1 : 0) != 0;
        const double mu = sm_coef[LSM_C_HINT];
        __syncthreads();  // sm_mom / sm_coef are rewritten on the next round
        if (refine) {  // the workers answer a refinement request with one more round for the same date
            lsm_reduce_solve_publish<NB>(a, G, LSM_AREA_REFINE, gave_up, sm_mom, sm_coef, true, mu, ws, round++);
            __syncthreads();
        }
        area ^= 1;
    }
}

template <int NB, int NU>
__global__ __launch_bounds__(256, 2) void k_lsm_big(LsmCoopArgs a) {
    constexpr int NM = 3 * NB - 1;
    constexpr int LA = LSM_RING_SLOTS / 2;  // units in flight ahead of the one being processed
    static_assert(NU % LA == 0, "a date's units fill the ring a whole number of times: slot numbers are compile-time");
    typedef double v2d __attribute__((ext_vector_type(2)));
    extern __shared__ double2 ring[];  // [LSM_RING_SLOTS][256]; unit u: S_j in slot 2u mod 16, S_{j-1} in the next
    __shared__ double red[NM * 4];
    __shared__ double sm_mom[32];
    __shared__ double sm_coef[LSM_COEF_STRIDE];
    __shared__ double sm_ws[lsm_ws_doubles(NB)];
    const bool call = a.is_call != 0;
    const int tid = threadIdx.x, wave = tid >> 6;
    // Workgroup 0 holds no paths: it only reduces and solves (lsm_reduce_loop).  As a separate branch of the kernel its
    // registers are allocated without V -- inlined into the workers' exchange, the reduction and the solve would push
    // V (128 registers) out to scratch once per date -- and it starts polling the moment a date begins.
    const unsigned G = gridDim.x - 1;  // workers
    if (blockIdx.x == 0) {
        lsm_reduce_loop<NB>(a, G, sm_mom, sm_coef, sm_ws);
        return;
    }
    const unsigned wg = blockIdx.x - 1;
    // Unit u of this thread = columns 2 (first2 + u * su) and the one after it.  The lane's byte offset lane_off is the
    // only per-lane part of an address (32 bits: a row of this kernel's shards is < 4 GB); row bases and the unit
    // offsets u * su * 16 are wave-uniform and live on the scalar unit.
    const unsigned first2 = wg * 256u + (unsigned)tid;
    const unsigned lane_off = first2 * 16u;
    int64_t su = (int64_t)G * 256;
    const int64_t first = 2 * (int64_t)first2, stride = 2 * su;
    // units 0 .. n0-1 of this thread exist; the last of them may hold one path only (n0 / n1: live first / second paths)
    const int n0 = first < a.n ? (int)std::min<int64_t>(NU, (a.n - 1 - first) / stride + 1) : 0;
    const int n1 = (n0 > 0 && first + (int64_t)(n0 - 1) * stride + 1 >= a.n) ? n0 - 1 : n0;
    // A path beyond the shard is given a price far out of the money (payoff 0: its V stays 0 and it never enters a
    // regression) by two selects on the loaded value; the arithmetic then needs no lane masks.  (The 2 NU masks are
    // recomputed per use: kept, they would be spilled scalar registers.)
    const double safe = call ? 0.0 : 2.0 * a.K + 1.0;
    const LsmPay pf = {call ? 1.0 : -1.0, call ? -a.K : a.K};
    auto sanitize = [&](double2 s, int u) {
        int m0 = n0, m1 = n1;
        asm volatile("" : "+v"(m0), "+v"(m1));
        s.x = u < m0 ? s.x : safe;
        s.y = u < m1 ? s.y : safe;
        return s;
    };
    auto row_of = [&](int j) { return reinterpret_cast<const char*>(a.data + (int64_t)j * a.ld); };
    // Unconditional loads (a predicated load is a branch).  A unit beyond the shard reads at most 512 NU columns past
    // the row's end, i.e. inside the next row (the host launches this kernel for n >= 512 NU only); the terminal row,
    // which has no next row, clamps its index instead.
    auto unit_ptr = [&](const char* row, int u) {
        asm volatile("" : "+s"(su));  // u * su recomputed on the scalar unit per use (hoisted: 2 SGPRs per unit and row)
        return row + (int64_t)u * su * 16 + lane_off;
    };
    // LDS-DMA of unit u of `row` into ring slot `slot` (this wave's 1 KiB of it)
    auto fetch = [&](const char* row, int u, int slot) {
        double2* dst = ring + slot * 256 + wave * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)unit_ptr(row, u),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const unsigned ring_lane = (unsigned)(size_t)(__attribute__((address_space(3))) void*)ring + (unsigned)tid * 16u;
    int area = 0;  // slot set of the regular round
    bool gave_up = a.spin_limit == 0;

    double V[2 * NU];
    {
        const double2* last = reinterpret_cast<const double2*>(row_of(a.n_cols - 1));
        const unsigned last2 = (unsigned)(a.ld / 2 - 1);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const double2 s = sanitize(last[min(first2 + (unsigned)u * (unsigned)su, last2)], u);
            V[2 * u] = lsm_pay(pf, s.x);  // LSMPricer.cpp:37-40
            V[2 * u + 1] = lsm_pay(pf, s.y);
            if ((u % 8) == 7) asm volatile("" ::: "memory");  // eight loads in flight, not all NU
        }
    }
    int j = a.n_cols - 2;
    for (; j >= 0 && j * a.dt > a.maturity; --j) {  // LSMPricer.cpp:43-49, uniform over the grid
#pragma unroll
        for (int q = 0; q < 2 * NU; ++q) V[q] *= a.disc;
    }
    double m[NM];
#pragma unroll
    for (int t = 0; t < NM; ++t) m[t] = 0.0;
    if (j >= 0) {  // moments of the first regression date (plain loads: nothing else is in flight yet)
        const char* row = row_of(j);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const double2 s = sanitize(*reinterpret_cast<const double2*>(unit_ptr(row, u)), u);
            lsm_accumulate<NB>(m, pf, s.x, V[2 * u], a.invK, a.disc);
            lsm_accumulate<NB>(m, pf, s.y, V[2 * u + 1], a.invK, a.disc);
            if ((u % 8) == 7) asm volatile("" ::: "memory");
        }
    }
    for (; j >= 0; --j) {  // every remaining date regresses (this_time <= maturity from here on)
        const char* row_j = row_of(j);
        // At j = 0 there is no earlier date: the pass still runs its moment half, on row 0 again, and nobody reads
        // the result -- cheaper than a second copy of the loop or a branch per unit.
        const char* row_n = row_of(j >= 1 ? j - 1 : 0);
        lsm_publish_partials<NB>(a, m, G, wg, area, red);
#pragma unroll
        for (int u = 0; u < LA; ++u) {  // the first units of this date travel while the moments and coefficients do
            fetch(row_j, u, (2 * u) % LSM_RING_SLOTS);
            fetch(row_n, u, (2 * u + 1) % LSM_RING_SLOTS);
        }
        lsm_poll_coefficients(a, area, gave_up, sm_coef);
        if (__builtin_amdgcn_readfirstlane(sm_coef[LSM_C_REFINE] != 0.0 ? 1 : 0)) {  // (the same LDS word in every lane)
            // Grid-uniform and rare (lsm_solve_nb): date j is re-fitted about the mean of its regressor.  Row j is read
            // once more, by plain loads (V still holds the values its first moments were formed with), and the centred
            // moments take one more exchange with the reducing workgroup.
            const double mu = sm_coef[LSM_C_HINT];
            __syncthreads();  // everyone has read the block before the next poll rewrites sm_coef
#pragma unroll
            for (int t = 0; t < NM; ++t) m[t] = 0.0;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const double2 sv = sanitize(*reinterpret_cast<const double2*>(unit_ptr(row_j, u)), u);
                lsm_accumulate_centered<NB>(m, lsm_pay(pf, sv.x) > 1e-14, sv.x, V[2 * u], a.invK, mu, a.disc);
                lsm_accumulate_centered<NB>(m, lsm_pay(pf, sv.y) > 1e-14, sv.y, V[2 * u + 1], a.invK, mu, a.disc);
                if ((u % 4) == 3) asm volatile("" ::: "memory");
            }
            lsm_publish_partials<NB>(a, m, G, wg, LSM_AREA_REFINE, red);
            lsm_poll_coefficients(a, LSM_AREA_REFINE, gave_up, sm_coef);
        }
        double c[NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) c[t] = sm_coef[t];
        const double center = sm_coef[LSM_C_CENTER];
        const unsigned long long any_itm = sm_coef[LSM_C_COUNT] > 0.0 ? ~0ull : 0ull;
#pragma unroll
        for (int t = 0; t < NM; ++t) m[t] = 0.0;
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            // Units u .. min(u + LA, NU) - 1 have been requested, two DMAs each, and DMAs land in order: unit u is in
            // LDS once at most 2 (that count - 1) are outstanding.  The ring is read by hand-written ds_read (through the
            // compiler every LDS read would wait for ALL DMAs in flight), and the reads are complete before the slots are
            // handed to unit u + LA.
            constexpr int o0 = ((2 * u) % LSM_RING_SLOTS) * 4096, o1 = ((2 * u + 1) % LSM_RING_SLOTS) * 4096;
            constexpr int pending = 2 * ((u + LA < NU ? LA : NU - u) - 1);
            v2d r0, r1;
            asm volatile(
                "s_waitcnt vmcnt(%3)\n\t"
                "ds_read_b128 %0, %2 offset:%4\n\t"
                "ds_read_b128 %1, %2 offset:%5\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(r0), "=&v"(r1)
                : "v"(ring_lane), "n"(pending), "n"(o0), "n"(o1)
                : "memory");
            if (u + LA < NU) {
                fetch(row_j, u + LA, (2 * u) % LSM_RING_SLOTS);
                fetch(row_n, u + LA, (2 * u + 1) % LSM_RING_SLOTS);
            }
            double2 sj = make_double2(r0.x, r0.y), sn = make_double2(r1.x, r1.y);
            if (u >= a.u_full) {  // wave-uniform: only the shard's last unit or two can hold paths that do not exist
                sj = sanitize(sj, u);
                sn = sanitize(sn, u);
            }
            V[2 * u] = lsm_update<NB>(pf, sj.x, V[2 * u], c, center, any_itm, a.invK, a.disc);
            V[2 * u + 1] = lsm_update<NB>(pf, sj.y, V[2 * u + 1], c, center, any_itm, a.invK, a.disc);
            lsm_accumulate<NB>(m, pf, sn.x, V[2 * u], a.invK, a.disc);
            lsm_accumulate<NB>(m, pf, sn.y, V[2 * u + 1], a.invK, a.disc);
            // One unit's arithmetic at a time: interleaving more of them for ILP costs registers this kernel does not
            // have (V alone is 128 of the 256), and two paths already give the scheduler independent chains.  The empty
            // asm makes the moments "used" here: without it hipcc postpones every unit's accumulation to the end of the
            // date and keeps all NU loaded values of row j-1 alive until then.
#pragma unroll
            for (int t = 0; t < NM; ++t) asm volatile("" : "+v"(m[t]));
            __builtin_amdgcn_sched_barrier(0);
        });
        area ^= 1;
    }
    double f[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2 * NU; ++q) {  // paths beyond the shard hold 0
        f[0] += V[q];
        f[1] += V[q] * V[q];
    }
    __syncthreads();
    block_sum<2, 4>(f, red);
    if (tid == 0) {
        a.out[2 * (int64_t)wg] = f[0];
        a.out[2 * (int64_t)wg + 1] = f[1];
    }
}

namespace {

struct CoopVariant {
    const void* fn;
    int ppt;        // paths per thread
    size_t dyn_lds; // dynamic LDS per workgroup (k_lsm_big: the LDS-DMA ring)
    bool exact;     // occupancy is fixed by LDS (2 workgroups per CU): no margin below the query
};

constexpr int LSM_N_VARIANTS = 5;

template <int NB>
const CoopVariant* coop_variants() {
    static const CoopVariant v[LSM_N_VARIANTS] = {
        {(const void*)k_lsm_coop<NB, 4, true>, 4, 0, false},
        {(const void*)k_lsm_coop<NB, 8, true>, 8, 0, false},
        {(const void*)k_lsm_coop<NB, 16, true>, 16, 0, false},
        {(const void*)k_lsm_big<NB, 16>, 32, LSM_RING_SLOTS * 256 * sizeof(double2), true},
        {(const void*)k_lsm_big<NB, 32>, 64, LSM_RING_SLOTS * 256 * sizeof(double2), true}};
    return v;
}

const CoopVariant* coop_variants_for(int nb) {
    switch (nb) {
        case 1: return coop_variants<1>();
        case 2: return coop_variants<2>();
        case 3: return coop_variants<3>();
        case 4: return coop_variants<4>();
        case 5: return coop_variants<5>();
        default: return nullptr;
    }
}

}  // namespace

// Returns MCG_OK with *done = true when the one-launch sweep ran; *done = false when this shape has to take the
// per-date kernels (too many paths for the register file, order too high, or a time-out earlier on this context).
static int run_lsm_coop_impl(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                             int nb, double* sums3, bool* done) {
    *done = false;
    const CoopVariant* vars = coop_variants_for(nb);
    const int64_t N = P->n_paths;
    // Sharded over the GPUs of a node (mcg_comm_init_shm): the reducing workgroups exchange through the node mailbox.
    // The ranks' shards differ by at most a pair of paths, so they all come here, pick a variant (possibly different
    // ones: irrelevant) and run the same number of exchange rounds.
    double* mbox = shm_mailbox_device(ctx);
    const int rounds_needed = 2 * (P->n_steps + 1);
    const int nm = 3 * nb - 1;
    // coop_launch: cleared after a hand-shake time-out, for the next LSM_COOP_RETRY_AFTER prices
    if (!ctx->coop_launch && ctx->coop_retry_in > 0 && --ctx->coop_retry_in == 0) ctx->coop_launch = true;
    bool eligible = vars && ctx->coop_launch && N > 1024 && !(mbox && (rounds_needed > SHM_MAX_ROUNDS || nm > SHM_ROW_DOUBLES));
    // Few paths per thread keep each workgroup's serial work per date short, a small grid keeps the reduction in
    // workgroup 0 short: take the fewest paths per thread that need at most two workgroups per CU, else the most.
    const CoopVariant* use = nullptr;
    int grid = 0, workers = 0;
    static const int min_ppt = study_switch("MCG_LSM_COOP_MIN_PPT", 0);
    static std::atomic<int> occ_cache[10][LSM_N_VARIANTS];  // workgroups per CU of each variant (0 = not asked yet); same on every device
    static std::atomic<int> regs_cache[10][LSM_N_VARIANTS]; // its VGPR count (hipFuncGetAttributes)
    for (int k = 0; eligible && k < LSM_N_VARIANTS; ++k) {
        if (vars[k].ppt < min_ppt) continue;
        int occ = occ_cache[nb][k].load(std::memory_order_relaxed);
        if (occ == 0) {
            if (vars[k].dyn_lds > 48 * 1024 &&
                hipFuncSetAttribute(vars[k].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vars[k].dyn_lds) != hipSuccess) {
                (void)hipGetLastError();
                occ = -1;
            } else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vars[k].fn, 256, vars[k].dyn_lds) != hipSuccess || occ < 1) {
                (void)hipGetLastError();
                occ = -1;
            }
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, vars[k].fn) == hipSuccess) regs_cache[nb][k].store(fa.numRegs, std::memory_order_relaxed);
            else (void)hipGetLastError();
            occ_cache[nb][k].store(occ, std::memory_order_relaxed);
        }
        if (occ < 1) continue;
        // The occupancy query can read one workgroup per CU high for kernels with ~100 SGPRs (MI355X_MICROARCH.md,
        // "Correctness boundaries"), and nothing would reject the over-sized grid: stay an eighth below it -- except
        // where LDS alone fixes two workgroups per CU (k_lsm_big: 64 KiB each of the CU's 160).
        // ... or the registers do: a kernel with more than 168 VGPRs cannot have a third workgroup (4 waves) on a CU
        // whatever the query says (k_lsm_coop at 8 and 16 paths per thread).
        const bool pinned = vars[k].exact || regs_cache[nb][k].load(std::memory_order_relaxed) > 168;
        const int64_t g_max = pinned ? std::min<int64_t>((int64_t)std::min(occ, 2) * ctx->n_cus, LSM_COOP_MAX_GRID)
                                     : std::min<int64_t>((int64_t)std::min(occ, 4) * ctx->n_cus * 7 / 8, LSM_COOP_MAX_GRID);
        const int64_t per_block = 256 * (int64_t)vars[k].ppt;
        // k_lsm_big: workgroup 0 only reduces and solves, the paths belong to workgroups 1 .. grid-1
        const int64_t extra = vars[k].exact ? 1 : 0;
        if ((g_max - extra) * per_block < N) continue;
        if (vars[k].exact && N < 2 * per_block) continue;  // (its unmasked loads assume a row much longer than a unit stride)
        use = &vars[k];
        workers = (int)((N + per_block - 1) / per_block);
        grid = workers + (int)extra;
        if (grid <= 2 * ctx->n_cus) break;
    }
    if (mbox) {  // every rank of the node takes the one-launch sweep, or none does
        int all = 0;
        int rc0 = shm_sum_flag(ctx, use ? 1 : 0, &all);
        if (rc0) return rc0;
        if (all != shm_n_ranks(ctx)) return MCG_OK;
    }
    if (!use) return MCG_OK;
    // buffer: {sum, sum^2} per contributing workgroup | [3][nm][workers] moment slots | [3][LSM_COEF_STRIDE] coefficient slots
    // (Until the per-date solve left scratch memory, workgroups of small grids gathered ALL partial moments themselves and
    // solved redundantly -- one trip per date instead of two.  With a 1-us solve the single reducer is as fast at 244
    // workgroups and twice as fast at 488, where the gathering workgroups' polls crowd out the stores they wait for.)
    const size_t n_slots = 3 * (size_t)nm * workers + 3 * LSM_COEF_STRIDE;  // three slot sets (two alternating + the refinement rounds'), three coefficient blocks
    int rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, 2 * (size_t)workers + n_slots);
    if (rc) return rc;
    LsmCoopArgs a;
    a.data = P->data;
    a.ld = P->ld;
    a.n = N;
    a.n_cols = P->n_steps + 1;
    a.K = K;
    a.invK = 1.0 / K;
    a.maturity = maturity;
    a.dt = dt;
    a.disc = std::exp(-r * dt);  // LSMPricer.cpp:46,:69,:92
    a.is_call = is_call;
    a.out = ctx->partials;  // finish_sums reads {sum, sum^2} pairs from the head of the buffer
    a.partials = ctx->partials + 2 * (size_t)workers;
    a.coef = a.partials + 3 * (size_t)nm * workers;
    a.timeout = reinterpret_cast<unsigned*>(ctx->scalars + SC_BARRIER);
    a.u_full = (int)(N / ((int64_t)workers * 512));  // (k_lsm_big: units per thread that every thread has in full)
    a.mbox = mbox;
    a.mb_ranks = shm_n_ranks(ctx);
    a.mb_rank = shm_rank(ctx);
    double* const* peers = shm_mailbox_peers(ctx);
    a.mb_push = peers ? 1 : 0;
    for (int q = 0; q < SHM_MAX_RANKS; ++q) a.mb_peer[q] = peers ? peers[q] : nullptr;
    // test hooks (mcg_debug_lsm_hooks): spin limit 0 makes every wait give up at once and raises the time-out flag,
    // which drives the time-out -> per-date fall-back branch below on a healthy device
    a.spin_limit = ctx->lsm_spin_limit < 0 ? LSM_SPIN_LIMIT : (unsigned)ctx->lsm_spin_limit;
    a.poll_delay = (unsigned)ctx->lsm_poll_delay;
    MCG_HIP(hipMemsetD32Async((hipDeviceptr_t)a.partials, (int)LSM_SENTINEL32, 2 * n_slots, ctx->stream));
    MCG_HIP(hipMemsetAsync(a.timeout, 0, sizeof(unsigned), ctx->stream));
    if (mbox) {  // this rank's mailbox rows hold the reserved NaN again, and so do everybody else's, before anyone launches
        rc = shm_arm_mailbox(ctx, rounds_needed, ((uint64_t)LSM_SENTINEL32 << 32) | LSM_SENTINEL32);
        if (rc) return rc;
    }
    void* params[] = {&a};
    {
        // One such kernel at a time per device and process: two grids of spinning workgroups that are each only partly
        // resident would wait for each other.  The lock is held until the stream has drained.  The sweeps of ONE sharded
        // job are the exception -- they wait for each other's moments, so they must be in flight together: ranks that
        // are threads of this process (one per GPU, or several on one GPU in a rehearsal) share the lock.  (An ordinary
        // launch, not hipLaunchCooperativeKernel: nothing of the cooperative-groups runtime is used, the grid is sized to
        // be co-resident by the occupancy query above, and rocprofv3 crashes at exit after a cooperative launch.)
        static std::shared_mutex one_at_a_time[LSM_MAX_DEVICES];
        std::shared_mutex& lk = one_at_a_time[ctx->device % LSM_MAX_DEVICES];
        struct Hold {
            std::shared_mutex& m;
            bool shared;
            Hold(std::shared_mutex& mm, bool sh) : m(mm), shared(sh) { shared ? m.lock_shared() : m.lock(); }
            ~Hold() { shared ? m.unlock_shared() : m.unlock(); }
        } hold(lk, mbox != nullptr);
        {
            TimedLaunch t(ctx, MCG_K_LSM_SWEEP);
            MCG_HIP(hipLaunchKernel(use->fn, dim3((unsigned)grid), dim3(256), params, use->dyn_lds, ctx->stream));
        }
        MCG_HIP(hipMemcpyAsync(ctx->h_scalars + SC_BARRIER, a.timeout, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        rc = finish_sums(ctx, workers, N, sums3);  // synchronises the stream
    }
    if (rc) return rc;
#ifdef MCG_LSM_TRACE
    {
        unsigned long long tr[32 * 8];
        if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_lsm_trace), sizeof(tr)) == hipSuccess) {
            for (int r = 4; r < 12; ++r) {
                std::fprintf(stderr, "lsm trace round %2d (us):", r);
                for (int k = 1; k < 8; ++k) std::fprintf(stderr, " %d:%.2f", k, (double)(tr[r * 8 + k] - tr[r * 8]) * 0.01);
                std::fprintf(stderr, "  next:%.2f\n", (double)(tr[(r + 1) * 8] - tr[r * 8]) * 0.01);
            }
        }
    }
#endif
    int timed_out = reinterpret_cast<const unsigned*>(ctx->h_scalars + SC_BARRIER)[0] != 0 ? 1 : 0;
    if (mbox) {  // the ranks agree: one time-out anywhere voids the sweep everywhere (they fall back together)
        rc = shm_sum_flag(ctx, timed_out, &timed_out);
        if (rc) return rc;
    }
    if (timed_out) {
        // A spin gave up: the grid was not co-resident after all.  The result is discarded, this context stops using
        // the one-launch sweep, and the caller runs the per-date kernels.
        // Another process was holding part of the GPU: that passes.  The next LSM_COOP_RETRY_AFTER prices of this ctx
        // take the per-date kernels, then the one-launch sweep is tried again (sharded: every rank counts the same
        // prices, the time-out flag having been summed over the ranks); mcg_lsm_one_launch_reset does it at once.
        ctx->coop_launch = false;
        ctx->coop_retry_in = LSM_COOP_RETRY_AFTER;
        g_stats.lsm_one_launch_timeouts.fetch_add(1, std::memory_order_relaxed);  // (reported through mcg_stats; the library prints nothing)
        return MCG_OK;
    }
    g_stats.lsm_one_launch_sweeps.fetch_add(1, std::memory_order_relaxed);
    *done = true;
    return MCG_OK;
}

static int run_lsm_coop(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                        int nb, double* sums3, bool* done) {
    const int rc = run_lsm_coop_impl(ctx, P, r, K, maturity, dt, is_call, nb, sums3, done);
    // Sharded over the node segment, a rank that fails locally between two collective steps would leave its peers
    // waiting in a barrier it never enters: poison the segment, so that they fail at once with MCG_ERR_COMM.
    if (rc != MCG_OK) shm_poison(ctx);
    return rc;
}

// partials[grid][nm] -> moments (fixed order) -> optional all-reduce -> coefficients in ctx->scalars.
// MartingaleOptimization's refit (the LSM sweep has its own per-date kernel, k_lsm_date).
// mo_mode: 1 = first pass, leave the refinement request in the coefficient block; 2 = the moments are about mo_mu,
// solve them with lsm_solve_centered.
int lsm_reduce_allreduce_solve(mcg_ctx* ctx, int grid, int nm, int nb, double min_count, double K, int mo_mode, double mo_mu) {
    double* moments = ctx->scalars + SC_LSM_MSG;  // (room for the 3p+2 moments + the primal sum of orders up to 15)
    double* coef = ctx->scalars + SC_COEF;
    LsmRefine rf{K, mo_mode == 1, mo_mode == 2, mo_mu};
    if (ctx->allreduce) {
        {
            TimedLaunch t(ctx, MCG_K_LSM_SOLVE);
            hipLaunchKernelGGL(k_lsm_reduce_solve, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, grid, nm, nb,
                               moments, coef, 1, 0, min_count, rf);
        }
        if (ctx->allreduce(ctx->allreduce_user, moments, nm, (void*)ctx->stream) != 0)
            return fail(MCG_ERR_COMM, "all-reduce of regression moments failed");
        TimedLaunch t(ctx, MCG_K_LSM_SOLVE);
        hipLaunchKernelGGL(k_lsm_reduce_solve, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, grid, nm, nb, moments,
                           coef, 0, 1, min_count, rf);
    } else {
        TimedLaunch t(ctx, MCG_K_LSM_SOLVE);
        hipLaunchKernelGGL(k_lsm_reduce_solve, dim3(1), dim3(256), 0, ctx->stream, ctx->partials, grid, nm, nb, moments,
                           coef, 1, 1, min_count, rf);
    }
    MCG_HIP(hipGetLastError());
    return MCG_OK;
}

// k_lsm_date<nb> for nb = 1 .. LSM_MAX_NB (poly_order 0 .. 15)
typedef void (*LsmDateKernel)(LsmDateArgs);
static LsmDateKernel date_kernel(int nb) {
    static const LsmDateKernel k[LSM_MAX_NB] = {k_lsm_date<1>,  k_lsm_date<2>,  k_lsm_date<3>,  k_lsm_date<4>,
                                                k_lsm_date<5>,  k_lsm_date<6>,  k_lsm_date<7>,  k_lsm_date<8>,
                                                k_lsm_date<9>,  k_lsm_date<10>, k_lsm_date<11>, k_lsm_date<12>,
                                                k_lsm_date<13>, k_lsm_date<14>, k_lsm_date<15>, k_lsm_date<16>};
    return k[nb - 1];
}

// Workgroups of k_lsm_date<nb> a CU holds at once: the per-date route launches exactly one resident wave of them.
static int date_kernel_occupancy(int nb) {
    static std::atomic<int> cache[LSM_MAX_NB + 1];
    int occ = cache[nb].load(std::memory_order_relaxed);
    if (occ > 0) return occ;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)date_kernel(nb), 256, 0) != hipSuccess || occ < 1) {
        (void)hipGetLastError();
        occ = 1;
    }
    // three per CU even where four fit (orders <= 3): 768 workgroups stream as fast as 1024 and leave fewer partials and
    // tickets to the tail (C5 shard, span per pass: 10.86 ms at four, 10.69 at three and at two)
    const int want = std::max(1, study_switch("MCG_LSM_DATE_WGS_PER_CU", 3));
    occ = std::min(occ, want);
    cache[nb].store(occ, std::memory_order_relaxed);
    return occ;
}

static void launch_date(mcg_ctx* ctx, int nb, int grid, const LsmDateArgs& a) {
    hipLaunchKernelGGL(date_kernel(nb), dim3(grid), dim3(256), 0, ctx->stream, a);
}

int run_lsm(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
            int poly_order, double* mean, double* std_err) {
    const int nb = poly_order + 1;
    const int nm = 3 * nb - 1;
    const int64_t N = P->n_paths;
    const int M = P->n_steps + 1;
    int grid = (int)std::min<int64_t>((N + 511) / 512, (int64_t)ctx->n_cus * date_kernel_occupancy(nb));  // 512 paths per workgroup and trip
    grid = std::min(grid, LSM_DATE_GROUP * LSM_DATE_MAX_GROUPS);
    if (grid < 1) grid = 1;

    if (N >= 1 && N <= 1024 && !ctx->allreduce && nb <= 9) {  // one launch for the whole sweep
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
    int rc;
    if ((N > 1024 && !ctx->allreduce) || shm_mailbox_device(ctx)) {  // (with the node mailbox every rank asks: they agree inside)
        bool done = false;
        double s3[3];
        rc = run_lsm_coop(ctx, P, r, K, maturity, dt, is_call, nb, s3, &done);
        if (rc) return rc;
        if (done) {
            const double n = s3[2], m = s3[0] / n;
            *mean = m;
            if (std_err) {
                const double var = n > 1.0 ? std::max(0.0, (s3[1] - n * m * m) / (n - 1.0)) : 0.0;
                *std_err = std::sqrt(var / n);
            }
            return MCG_OK;
        }
    }
    rc = ensure_cap(ctx, &ctx->lsm_v, &ctx->lsm_v_cap, (size_t)std::max<int64_t>(N, 1) + 1);  // (a whole two-path unit at the end)
    if (rc) return rc;
    rc = ensure_cap(ctx, &ctx->partials, &ctx->partials_cap, ((size_t)grid + LSM_DATE_MAX_GROUPS) * (size_t)std::max(nm, 2));  // (k_lsm_final: 2 per workgroup)
    if (rc) return rc;

    const double disc = std::exp(-r * dt);  // LSMPricer.cpp:46,:69,:92
    LsmDateArgs a;
    a.data = P->data;
    a.ld = P->ld;
    a.n = N;
    a.V = ctx->lsm_v;
    a.K = K;
    a.invK = 1.0 / K;
    a.disc = disc;
    a.dt = dt;
    a.maturity = maturity;
    a.is_call = is_call;
    a.msg = ctx->scalars + SC_LSM_MSG;
    a.state = ctx->scalars + SC_LSM_STATE;
    a.partials = ctx->partials;
    a.ticket = reinterpret_cast<unsigned*>(ctx->scalars + SC_LSM_TICKET);
    a.spin_limit = ctx->lsm_date_spin_limit < 0 ? LSM_DATE_SPIN_LIMIT : (unsigned)ctx->lsm_date_spin_limit;
    a.hook_mode = ctx->lsm_date_hook[0];
    a.hook_date = ctx->lsm_date_hook[1];
    a.hook_wg = ctx->lsm_date_hook[2];
    a.hook_delay = ctx->lsm_date_hook[3];
    // Consecutive dates walk the paths in opposite directions (k_lsm_date: by the parity of j): what date j touched last
    // (the tail of V and of row j-1, which date j-1 reads again) is what date j-1 touches first, while it is still in
    // the 256 MB memory-side cache.
    ctx->h_scalars[SC_LSM_STATE + LSM_ST_J] = (double)(M - 1);  // the terminal payoff, fused with the moments of date M-2
    ctx->h_scalars[SC_LSM_STATE + LSM_ST_PHASE] = (double)LSM_PH_INIT;
    ctx->h_scalars[SC_LSM_STATE + LSM_ST_MU] = 0.0;
    ctx->h_scalars[SC_LSM_STATE + LSM_ST_FAULT] = 0.0;
    MCG_HIP(hipMemcpyAsync(a.state, ctx->h_scalars + SC_LSM_STATE, LSM_ST_WORDS * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MCG_HIP(hipMemsetAsync(a.ticket, 0, (LSM_DATE_MAX_GROUPS + 1) * sizeof(unsigned), ctx->stream));
    MCG_HIP(hipMemsetAsync(a.msg, 0, 48 * sizeof(double), ctx->stream));
    // every partial-moment slot (workgroups' and groups') starts armed with the reserved NaN
    MCG_HIP(hipMemsetD32Async((hipDeviceptr_t)a.partials, (int)LSM_SENTINEL32, 2 * ((size_t)grid + LSM_DATE_MAX_GROUPS) * (size_t)nm, ctx->stream));

    // Exactly M launches when no date asks for a re-fit, one more per date that does (orders >= 4: every date with a path
    // in the money; at lower orders only near-degenerate dates).  The host queues M, reads the state back and queues what is
    // left -- one launch per remaining date plus the share of second launches the dates behind it took -- until the sweep is
    // through (a batch advances the sweep by at least half its launches).  A batch may overshoot the end by the few launches
    // its estimate is off by -- at ANY order once a date has re-fitted: such a launch returns at once (k_lsm_date: j < 0), and
    // sharded it is still preceded by its all-reduce of a.msg, which then re-sums moments nobody reads: a.msg is DEAD after
    // the last date (nothing below this loop touches it; k_lsm_final reads V only) -- keep it that way.  Launches and
    // collectives stay paired on every rank (launches = collectives + 1), so the ranks never disagree about a collective.
    int dates_left = M, progress = 2 * M + 1;  // progress: launches the sweep still needs at least, x 2 (must fall with every batch)
    int64_t batch = M;
    bool first = true;
    g_stats.lsm_per_date_sweeps.fetch_add(1, std::memory_order_relaxed);
    while (dates_left > 0) {
        {
            // timing: ONE event pair around the queued sequence (launches, the gaps and the collectives between them)
            TimedLaunch t(ctx, MCG_K_LSM_SWEEP, batch);
            for (int64_t k = 0; k < batch; ++k) {
                if (!first && ctx->allreduce) {
                    if (ctx->allreduce(ctx->allreduce_user, a.msg, nm, (void*)ctx->stream) != 0)
                        return fail(MCG_ERR_COMM, "all-reduce of regression moments failed");
                }
                first = false;
                launch_date(ctx, nb, grid, a);
            }
        }
        g_stats.lsm_per_date_launches.fetch_add(batch, std::memory_order_relaxed);
        MCG_HIP(hipGetLastError());
        if (ctx->allreduce) {  // a rank whose consumers gave a slot up ends its sweep: all ranks must, together
            if (ctx->allreduce(ctx->allreduce_user, a.state + LSM_ST_FAULT, 1, (void*)ctx->stream) != 0)
                return fail(MCG_ERR_COMM, "all-reduce of the sweep's fault flag failed");
        }
        MCG_HIP(hipMemcpyAsync(ctx->h_scalars + SC_LSM_STATE, a.state, LSM_ST_WORDS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        MCG_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->h_scalars[SC_LSM_STATE + LSM_ST_FAULT] != 0.0) {
            g_stats.lsm_per_date_faults.fetch_add(1, std::memory_order_relaxed);
            return fail(MCG_ERR_HIP, "LSM per-date sweep: partial moments of a workgroup did not arrive (on %s rank); the price is void",
                        ctx->allreduce ? "some" : "this");
        }
        const int left = (int)ctx->h_scalars[SC_LSM_STATE + LSM_ST_J] + 1;
        const int half_done = (int)ctx->h_scalars[SC_LSM_STATE + LSM_ST_PHASE] == LSM_PH_REFINED ? 1 : 0;  // its re-fit's moments are in
        if (2 * left - half_done >= progress) return fail(MCG_ERR_HIP, "LSM per-date sweep did not advance (%d dates left)", left);
        progress = 2 * left - half_done;
        const int64_t dates_done = dates_left - left, second_launches = batch - dates_done;
        g_stats.lsm_per_date_refits.fetch_add(second_launches, std::memory_order_relaxed);
        dates_left = left;
        // The next batch: one launch per remaining date PLUS as many second launches as the dates just swept took on
        // average (orders >= 4 re-fit every date with a path in the money: a batch of `left` launches would cover half of
        // what is left, and the sweep would end after ~log2(M) read-backs, each a host synchronisation and, sharded, one more
        // collective -- ADVICE r4).  With the ratio carried over the sweep ends after three or four batches; should a batch
        // overshoot, a launch past the end returns at once (k_lsm_date: j < 0) and is counted with the second launches.
        // Order 2: the ratio is 0.
        // (rounded DOWN, and only while at least 16 dates are left: the estimate should fall short by a few launches, which
        //  the next, small batch picks up, rather than overshoot; the last dates go one launch per date as before)
        const int64_t extra = dates_done > 0 && left >= 16 && study_switch("MCG_LSM_DATE_ADAPTIVE", 1)
                                  ? std::min<int64_t>(left, second_launches * left / dates_done) : 0;
        batch = dates_left + extra;
    }

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
