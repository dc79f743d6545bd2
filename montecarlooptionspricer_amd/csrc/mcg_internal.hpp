// Internal definitions behind include/mcgpu.h: context, device-buffer pool, error plumbing,
// per-kernel HIP-event timing.  Not installed; not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mcgpu.h"
#include "../../include/mcgpu_debug.h"

namespace mcg {

void set_error(const char* fmt, ...);
int fail(int status, const char* fmt, ...);

// Timing-study switches (another grid size, another variant): read from the environment in A/B builds only
// (tools/build_variant.sh ... -DMCG_STUDY_SWITCHES); the product library reads no such variable.
inline int study_switch(const char* name, int fallback) {
#ifdef MCG_STUDY_SWITCHES
    if (const char* e = std::getenv(name)) return std::atoi(e);
#else
    (void)name;
#endif
    return fallback;
}

// Process-wide event counters behind mcg_stats (relaxed atomics: any thread, any ctx).
struct Stats {
    std::atomic<int64_t> lsm_one_launch_sweeps{0}, lsm_one_launch_timeouts{0}, lsm_per_date_sweeps{0}, lsm_per_date_launches{0},
        lsm_per_date_refits{0}, lsm_per_date_faults{0}, shm_barrier_failures{0}, peer_mailbox_enabled{0}, peer_mailbox_refused{0},
        batch_calls{0}, batch_chunks{0}, batch_rows{0}, batch_rows_singly{0}, batch_peak_workspace_bytes{0}, peer_mailbox_kept{0},
        coalesced_rounds{0}, coalesced_calls{0}, coalesced_peak_calls_per_round{0}, coalesced_fallbacks{0},
        coalesced_round_us{0}, coalesced_device_wait_us{0}, coalesced_wake_us{0}, coalesced_prefetched{0}, coalesced_prefetch_hits{0};
};
extern Stats g_stats;

#define MCG_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::mcg::fail(_e == hipErrorOutOfMemory ? MCG_ERR_OOM : MCG_ERR_HIP,          \
                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                               __LINE__);                                                      \
    } while (0)

struct PoolBuf {
    void* ptr;
    size_t bytes;
};

struct EventPair {
    hipEvent_t a, b;
    int64_t launches;  // kernel launches the pair brackets (1, or a whole queued sequence: TimedLaunch)
};

struct ShmComm;  // comm_shm.cpp: node-local collective over POSIX shared memory
constexpr int SHM_MAX_RANKS = 16;     // processes (GPUs) sharing one segment
constexpr int SHM_MAX_ROUNDS = 4096;  // exchange rounds of one LSM sweep the device mailbox holds (2 per exercise date at most)
constexpr int SHM_ROW_DOUBLES = 16;   // one rank's row of a round: up to 14 moments (orders <= 4)

}  // namespace mcg

struct mcg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int n_cus = 256;
    bool coop_launch = false;    // the one-launch LSM sweep (k_lsm_coop / k_lsm_big) is allowed on this context
    int coop_retry_in = 0;       // after a hand-shake time-out: LSM prices left before it is allowed again (0: stays off until reset)
    long long lsm_spin_limit = -1;  // test hooks (mcg_debug_lsm_hooks): polling rounds before a spin gives up (< 0: default)
    int lsm_poll_delay = 0;         //   workgroups other than 0 reach their coefficient poll late
    long long lsm_date_spin_limit = -1;  // mcg_debug_lsm_date_fault: polls before a consumer of k_lsm_date gives a slot up (< 0: default)
    int lsm_date_hook[4] = {0, 0, 0, 0}; //   {mode, date, workgroup, delay}: that workgroup's partial moments never land (1) / land late (2)
    size_t batch_budget = 0;             // mcg_debug_batch_budget: workspace bytes of one chunk of mcg_batch_price_rows (0: a quarter of free memory)
    // batched rows (kernels_batch.hip): the two latency-bound row pricers run beside the two issue-bound ones on a second stream
    hipStream_t batch_aux = nullptr;
    hipEvent_t batch_fork = nullptr, batch_join = nullptr;

    // cached device buffers (path matrices are tens of GB: never hipMalloc per call in steady state)
    std::vector<mcg::PoolBuf> pool;
    // path-matrix handles handed out and not yet freed: mcg_finalize takes their device memory back and orphans
    // them (ctx = data = NULL), so a late mcg_paths_free only deletes the handle instead of touching a dead ctx
    std::vector<mcg_paths*> live_paths;
    // small persistent workspace
    double* partials = nullptr;  // per-block partial sums
    size_t partials_cap = 0;     // in doubles
    double* fin_chunks = nullptr;  // finish_sums: chunk sums of a long partials list
    size_t fin_chunks_cap = 0;
    double* scalars = nullptr;   // device: sums, moments, coefficients
    double* h_scalars = nullptr; // pinned host mirror
    double* weights = nullptr;   // rBergomi spectral amplitudes + compensator
    size_t weights_cap = 0;
    double* lsm_v = nullptr;     // LSM value vector
    size_t lsm_v_cap = 0;
    unsigned long long* clk_stamps = nullptr;  // [GBM_CLK_SLOTS][2] {shader cycles, 100 MHz ticks} of the last GBM launch's stamping workgroups
    int clk_slots_used = 0;
    bool clk_armed = false;                    // mcg_generator_clock_arm: only armed launches stamp (and pay the memset ahead of them)
    double* log_tab = nullptr;   // device copy of fm::LOG_TAB_HOST + fm::SINCOS_TAB_HOST + fm::EXP2_TAB_HOST (34 KiB), staged to LDS

    // collective
    mcg_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    void* rccl_comm = nullptr;
    mcg::ShmComm* shm = nullptr;  // mcg_comm_init_shm
    int n_ranks = 1, rank = 0;

    // timing
    bool timing = false;
    unsigned timing_mask = ~0u;  // bit k: kernel k is bracketed while timing is on (mcg_timing_select)
    std::vector<mcg::EventPair> ev_free;
    std::vector<std::pair<int, mcg::EventPair>> ev_live;
    double t_total[MCG_K_COUNT] = {0};
    int64_t t_count[MCG_K_COUNT] = {0};
};

struct mcg_paths {
    mcg_ctx* ctx = nullptr;
    double* data = nullptr;
    size_t bytes = 0;
    int64_t n_paths = 0;
    int n_steps = 0;
    int64_t ld = 0;
    uint64_t path_begin = 0;
    // fused terminal-payoff sums left by *_payoff generators
    bool has_sums = false;
    double sums_K = 0.0;
    int sums_is_call = 0;
    double sums[3] = {0, 0, 0};  // host copy {sum, sumsq, n}: all-reduced totals when the ctx has a collective
};

namespace mcg {

constexpr int GBM_CLK_SLOTS = 64;     // stamping workgroups of a GBM generator launch (mcg_generator_clock)
constexpr int SCALARS_DOUBLES = 256;
// layout of ctx->scalars (doubles)
constexpr int SC_SUMS = 0;     // [0..3)  sum, sumsq, n
constexpr int SC_COEF = 40;    // [40..60) the LSM coefficient block of the current date (lsm_device.hpp: LSM_C_*)
constexpr int SC_FINAL = 64;   // [64..67) LSM final sums
constexpr int SC_BARRIER = 72; // [72] 32-bit timeout flag of k_lsm_coop's hand-shake
constexpr int SC_TICKET = 80;  // [80] 64-bit share ticket of the persistent rBergomi generator (zeroed before each launch)
constexpr int SC_LSM_TICKET = 160; // [160..225) 129 32-bit "workgroups done" tickets of the per-date LSM kernel (k_lsm_date)
constexpr int SC_LSM_MSG = 96;     // [96..144) per-date LSM message: the 3p+2 <= 47 moments the next launch solves (the all-reduced part)
constexpr int SC_LSM_STATE = 144;  // [144..147) per-date LSM state: date, phase, centre (kernels_lsm.hip: LSM_ST_*)

int pool_alloc(mcg_ctx* ctx, size_t bytes, void** out);
void pool_release(mcg_ctx* ctx, void* ptr, size_t bytes);
int ensure_cap(mcg_ctx* ctx, double** buf, size_t* cap, size_t need_doubles);

// timing scope: records events around a launch when ctx->timing is on
struct TimedLaunch {
    mcg_ctx* ctx;
    int kernel;
    EventPair ev{};
    bool on;
    // launches > 1: ONE event pair around a queued sequence of that many launches (the per-date LSM route: an event
    // pair per launch costs 9 us of the 60 a date takes); the reported time then includes what lies between them
    TimedLaunch(mcg_ctx* c, int k, int64_t launches = 1);
    ~TimedLaunch();
};

void comm_release(mcg_ctx* ctx);  // comm_rccl.cpp
int comm_rccl_count(mcg_ctx* ctx);  // ncclCommCount of the built-in communicator (0: none / not available)
// comm_shm.cpp
void shm_release(mcg_ctx* ctx);
int shm_arm_mailbox(mcg_ctx* ctx, int rounds, uint64_t sentinel_bits);
int shm_sum_flag(mcg_ctx* ctx, int flag, int* total);
void shm_poison(mcg_ctx* ctx);  // a rank failed between two collective steps: every later barrier fails at once on every rank
double* shm_mailbox_device(mcg_ctx* ctx);        // the mailbox this rank polls (host segment, or its own HBM)
double* const* shm_mailbox_peers(mcg_ctx* ctx);  // peer-memory mode: every rank's mailbox as mapped here; else nullptr
int shm_rank(mcg_ctx* ctx);
int shm_n_ranks(mcg_ctx* ctx);
int shm_attached(mcg_ctx* ctx);
bool shm_peer_active(mcg_ctx* ctx);
int peer_ping(mcg_ctx* ctx, double* const* peers, int n, int rank);  // comm_peer.hip
int peer_arm(mcg_ctx* ctx, double* mbox, int rounds, int n, uint64_t sentinel);  // comm_peer.hip: 0 = armed and synchronised

int paths_new(mcg_ctx* ctx, int64_t n_paths, int n_steps, uint64_t path_begin, mcg_paths** out);

// kernel launchers (one per .hip file)
int launch_gbm(mcg_ctx* ctx, mcg_paths* P, uint64_t seed, double S0, double r, double sigma, double dt,
               bool want_payoff, double K, int is_call);
int launch_rbergomi(mcg_ctx* ctx, mcg_paths* P, uint64_t seed, double S0, double r, double xi, double H,
                    double eta, double dt, bool want_payoff, double K, int is_call);
int launch_payoff_sums(mcg_ctx* ctx, const mcg_paths* P, double K, int is_call, double out3[3]);
int generator_clock(mcg_ctx* ctx, double* ghz_median, int* n_stamps, double* ghz_min, double* ghz_max);  // kernels_gbm.hip
int finish_sums(mcg_ctx* ctx, int64_t n_blocks, int64_t n_local, double out3[3]);
int run_lsm(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
            int poly_order, double* mean, double* std_err);

// MartingaleOptimization's refit (its driver re-accumulates on request): mo_mode 1 = first pass, leave the refinement
// request in the coefficient block; 2 = the moments are about mo_mu, solve them with lsm_solve_centered.
int lsm_reduce_allreduce_solve(mcg_ctx* ctx, int grid, int nm, int nb, double min_count, double K, int mo_mode, double mo_mu);
int run_martingale(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                   int poly_order, int max_iterations, double* price, double* lower, double* upper);
int run_branching(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                  int num_branches, const int* exercise_times, int n_ex, uint64_t seed, double* price, double* lower,
                  double* upper);
// priced (optional, n_rows bytes): 1 where the row was priced, 0 where the driver's own checks answer it with zeros
int run_batch_rows(mcg_ctx* ctx, const mcg_row* rows, int64_t n_rows, int n_paths, double r, double dt, int num_branches,
                   int poly_order, int max_iterations, uint64_t seed, double* out, unsigned char* priced);
int probe_write_ceiling(mcg_ctx* ctx, int64_t n_paths, int n_steps, int reps, double* gb_per_s, double* ms_per_launch);  // kernels_probe.hip
int run_asymptotic(mcg_ctx* ctx, const mcg_paths* P, double r, double K, double maturity, double dt, int is_call,
                   double sigma, double dividend, double* price);

// host-side math (host/volterra.cpp, host/estimators.cpp, host/asymptotic.cpp)
void host_asymptotic_tables(int n_cols, double r, double K, double maturity, double dt, int is_call, double sigma,
                            double dividend, std::vector<double>& bnd, std::vector<double>& disc);
int host_estimate_params(const double* hist, size_t n, double out5[5]);
void host_row_features(const double* hist, size_t n, double* vol, double* momentum);  // host/features.cpp
int host_row_build(const double* hist, size_t n, double underlying_last, double dte, double strike_dist_pct, int option_type,
                   double dividend, mcg_row* row, double features2[2]);
int host_rbergomi_spectrum(double H, double eta, double dt, int n_steps, std::vector<double>& amp,
                           std::vector<double>& comp);

}  // namespace mcg
