/* mcgpu.h -- C ABI of the MI355X-native Monte Carlo path engine (libmcgpu.so).
 *
 * This is the drop-in boundary for the hot path of bcosm/MonteCarloOptionsPricer
 * (RNG -> GBM / rBergomi time-stepping -> per-path payoff -> Longstaff-Schwartz sweep).
 * Plain pointers and sizes only; no C++ or torch types cross it.  The host-side C++ classes with
 * the reference's exact signatures (include/models/RoughVolatility.h, include/models/LSMPricer.h)
 * are thin shims over these entry points; INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - every function returns an int status (MCG_OK == 0); nothing throws across the ABI;
 *     mcg_last_error() returns a thread-local message for the last non-zero status on this thread.
 *   - all arithmetic and storage is IEEE binary64, like the reference.
 *   - a path matrix lives on the device, STEP-MAJOR: element (step j, path p) at data[j*ld + p],
 *     j = 0..n_steps (column 0 = S0, like RoughVolatility.cpp:344,:354), p = 0..n_paths-1.
 *     The reference's host layout (path-major vector<vector<double>>) is produced/consumed by
 *     mcg_paths_to_host / mcg_paths_from_host.
 *   - RNG contract: Philox4x32-10, key = seed, counter = (global path id, block, stream); a path's
 *     values depend only on (seed, global path id), never on how paths are sharded over GPUs.
 *   - re-entrant: one mcg_ctx per host thread (or per GPU); a ctx owns its stream and workspace.
 */
#ifndef MCGPU_H
#define MCGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcg_ctx mcg_ctx;     /* device, stream, workspace, optional collective */
typedef struct mcg_paths mcg_paths; /* device-resident (n_steps+1) x n_paths matrix      */

enum mcg_status {
    MCG_OK = 0,
    MCG_ERR_INVALID = 1,           /* bad argument                                        */
    MCG_ERR_NO_DEVICE = 2,         /* no usable MI355X / HIP runtime                      */
    MCG_ERR_HIP = 3,               /* a HIP call failed (message has the HIP error)       */
    MCG_ERR_OOM = 4,               /* device allocation failed                            */
    MCG_ERR_HISTORY_TOO_SMALL = 5, /* == RoughVolatility.cpp:317-319                      */
    MCG_ERR_EMPTY_PATHS = 6,       /* == LSMPricer.cpp:28-30                              */
    MCG_ERR_COMM = 7               /* collective failed                                   */
};

/* kernels whose device time mcg_timing_get reports */
enum mcg_kernel {
    MCG_K_GBM = 0,        /* GBM path generation (+ fused payoff partials)  */
    MCG_K_RBERGOMI = 1,   /* rBergomi path generation                       */
    MCG_K_PAYOFF = 2,     /* terminal payoff reduction over a stored matrix */
    MCG_K_LSM_SWEEP = 3,  /* LSM sweep: the one launch, or the queued per-date sequence (launches, gaps, collectives) */
    MCG_K_LSM_SOLVE = 4,  /* LSM per-date reduce+solve kernels              */
    MCG_K_TRANSPOSE = 5,  /* layout change for the host class API           */
    MCG_K_ASYM = 6,       /* AsymptoticAnalysis boundary scan               */
    MCG_K_MARTINGALE = 7, /* MartingaleOptimization primal/offset/dual scans */
    MCG_K_BRANCHING = 8,  /* BranchingProcesses suffix-max + bounds kernels   */
    MCG_K_BATCH = 9,      /* the six kernels of mcg_batch_price_rows (one span) */
    MCG_K_COUNT = 10
};

const char* mcg_last_error(void);
const char* mcg_version(void);
int mcg_device_count(int* count);

/* ---- context ---------------------------------------------------------------------------- */
/* mcg_init: the ctx creates its own non-blocking stream.
 * mcg_init_on_stream: the ctx launches on the caller's hipStream_t (e.g. torch's current stream);
 * NULL there means the legacy default stream.  Use this form when a collective installed with
 * mcg_set_allreduce enqueues work on that same stream. */
int mcg_init(mcg_ctx** ctx, int device);
int mcg_init_on_stream(mcg_ctx** ctx, int device, void* stream);
int mcg_finalize(mcg_ctx* ctx);
int mcg_synchronize(mcg_ctx* ctx);
int mcg_trim(mcg_ctx* ctx); /* release cached device buffers */

/* Optional sum-all-reduce used when paths are sharded over several GPUs (one process per GPU).
 * fn must sum `count` doubles at device pointer `buf` in place over all ranks, ordered on
 * `stream` (a hipStream_t).  With a collective installed, mcg_price_european / mcg_price_lsm
 * return the GLOBAL price on every rank.  Payloads are 3 doubles (European) or 3p+2 doubles per
 * exercise date (LSM regression moments). */
typedef int (*mcg_allreduce_fn)(void* user, double* buf, int count, void* stream);
int mcg_set_allreduce(mcg_ctx* ctx, mcg_allreduce_fn fn, void* user);

/* Built-in RCCL collective (librccl is dlopen'ed on first use).  The 128-byte id comes from rank
 * 0's mcg_comm_unique_id and is broadcast by the launcher (torch.distributed store, MPI, ...). */
int mcg_comm_unique_id(unsigned char id[128]);
int mcg_comm_init_rank(mcg_ctx* ctx, const unsigned char id[128], int n_ranks, int rank);

/* Node-local collective over POSIX shared memory (one process per GPU, all on one host): `name` is a segment name
 * starting with '/', the same on every rank and unique to the job (rank 0 creates it, mcg_finalize removes it).
 * Installs a host all-reduce for the payoff sums AND lets the one-launch LSM sweeps exchange their per-date
 * regression moments between the GPUs INSIDE the kernel, through a device-mapped mailbox in the segment: a sharded
 * American price then costs one launch per GPU instead of three launches and one collective per exercise date.
 * At most 16 ranks. */
int mcg_comm_init_shm(mcg_ctx* ctx, const char* name, int n_ranks, int rank);
/* The ranks may also be THREADS of one process, one ctx each (one host thread per GPU -- the shape of the reference's own
 * OpenMP driver): every collective call then blocks until all of them have made it, so each rank needs its own thread. */

/* Opt-in, after mcg_comm_init_shm, collective over its ranks: keep the in-kernel mailbox in the GPUs' own HBM instead
 * of the host segment.  Every rank allocates a mailbox in device memory, exports it (hipIpcGetMemHandle, handed over
 * through the segment) and opens the peers' (hipIpcOpenMemHandle: on a multi-GPU node, peer memory over xGMI); a
 * reducing workgroup then PUSHES its moments into every peer's mailbox and polls local memory only.  Taken into use
 * only if every rank got through allocation, export, open and an in-kernel ping over the mappings; otherwise all
 * ranks stay on the host mailbox together (status MCG_OK, *active = 0).  enable = 0 goes back to the host mailbox.
 * The host segment keeps serving the barrier, the flags and the host all-reduce.
 * Lifetime: a rank's mailbox outlives every peer that maps it.  Between processes the IPC mapping sees to that by itself;
 * rank THREADS of one process use the owner's pointer as it stands, so the library counts them in the segment: a rank that
 * finalises (or switches back) while a rank thread still holds its mailbox leaves the mailbox to the LAST of them to let go,
 * which frees it -- nobody waits, nothing leaks, in whatever order the rank threads' contexts are closed. */
int mcg_comm_shm_peer_mailbox(mcg_ctx* ctx, int enable, int* active);

/* What collective this ctx holds and how many ranks it has SEEN: kind 0 none, 1 callback (mcg_set_allreduce),
 * 2 built-in RCCL, 3 node-local shared memory (host mailbox), 4 the same with the peer-memory mailbox active;
 * n_ranks / rank as given at set-up; seen_ranks = ncclCommCount of the communicator (kind 2) or the number of
 * processes attached to the segment (kinds 3, 4), 0 for a callback.  Any out pointer may be NULL. */
int mcg_comm_info(mcg_ctx* ctx, int* kind, int* n_ranks, int* rank, int* seen_ranks);

/* ---- path generation (replaces RoughVolatility.cpp:346-365, device side) ---------------- */
/* GBM: the stepping loop of RoughVolatility.cpp:354-364 with v == sigma^2.
 * Paths [path_begin, path_begin + n_paths) of the global Philox stream `seed`. */
int mcg_paths_gbm(mcg_ctx* ctx, uint64_t seed, double S0, double r, double sigma, double dt,
                  int n_steps, uint64_t path_begin, int64_t n_paths, mcg_paths** out);

/* rBergomi as the reference simulates it (RoughVolatility.cpp:342-364) with explicit parameters.
 * rho is accepted for interface parity; it does not change the law.
 * Paths are generated in pairs: path_begin must be even. */
int mcg_paths_rbergomi(mcg_ctx* ctx, uint64_t seed, double S0, double r, double xi, double H,
                       double eta, double rho, double dt, int n_steps, uint64_t path_begin,
                       int64_t n_paths, mcg_paths** out);

/* Fused GBM generation + terminal payoff reduction in one kernel (the measured headline path):
 * writes the full matrix AND leaves {sum payoff, sum payoff^2, n} for mcg_price_european to reuse
 * when called with the same (K, is_call). */
int mcg_paths_gbm_payoff(mcg_ctx* ctx, uint64_t seed, double S0, double r, double sigma, double dt,
                         int n_steps, uint64_t path_begin, int64_t n_paths, double K, int is_call,
                         mcg_paths** out);
int mcg_paths_rbergomi_payoff(mcg_ctx* ctx, uint64_t seed, double S0, double r, double xi, double H,
                              double eta, double rho, double dt, int n_steps, uint64_t path_begin,
                              int64_t n_paths, double K, int is_call, mcg_paths** out);

/* Upload a host matrix in the reference's layout: row_major[p*n_cols + j], n_cols = n_steps+1. */
int mcg_paths_from_host(mcg_ctx* ctx, const double* row_major, int64_t n_paths, int n_cols,
                        mcg_paths** out);
/* Download into the reference's layout (path-major, what GenerateStockPricePaths returns). */
int mcg_paths_to_host(const mcg_paths* paths, double* row_major_out);
/* Download as stored: out[j*n_paths + p]. */
int mcg_paths_to_host_step_major(const mcg_paths* paths, double* step_major_out);
int mcg_paths_info(const mcg_paths* paths, int64_t* n_paths, int* n_steps, int64_t* ld,
                   void** device_ptr);
int mcg_paths_free(mcg_paths* paths);

/* ---- pricing ---------------------------------------------------------------------------- */
/* e^{-rT} * mean(PayoffFunction(S_T)) (include/core/common.h:8-14 on the last column) and its
 * Monte Carlo standard error.  sums3 (optional) receives {sum, sum^2, n} before discounting. */
int mcg_price_european(mcg_ctx* ctx, const mcg_paths* paths, double K, double r, double T,
                       int is_call, double* mean, double* std_err);

/* LSM::PredictOptionPrice (src/models/LSMPricer.cpp:19-102) on a device-resident matrix.
 * Returns mean_i V[i][0]; std_err is an addition (the reference returns a bare mean).
 * poly_order in [0, 15] (orders above 8 take one launch per exercise date whatever the path count). */
int mcg_price_lsm(mcg_ctx* ctx, const mcg_paths* paths, double r, double K, double maturity,
                  double dt, int is_call, int poly_order, double* mean, double* std_err);

/* Whether this ctx currently uses the one-launch LSM sweep (one launch per price up to 8.37M paths per GPU, order <= 4):
 * it is switched off for the next eight LSM prices when the in-kernel hand-shake between workgroups times out
 * (another process holding part of the GPU); mcg_price_lsm answers those -- and the call that timed out -- from the
 * per-date kernels (one launch per exercise date). */
int mcg_lsm_one_launch_enabled(mcg_ctx* ctx, int* enabled);
/* Allow the one-launch sweep again at once (after a time-out it comes back by itself eight LSM prices later).
 * Sharded over mcg_comm_init_shm: call it on every rank or on none. */
int mcg_lsm_one_launch_reset(mcg_ctx* ctx);

/* AsymptoticAnalysis::PredictOptionPrice (src/models/AsymptoticAnalysisPricer.cpp:38-113) on a
 * device-resident matrix: mean over paths of the best discounted payoff among the dates (t <= maturity)
 * at which S lies beyond the short-time exercise boundary.  sigma <= 0 is MCG_ERR_INVALID with the
 * reference's message "AsymptoticAnalysis: Volatility must be positive."   */
int mcg_price_asymptotic(mcg_ctx* ctx, const mcg_paths* paths, double r, double K, double maturity,
                         double dt, int is_call, double sigma, double dividend, double* price);

/* MartingaleOptimization::PredictOptionPrice (src/models/MartingaleOptimizationPricer.cpp:21-189):
 * 0.5 * (primal + dual) after max_iterations iterations; lower/upper (optional) receive the two bounds.
 * poly_order in [0, 15]; max_iterations <= 0 is MCG_ERR_INVALID with the reference's message.
 *  */
int mcg_price_martingale(mcg_ctx* ctx, const mcg_paths* paths, double r, double K, double maturity,
                         double dt, int is_call, int poly_order, int max_iterations, double* price,
                         double* lower, double* upper);

/* BranchingProcesses::PredictOptionPrice (src/models/BranchingProcessPricer.cpp:12-134): midpoint of the
 * first-positive-payoff lower bound and the resampled-branch upper bound.  exercise_times are column indices
 * (the reference's driver passes 0..steps-1, PredictionGen.cpp:780-783).  The reference resamples with an
 * unseeded mt19937; here the resampling is Philox stream 2 of `seed`, so a call is reproducible.
 * Error messages are the reference's.   */
int mcg_price_branching(mcg_ctx* ctx, const mcg_paths* paths, double r, double K, double maturity,
                        double dt, int is_call, int num_branches, const int* exercise_times,
                        int n_exercise_times, uint64_t seed, double* price, double* lower, double* upper);

/* ---- batched driver rows ------------------------------------------- */
/* One option row of the reference's production caller (src/core/PredictionGen.cpp:566-791): path-engine
 * parameters (mcg_estimate_params of the row's spot history), contract terms and AsymptoticAnalysis inputs. */
typedef struct mcg_row {
    double S0, xi, H, eta, rho;                  /* RoughVolatility.cpp:327-331                          */
    double strike, maturity, sigma, dividend;    /* PredictionGen.cpp:701-709                            */
    int n_steps;                                 /* floor(maturity*252), :718                            */
    int is_call;
} mcg_row;

/* Prices n_rows option rows in six launches per chunk of rows (one chunk unless the rows' workspace exceeds the memory
 * budget, see below): n_paths (the driver uses 250) rBergomi paths per row,
 * then AsymptoticAnalysis, BranchingProcesses(num_branches, exercise dates 0..n_steps-1), LSM(poly_order) and
 * MartingaleOptimization(poly_order, max_iterations) on them.  out[4*i + {0,1,2,3}] = the four prices of row i
 * in the driver's column order (asymPrice, branchPrice, lsmPriceVal, martinPrice, :809-814).  Rows the driver
 * would answer with zeros (no steps, degenerate estimates, sigma <= 0, strike <= 0, an inf / nan among the row's generated
 * paths: :739-777 -- the row kernels scan every row's block for it, rows priced singly are scanned by a pass of their own) get zeros; every other row gets what its pricers
 * returned, finite or not (:809-816).
 * Row i uses Philox path ids (i << 32) + p of `seed`: its prices equal the single-contract entry points
 * called with path_begin = i << 32 -- and a row of more than 1020 steps (four years of trading days) IS priced through
 * them, after the batch, one row at a time -- as is every row of a call with n_paths > 256 or poly_order > 4 (the row
 * kernels' limits).  poly_order in [0, 15]. */
int mcg_batch_price_rows(mcg_ctx* ctx, const mcg_row* rows, int64_t n_rows, int n_paths, double r, double dt,
                         int num_branches, int poly_order, int max_iterations, uint64_t seed, double* out);
/* Any number of rows: the rows are processed in chunks whose workspace (every row's own (n_steps+1) x 256 block of the
 * path matrix, its amplitudes and compensator -- nothing is padded to the longest row) stays under a quarter of the
 * device memory that is free at the call; a row's Philox ids, and therefore its prices, do not depend on the chunking. */

/* The driver's two remaining feature columns (src/core/PredictionGen.cpp:313-347, compute20DayVolAndMomentum): annualised
 * standard deviation and sum of the last 20 log returns of the spot history; {0, 0} for fewer than 21 prices.
 * twenty_day_vol is also the `sigma` the driver hands to AsymptoticAnalysis (:706). */
int mcg_row_features(const double* hist, size_t n, double* twenty_day_vol, double* twenty_day_momentum);

/* One driver row from the driver's own inputs (PredictionGen.cpp:664-719): the spot history fetched for the row (the
 * driver appends underlying_last when it holds fewer than two prices, :671-673 -- so does this), the CSV fields
 * underlying_last, dte, strike_dist_pct, option_type (1 = call), dividend.  Fills *row (path-engine parameters by
 * mcg_estimate_params, strike = underlying_last (1 - strike_dist_pct), maturity = dte / 365, sigma = twenty_day_vol,
 * n_steps = floor(maturity 252)) and features2 = {twenty_day_vol, twenty_day_momentum}.  A row the driver answers with
 * ",0,0,0,0,0,0" (empty or non-finite history, inputs it rejects at :612-620) comes back with n_steps = 0 and zero
 * features, status MCG_OK: mcg_batch_price_rows* prices it to zeros like the driver. */
int mcg_row_build(const double* hist, size_t n, double underlying_last, double dte, double strike_dist_pct,
                  int option_type, double dividend, mcg_row* row, double features2[2]);

/* mcg_batch_price_rows with the driver's SIX output columns (:471-477, :809-816): out6[6*i + {0..3}] = the four model
 * prices, out6[6*i + {4,5}] = features2 of row i (mcg_row_build; NULL: zeros) -- except that a row whose pricing the
 * driver skips (n_steps < 1) keeps all six at zero, as the driver writes it. */
int mcg_batch_price_rows6(mcg_ctx* ctx, const mcg_row* rows, const double* features2, int64_t n_rows, int n_paths,
                          double r, double dt, int num_branches, int poly_order, int max_iterations, uint64_t seed,
                          double* out6);

/* ---- host-side pieces of the class-level API (a2/a3 of SURVEY.md section 8) --------------- */
/* RoughVolatility.cpp:324-331: out5 = {xi, H, eta, rho, S0}. */
int mcg_estimate_params(const double* hist, size_t n, double out5[5]);
/* Spectral amplitudes amp[0..Mz) and compensator comp[0..n_steps) staged in LDS by the rBergomi kernels
 * (DESIGN.md); Mz = nextpow2(n_steps). */
int mcg_rbergomi_spectrum(double H, double eta, double dt, int n_steps, double* amp, double* comp,
                          int* Mz);

/* The reference's class API through the C ABI (what the C++ shims in include/models call):
 * GenerateStockPricePaths(hist, steps, paths) -> out[paths][steps+1], and
 * LSM::PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, polyOrder).
 * Both use a lazily created per-thread ctx on device 0 (MCG_DEVICE overrides). */
int mcg_compat_set_seed(uint64_t seed, int enabled); /* default: std::random_device per call */
/* Calls of the class API that arrive from DIFFERENT host threads while a round of them is on the device are answered
 * together: one launch per kind of call over all of them (the row kernels of mcg_batch_price_rows, one workgroup per
 * matrix), the reference's driver unchanged (src/core/PredictionGen.cpp:542-570, :736-737, :788-791: one row per OpenMP
 * thread, five calls per row).  Shapes the row kernels serve -- at most 256 paths, 1 .. 1020 steps, polynomial order <= 4,
 * BranchingProcesses with the driver's exercise dates 0 .. steps - 1; anything else, or everything after
 * mcg_compat_set_coalescing(0), runs on the calling thread's own context as before.  Mode 1 (the default) also PREFETCHES: the
 * first pricer call on a matrix the library already holds on the device queues the driver's other three pricers with the driver's
 * arguments (:788-791: the same r, strike, maturity, dt, isCall; 10 branches, order 2, 5 iterations), each in the lane of its kind,
 * so that the four run side by side; a later call is answered from that only if it asks for exactly what was computed on exactly
 * that matrix.  Mode 2: coalescing without the prefetch.  A lone caller is a round of one.  mcg_stats counts rounds, calls,
 * prefetches, hits and fall-backs.
 * Resources: the first class-API call that takes this route creates five contexts on device MCG_DEVICE (default 0) and starts
 * five service threads inside the library (one per kind of call; they sleep while nothing is queued and are joined by an atexit
 * handler before the HIP runtime shuts down); every calling thread gets a 2 MB slot of device memory (allocated 32 slots at a
 * time) and a pinned host buffer of its matrix's size, both released when the thread ends.  A process that forks must do so
 * before its first call, like any user of the HIP runtime. */
int mcg_compat_set_coalescing(int mode);
int mcg_compat_generate_paths(const double* hist, size_t n, int forward_steps, int path_num,
                              double* row_major_out);
int mcg_compat_lsm_price(const double* row_major, int64_t n_paths, int n_cols, double r,
                         double strike, double maturity, double dt, int is_call, int poly_order,
                         double* price);
/* AsymptoticAnalysis::PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, sigma, dividend);
 * like the reference, empty or ragged input prices to 0.0 (status MCG_OK). */
/* MartingaleOptimization::PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, polyOrder, maxIterations) */
int mcg_compat_martingale_price(const double* row_major, int64_t n_paths, int n_cols, double r,
                                double strike, double maturity, double dt, int is_call, int poly_order,
                                int max_iterations, double* price);
/* BranchingProcesses::PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, numBranches, exerciseTimes) */
int mcg_compat_branching_price(const double* row_major, int64_t n_paths, int n_cols, double r,
                               double strike, double maturity, double dt, int is_call, int num_branches,
                               const int* exercise_times, int n_exercise_times, double* price);
int mcg_compat_asymptotic_price(const double* row_major, int64_t n_paths, int n_cols, double r,
                                double strike, double maturity, double dt, int is_call, double sigma,
                                double dividend, double* price);

/* ---- measurement ------------------------------------------------------------------------ */
/* When enabled, every kernel launch is bracketed by HIP events on the ctx stream. */
int mcg_timing_enable(mcg_ctx* ctx, int on);
/* Which kernels are bracketed while timing is enabled: bit k = enum mcg_kernel k (default: all).  An event pair costs
 * several microseconds on the stream, so a measurement of one kernel's duration inside a timed loop selects that one. */
int mcg_timing_select(mcg_ctx* ctx, unsigned mask);
int mcg_timing_reset(mcg_ctx* ctx);
int mcg_timing_get(mcg_ctx* ctx, int kernel /* enum mcg_kernel */, double* total_ms,
                   int64_t* launches);

/* What this board writes right now with the path matrix's store pattern and NO arithmetic: `reps` timed launches (after
 * two untimed ones) of a kernel that stores an n_paths x (n_steps+1) fp64 matrix exactly as the GBM generator does (two
 * adjacent paths per lane, one nontemporal 16-byte store per step, rows n_paths apart).  The ceiling the generator's
 * achieved GB/s is set against, measured in the same process on the same board (boards differ by ~10 %). */
int mcg_probe_write_ceiling(mcg_ctx* ctx, int64_t n_paths, int n_steps, int reps, double* gb_per_s, double* ms_per_launch);
/* Shader clock of a GBM generator launch, stamped inside the kernel by a few workgroups spread over the grid (s_memtime /
 * s_memrealtime around each one's whole life).  A measurement aid, OFF by default: mcg_generator_clock_arm(ctx, 1) makes the
 * NEXT launches of the GBM generator on this ctx stamp (one memset of the 1 KiB stamp buffer ahead of each, a few scalar
 * reads in ~40 workgroups), ..._arm(ctx, 0) ends it; launches that are not armed pass no stamp buffer and queue nothing
 * extra.  mcg_generator_clock returns the median in GHz, the number of stamps (0: the last launch was not armed, or too
 * small to stamp) and the lowest / highest.  The generator is power-limited; its clock under load is what separates boards. */
int mcg_generator_clock_arm(mcg_ctx* ctx, int on);
int mcg_generator_clock(mcg_ctx* ctx, double* ghz_median, int* n_stamps, double* ghz_min, double* ghz_max);

/* Process-wide event counters (all contexts, all threads): what ran and what fell back. */
typedef struct mcg_stats_t {
    int64_t lsm_one_launch_sweeps;     /* LSM prices answered by ONE launch (k_lsm_coop / k_lsm_big)                      */
    int64_t lsm_one_launch_timeouts;   /* one-launch sweeps whose hand-shake gave up: discarded, re-run on the per-date route */
    int64_t lsm_per_date_sweeps;       /* LSM prices answered by the per-date route                                       */
    int64_t lsm_per_date_launches;     /* k_lsm_date launches queued for them (one per column + one per re-fitted date)   */
    int64_t lsm_per_date_refits;       /* ... of which second launches of a re-fitted date                                */
    int64_t lsm_per_date_faults;       /* per-date sweeps ended because partial moments did not arrive (MCG_ERR_HIP)      */
    int64_t shm_barrier_failures;      /* barriers of the node segment that timed out or found it poisoned                */
    int64_t peer_mailbox_enabled;      /* mcg_comm_shm_peer_mailbox calls that ended with the mailbox in peer memory      */
    int64_t peer_mailbox_refused;      /* ... that left all ranks on the host mailbox (export, open or ping failed)       */
    int64_t batch_calls, batch_chunks; /* mcg_batch_price_rows*: calls, and the chunks they were processed in             */
    int64_t batch_rows;                /* rows priced by the row kernels                                                   */
    int64_t batch_rows_singly;         /* rows priced one by one through the single-contract entry points                 */
    int64_t batch_peak_workspace_bytes;/* largest device workspace a chunk has used                                        */
    int64_t peer_mailbox_kept;         /* peer-memory mailboxes left at release to the last same-process rank thread that held them (it frees them) */
    int64_t coalesced_rounds;          /* class-API calls of several host threads answered together: rounds (one set of launches each) */
    int64_t coalesced_calls;           /* ... and the calls they answered                                                  */
    int64_t coalesced_peak_calls_per_round; /* most calls one round has answered                                           */
    int64_t coalesced_fallbacks;       /* class-API calls that took the calling thread's own context instead (shape beyond the row kernels, coalescing off) */
    int64_t coalesced_round_us;        /* wall time of the rounds, summed (packing + upload + launches + synchronisation), microseconds */
    int64_t coalesced_device_wait_us;  /* ... of which inside hipStreamSynchronize                                          */
    int64_t coalesced_wake_us;         /* time the lanes' service threads spent waking the callers they had answered         */
    int64_t coalesced_prefetched;      /* pricer calls made ahead of the caller asking (the other pricers of a row, with the driver's arguments) */
    int64_t coalesced_prefetch_hits;   /* ... whose answer the caller then took (its call matched): no device round trip of its own */
} mcg_stats_t;
int mcg_stats(mcg_stats_t* out, int reset);

/* (Test hooks -- mcg_debug_* -- are declared in mcgpu_debug.h; nothing a caller of the product needs.) */

#ifdef __cplusplus
}
#endif
#endif /* MCGPU_H */
