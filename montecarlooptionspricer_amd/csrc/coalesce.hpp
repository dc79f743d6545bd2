// Cross-thread coalescing of the reference's class API.
//
// The reference's driver prices one option row per OpenMP thread and builds its five objects per row
// (/root/reference/src/core/PredictionGen.cpp:542-570): RoughVolatility::GenerateStockPricePaths (:736-737), then
// AsymptoticAnalysis, BranchingProcesses, LSM, MartingaleOptimization in turn (:788-791).  Through the drop-in classes every
// one of those five calls used to be a launch + a host synchronisation on the calling thread's own stream: ~0.7 ms per
// row and thread, 880 rows/s at 128 threads (the HIP runtime serialises them), against 2.6 M rows/s for mcg_batch_price_rows --
// which needs the driver's loop rewritten.  Here calls that arrive from DIFFERENT host threads while one round is on the device
// are answered together: each kind of call has a lane (queue + service thread + stream); the service thread issues ONE launch
// per group of queued calls -- the row kernels of kernels_batch.hip, one workgroup per row -- and hands the results back; a lone
// caller is a round of one.  No source change in the driver.
//
//   host/coalesce.cpp   the combiner: lanes, service threads, per-thread matrix slots and pinned buffers, prefetched calls
//   host/dropin.cpp     co_price: which calls take this route, and the prefetch of a row's other pricers
//   kernels_batch.hip   co::execute_round: one round = one upload of the round's descriptors, its launches and one
//                       synchronisation; paths and prices come back through device-visible pinned host memory
#pragma once
#include <cstddef>
#include <cstdint>

struct mcg_ctx;

namespace mcg {
namespace co {

constexpr int MAX_PATHS = 256;   // columns of a row's matrix block (kernels_batch.hip: BATCH_LD)
constexpr int MAX_STEPS = 1020;  // steps the row kernels serve (BATCH_MAX_STEPS)
constexpr size_t SLOT_DOUBLES = (size_t)MAX_PATHS * (size_t)(MAX_STEPS + 1);  // one thread's resident matrix block

enum Kind { GEN = 0, ASYM = 1, BRANCH = 2, LSM = 3, MART = 4, N_KINDS = 5 };

// One call of one thread.  The requester fills it, queues it and waits; the service thread of its lane reads the inputs and
// writes price / status / err.  Everything it points to belongs to the requester and stays
// alive until done.
struct Request {
    int kind = GEN;
    // the matrix: [n_paths][n_steps + 1] path-major in `host` (the requester's pinned, device-visible buffer) <-> the
    // requester's slot on the device, step-major with 256 columns ((double*)arena_base + slot_off)
    int n_paths = 0, n_steps = 0;
    int64_t slot_off = 0;
    double* host = nullptr;
    double* host_dev = nullptr;  // the same buffer as the device addresses it (hipHostGetDevicePointer)
    bool upload = false;  // pricers: the slot does not hold this matrix yet -- fill it from `host` first
    // GEN (RoughVolatility.cpp:312-368 with the estimates made on the requester's thread)
    double S0 = 0, xi = 0, H = 0, eta = 0;
    uint64_t seed = 0;
    const double* amp = nullptr;   // [M] spectral amplitudes (host/volterra.cpp), M = next power of two >= n_steps
    const double* comp = nullptr;  // [n_steps] compensator
    int M = 0;
    // pricers
    double r = 0, strike = 0, maturity = 0, dt = 0, sigma = 0, dividend = 0;
    int is_call = 0, poly_order = 2, num_branches = 10, max_iterations = 5;
    // answer
    double price = 0.0;
    int status = 0;
    char err[192] = {0};
};

// Pinned host staging of one round (allocated once, grown on demand) and its device mirror.
struct RoundBuffers {
    unsigned char* h = nullptr;   // pinned: the round's descriptors as uploaded
    unsigned char* d = nullptr;   // device copy
    size_t cap = 0;
    double* h_out = nullptr;      // pinned, device-visible: [max_requests][4] prices as the kernels write them
    double* d_out = nullptr;      // ... as the device addresses it
    double* d_scratch = nullptr;  // device: per-request flags nobody reads back
    size_t out_cap = 0;           // requests
};

// One round on ctx's stream: every request answered (status / price / err set).  arena_base: what slot_off is relative to.
// Returns 0 or the status of a failure that hit the whole round (then every request carries it).
int execute_round(mcg_ctx* ctx, RoundBuffers& rb, double* arena_base, Request** reqs, int n);

}  // namespace co
}  // namespace mcg
