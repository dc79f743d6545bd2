// The reference's class API (include/mcgpu/dropin.hpp) on top of the C ABI, plus the
// mcg_compat_* entry points that expose the same two calls to non-C++ hosts and to the tests.
//
//   RoughVolatility::GenerateStockPricePaths  <->  /root/reference/src/models/RoughVolatility.cpp:312-368
//   LSM::PredictOptionPrice                   <->  /root/reference/src/models/LSMPricer.cpp:19-102
//   AsymptoticAnalysis::PredictOptionPrice    <->  /root/reference/src/models/AsymptoticAnalysisPricer.cpp:38-113
//   MartingaleOptimization::PredictOptionPrice <-> /root/reference/src/models/MartingaleOptimizationPricer.cpp:21-189
//   BranchingProcesses::PredictOptionPrice     <-> /root/reference/src/models/BranchingProcessPricer.cpp:12-134
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mcgpu/dropin.hpp"
#include "../csrc/mcg_internal.hpp"
#include "coalesce_host.hpp"

namespace {

// One lazily created context per host thread: the reference driver builds its pricer objects per
// row inside an OpenMP region (PredictionGen.cpp:542-570), so construction must be free and calls
// re-entrant.  The holder tears the ctx down when the thread exits.
struct ThreadCtx {
    mcg_ctx* ctx = nullptr;
    // The reference's driver hands the SAME host matrix to four pricers in a row (PredictionGen.cpp:788-791).
    // The last matrix this thread generated or uploaded stays on the device together with a flat host copy of its
    // contents; a later call reuses the device copy only after comparing EVERY element of the caller's matrix with
    // that host copy (memcmp, row by row: microseconds for the driver's 250 x T rows), so a different matrix can
    // never be priced by mistake.  Matrices above CACHE_MAX_BYTES are not cached (the copy would double their
    // host footprint).
    static constexpr size_t CACHE_MAX_BYTES = 64u << 20;
    mcg_paths* cached = nullptr;
    size_t cached_n = 0, cached_m = 0;
    std::vector<double> cached_flat;  // [cached_n][cached_m]
    ~ThreadCtx() {
        if (cached) mcg_paths_free(cached);
        if (ctx) mcg_finalize(ctx);
    }
    void forget() {
        if (cached) mcg_paths_free(cached);
        cached = nullptr;
        cached_n = cached_m = 0;
        cached_flat.clear();
    }
    // takes ownership of p and of the flat copy of its contents
    void remember(mcg_paths* p, size_t n, size_t m, std::vector<double>&& flat) {
        if (cached && cached != p) mcg_paths_free(cached);
        cached = p;
        cached_n = n;
        cached_m = m;
        cached_flat = std::move(flat);
    }
    bool holds(const std::vector<std::vector<double>>& rows, size_t m) const {
        if (!cached || cached_n != rows.size() || cached_m != m) return false;
        for (size_t i = 0; i < rows.size(); ++i)
            if (std::memcmp(rows[i].data(), cached_flat.data() + i * m, m * sizeof(double)) != 0) return false;
        return true;
    }
    mcg_ctx* get() {
        if (!ctx) {
            int dev = 0;
            if (const char* e = std::getenv("MCG_DEVICE")) dev = std::atoi(e);
            if (mcg_init(&ctx, dev) != MCG_OK) throw std::runtime_error(mcg_last_error());
        }
        return ctx;
    }
};
thread_local ThreadCtx t_ctx;

// Calls of different host threads are answered together (csrc/coalesce.hpp) unless mcg_compat_set_coalescing(0) said otherwise;
// mode 1 (the default) also makes a row's other pricer calls AHEAD of the driver asking for them (co_price), mode 2 does not.
std::atomic<int> g_coalesce{1};

std::atomic<bool> g_seed_fixed{false};
std::atomic<uint64_t> g_seed{0};

uint64_t next_seed() {
    if (g_seed_fixed.load()) return g_seed.load();
    // the reference seeds from std::random_device on every call (RoughVolatility.cpp:239, :253)
    std::random_device rd;
    return ((uint64_t)rd() << 32) | rd();
}

[[noreturn]] void raise_last() { throw std::runtime_error(mcg_last_error()); }

struct PathsGuard {
    mcg_paths* p = nullptr;
    ~PathsGuard() { mcg_paths_free(p); }
    mcg_paths* release() {
        mcg_paths* q = p;
        p = nullptr;
        return q;
    }
};

// Device copy of a host path matrix: the thread's cached one when the contents are identical (ThreadCtx::holds),
// otherwise a fresh upload (which becomes the cached one when it is small enough to keep a host copy of).
// The returned handle is owned by the thread's cache or, for an uncached matrix, by `own` (freed by the caller's
// guard).  `ragged_msg` is the error message for a row shorter than the first.
mcg_paths* device_matrix(const std::vector<std::vector<double>>& pricePaths, const char* ragged_msg, PathsGuard& own) {
    const size_t N = pricePaths.size(), M = pricePaths[0].size();
    for (const auto& row : pricePaths)
        if (row.size() < M) throw std::runtime_error(ragged_msg);
    mcg_ctx* ctx = t_ctx.get();
    if (t_ctx.holds(pricePaths, M)) return t_ctx.cached;
    std::vector<double> flat(N * M);
    for (size_t i = 0; i < N; ++i) std::copy(pricePaths[i].begin(), pricePaths[i].begin() + M, flat.begin() + i * M);
    PathsGuard g;
    if (mcg_paths_from_host(ctx, flat.data(), (int64_t)N, (int)M, &g.p) != MCG_OK) raise_last();
    if (flat.size() * sizeof(double) <= ThreadCtx::CACHE_MAX_BYTES) {
        t_ctx.remember(g.p, N, M, std::move(flat));
        return g.release();
    }
    t_ctx.forget();
    own.p = g.release();
    return own.p;
}

// ---- the coalesced route (csrc/coalesce.hpp): shapes the row kernels serve -- at most 256 paths, 1..1020 steps ----------
bool co_shape(size_t n_paths, size_t n_cols) {
    return g_coalesce.load(std::memory_order_relaxed) != 0 && n_paths >= 1 && n_paths <= (size_t)mcg::co::MAX_PATHS && n_cols >= 2 &&
           n_cols <= (size_t)mcg::co::MAX_STEPS + 1 && mcg::co::thread_state().have_slot();   // (no slot left, no device: the own-context route says why)
}

void took_own_context() { mcg::g_stats.coalesced_fallbacks.fetch_add(1, std::memory_order_relaxed); }

// Do two requests ask the same pricer for the same thing?  (MartingaleOptimization: every iteration count >= 2 is the same
// computation, kernels_martingale.hip.)
bool same_call(const mcg::co::Request& a, const mcg::co::Request& b) {
    if (a.kind != b.kind || a.r != b.r || a.strike != b.strike || a.maturity != b.maturity || a.dt != b.dt || a.is_call != b.is_call) return false;
    switch (a.kind) {
        case mcg::co::BRANCH: return a.num_branches == b.num_branches;  // (the resampling seed is drawn with the call that is made first)
        case mcg::co::LSM: return a.poly_order == b.poly_order;
        case mcg::co::MART: return a.poly_order == b.poly_order && (a.max_iterations >= 2) == (b.max_iterations >= 2);
        default: return a.sigma == b.sigma && a.dividend == b.dividend;
    }
}

// A pricer call through the combiner: the thread's slot when it holds exactly this matrix (compared element by element),
// otherwise the matrix goes up with the request.  q carries the pricer's own arguments.
//
// PREFETCH.  The reference's driver calls its four pricers one after another on the matrix it has just generated, with the same
// (r, strike, maturity, dt, isCall) and its literals for the rest: 10 branches over the dates 0 .. steps - 1, polynomial order 2,
// five iterations (PredictionGen.cpp:788-791).  So the FIRST pricer call on a matrix the slot already holds also queues the
// other three with those arguments, each in the lane of its kind, without waiting for them: the four row kernels run side by
// side, and when the driver asks for the next price the answer is there or on its way -- a row costs two waits (its paths, its
// slowest pricer) instead of five.  An answer is handed out only to a call that asks for exactly what was computed (same_call) on
// exactly the matrix it was computed on (the slot's element-by-element comparison); any other call is made in the ordinary way.
// The resampling seed of BranchingProcesses is drawn when its request is queued -- "a fresh seed per call", as before.
double co_price(const std::vector<std::vector<double>>& pricePaths, const char* ragged_msg, mcg::co::Request& q) {
    const size_t N = pricePaths.size(), M = pricePaths[0].size();
    for (const auto& row : pricePaths)
        if (row.size() < M) throw std::runtime_error(ragged_msg);
    mcg::co::ThreadState& t = mcg::co::thread_state();
    q.n_paths = (int)N;
    q.n_steps = (int)M - 1;
    q.upload = false;
    if (t.holds(pricePaths, M)) {
        double price;
        // (BranchingProcesses: under mcg_compat_set_seed the answer must be the one THIS seed gives; unseeded, the seed drawn when the
        //  request was queued is as fresh as one drawn now)
        const bool seed_ok = q.kind != mcg::co::BRANCH || !g_seed_fixed.load() || t.ahead[q.kind].req.seed == q.seed;
        if (seed_ok && same_call(t.ahead[q.kind].req, q) && t.take_prefetched(q.kind, &price)) return price;
        if (!t.prefetched_for_this_matrix && g_coalesce.load(std::memory_order_relaxed) == 1 && q.strike > 0.0 && q.dt > 0.0) {
            t.prefetched_for_this_matrix = true;
            for (int kind : {mcg::co::BRANCH, mcg::co::LSM, mcg::co::MART}) {
                if (kind == q.kind) continue;
                mcg::co::Request a;
                a.kind = kind;
                a.n_paths = q.n_paths;
                a.n_steps = q.n_steps;
                a.r = q.r;
                a.strike = q.strike;
                a.maturity = q.maturity;
                a.dt = q.dt;
                a.is_call = q.is_call;
                a.num_branches = 10;   // PredictionGen.cpp:789
                a.poly_order = 2;      // :790-791
                a.max_iterations = 5;  // MartingaleOptimizationPricer.h: maxIterations = 5
                if (kind == mcg::co::BRANCH) a.seed = next_seed();
                t.prefetch(a);
            }
        }
    } else {
        if (t.prepare((int)N, (int)M) != MCG_OK) raise_last();   // (waits for whatever is still in flight on the old matrix)
        for (size_t i = 0; i < N; ++i) std::memcpy(t.pinned + i * M, pricePaths[i].data(), M * sizeof(double));
        q.upload = true;
    }
    if (t.submit(q) != MCG_OK) raise_last();
    return q.price;
}

}  // namespace

RoughVolatility::RoughVolatility() {}

std::vector<std::vector<double>> RoughVolatility::GenerateStockPricePaths(
    const std::vector<double>& historical_prices, int forward_steps, int path_num) {
    if (historical_prices.size() < 2) throw std::runtime_error("Historical prices vector too small.");
    if (forward_steps < 0 || path_num < 0)
        throw std::length_error("RoughVolatility: negative forward_steps or path_num");

    double p[5];
    if (mcg_estimate_params(historical_prices.data(), historical_prices.size(), p) != MCG_OK) raise_last();
    const double xi = p[0], H = p[1], eta = p[2], rho = p[3], S0 = p[4];
    const double r = 0.04, dt = 1.0 / 252.0;  // RoughVolatility.cpp:321-326

    std::vector<std::vector<double>> paths((size_t)path_num, std::vector<double>((size_t)forward_steps + 1, 0.0));
    if (path_num == 0) return paths;
    for (auto& row : paths) row[0] = S0;
    if (forward_steps == 0) return paths;

    // Degenerate estimates: the reference's arithmetic turns every step into NaN when rho is NaN
    // (two-point history: 0/0 at :164) or when lambda is non-finite (H < 0 at t = 0, :233).
    if (!(std::fabs(rho) <= 1.0) || !(H >= 0.0) || !std::isfinite(xi) || !std::isfinite(eta) || !(xi >= 0.0)) {
        for (auto& row : paths)
            for (size_t j = 1; j < row.size(); ++j) row[j] = std::numeric_limits<double>::quiet_NaN();
        return paths;
    }

    const size_t cols = (size_t)forward_steps + 1;
    if (co_shape((size_t)path_num, cols) && S0 > 0.0) {
        // Coalesced: the estimates and the spectrum are this thread's work; the generator launch, shared with whatever other
        // threads are asking at the moment, is the round leader's.  Same kernel code, same Philox ids (seed, paths 0 ..
        // path_num - 1) and the same host-made amplitudes as mcg_paths_rbergomi: the same matrix bit for bit.
        std::vector<double> amp, comp;
        if (mcg::host_rbergomi_spectrum(H, eta, dt, forward_steps, amp, comp) != MCG_OK) raise_last();
        mcg::co::ThreadState& t = mcg::co::thread_state();
        if (t.prepare(path_num, (int)cols) != MCG_OK) raise_last();
        mcg::co::Request q;
        q.kind = mcg::co::GEN;
        q.n_paths = path_num;
        q.n_steps = forward_steps;
        q.S0 = S0;
        q.xi = xi;
        q.H = H;
        q.eta = eta;
        q.seed = next_seed();
        q.amp = amp.data();
        q.comp = comp.data();
        q.M = (int)amp.size();
        if (t.submit(q) != MCG_OK) raise_last();
        for (size_t i = 0; i < (size_t)path_num; ++i) paths[i].assign(t.pinned + i * cols, t.pinned + (i + 1) * cols);
        return paths;  // (the slot keeps the matrix for the pricers that come next, PredictionGen.cpp:788-791)
    }
    took_own_context();
    mcg_ctx* ctx = t_ctx.get();
    PathsGuard g;
    if (mcg_paths_rbergomi(ctx, next_seed(), S0, r, xi, H, eta, rho, dt, forward_steps, 0, path_num, &g.p) != MCG_OK)
        raise_last();
    std::vector<double> flat((size_t)path_num * ((size_t)forward_steps + 1));
    if (mcg_paths_to_host(g.p, flat.data()) != MCG_OK) raise_last();
    for (size_t i = 0; i < (size_t)path_num; ++i) paths[i].assign(flat.begin() + i * cols, flat.begin() + (i + 1) * cols);
    if (flat.size() * sizeof(double) <= ThreadCtx::CACHE_MAX_BYTES)
        t_ctx.remember(g.release(), (size_t)path_num, cols, std::move(flat));  // the pricers come next
    return paths;
}

double LSM::PredictOptionPrice(const std::vector<std::vector<double>>& pricePaths, double r, double strike,
                               double maturity, double dt, bool isCall, int polyOrder) {
    if (pricePaths.empty() || pricePaths[0].empty())
        throw std::runtime_error("LSM::PredictOptionPrice: Empty pricePaths.");
    // (raw monomials beyond S^15 at S ~ 100 span more than thirty orders of magnitude: the reference's own solve keeps a
    // handful of singular directions of them; orders up to 15 are served and checked against the restated reference)
    if (polyOrder < 0 || polyOrder > 15) throw std::invalid_argument("LSM: polyOrder must be in [0, 15]");
    if (polyOrder <= 4 && dt > 0.0 && co_shape(pricePaths.size(), pricePaths[0].size())) {
        mcg::co::Request q;
        q.kind = mcg::co::LSM;
        q.r = r;
        q.strike = strike;
        q.maturity = maturity;
        q.dt = dt;
        q.is_call = isCall ? 1 : 0;
        q.poly_order = polyOrder;
        return co_price(pricePaths, "LSM: Invalid path index in regression", q);
    }
    took_own_context();
    PathsGuard own;
    mcg_paths* P = device_matrix(pricePaths, "LSM: Invalid path index in regression", own);
    double price = 0.0;
    if (mcg_price_lsm(t_ctx.get(), P, r, strike, maturity, dt, isCall ? 1 : 0, polyOrder, &price, nullptr) != MCG_OK)
        raise_last();
    return price;
}

double AsymptoticAnalysis::PredictOptionPrice(const std::vector<std::vector<double>>& pricePaths, double r,
                                              double strike, double maturity, double dt, bool isCall, double sigma,
                                              double dividend) {
    if (pricePaths.empty() || pricePaths[0].empty()) return 0.0;                     // :47-49
    if (sigma <= 0.0) throw std::runtime_error("AsymptoticAnalysis: Volatility must be positive.");  // :50-52
    const size_t M = pricePaths[0].size();
    for (const auto& row : pricePaths)
        if (row.size() != M) return 0.0;                                               // :57-61
    try {
        if (co_shape(pricePaths.size(), M)) {
            mcg::co::Request q;
            q.kind = mcg::co::ASYM;
            q.r = r;
            q.strike = strike;
            q.maturity = maturity;
            q.dt = dt;
            q.is_call = isCall ? 1 : 0;
            q.sigma = sigma;
            q.dividend = dividend;
            return co_price(pricePaths, "AsymptoticAnalysis: ragged pricePaths.", q);
        }
        took_own_context();
        PathsGuard own;
        mcg_paths* P = device_matrix(pricePaths, "AsymptoticAnalysis: ragged pricePaths.", own);
        double price = 0.0;
        if (mcg_price_asymptotic(t_ctx.get(), P, r, strike, maturity, dt, isCall ? 1 : 0, sigma, dividend, &price) != MCG_OK)
            raise_last();
        return price;
    } catch (const std::bad_alloc&) {
        return 0.0;  // the reference swallows every exception of its main loop and returns 0 (:110-112)
    }
}

double MartingaleOptimization::PredictOptionPrice(const std::vector<std::vector<double>>& pricePaths, double r,
                                                  double strike, double maturity, double dt, bool isCall, int polyOrder,
                                                  int maxIterations) {
    if (pricePaths.empty() || pricePaths[0].empty())
        throw std::runtime_error("MartingaleOptimization: Empty pricePaths.");                      // :31-33
    if (maxIterations <= 0) throw std::runtime_error("MartingaleOptimization: maxIterations must be positive.");  // :34-36
    if (polyOrder < 0 || polyOrder > 15) throw std::invalid_argument("MartingaleOptimization: polyOrder must be in [0, 15]");
    if (polyOrder <= 4 && strike != 0.0 && co_shape(pricePaths.size(), pricePaths[0].size())) {
        mcg::co::Request q;
        q.kind = mcg::co::MART;
        q.r = r;
        q.strike = strike;
        q.maturity = maturity;
        q.dt = dt;
        q.is_call = isCall ? 1 : 0;
        q.poly_order = polyOrder;
        q.max_iterations = maxIterations;
        return co_price(pricePaths, "MartingaleOptimization: ragged pricePaths.", q);
    }
    took_own_context();
    PathsGuard own;
    mcg_paths* P = device_matrix(pricePaths, "MartingaleOptimization: ragged pricePaths.", own);
    double price = 0.0;
    if (mcg_price_martingale(t_ctx.get(), P, r, strike, maturity, dt, isCall ? 1 : 0, polyOrder, maxIterations, &price, nullptr,
                             nullptr) != MCG_OK)
        raise_last();
    return price;
}

double BranchingProcesses::PredictOptionPrice(const std::vector<std::vector<double>>& pricePaths, double r,
                                              double strike, double maturity, double dt, bool isCall, int numBranches,
                                              const std::vector<int>& exerciseTimes) {
    if (pricePaths.empty() || pricePaths[0].empty()) throw std::runtime_error("BranchingProcesses: Empty pricePaths.");
    if (exerciseTimes.empty()) throw std::runtime_error("BranchingProcesses: No exercise times.");
    if (strike <= 0.0) throw std::runtime_error("BranchingProcesses: Strike must be positive.");
    // the row kernel walks the driver's exercise dates 0 .. steps - 1 (PredictionGen.cpp:780-783); any other list takes the
    // single-contract kernels on this thread's own context
    bool driver_dates = exerciseTimes.size() + 1 == pricePaths[0].size() && numBranches >= 0 && numBranches <= 1024;
    for (size_t i = 0; driver_dates && i < exerciseTimes.size(); ++i) driver_dates = exerciseTimes[i] == (int)i;
    if (driver_dates && co_shape(pricePaths.size(), pricePaths[0].size())) {
        mcg::co::Request q;
        q.kind = mcg::co::BRANCH;
        q.r = r;
        q.strike = strike;
        q.maturity = maturity;
        q.dt = dt;
        q.is_call = isCall ? 1 : 0;
        q.num_branches = numBranches;
        q.seed = next_seed();
        return co_price(pricePaths, "BranchingProcesses: ragged pricePaths.", q);
    }
    took_own_context();
    PathsGuard own;
    mcg_paths* P = device_matrix(pricePaths, "BranchingProcesses: ragged pricePaths.", own);
    double price = 0.0;
    if (mcg_price_branching(t_ctx.get(), P, r, strike, maturity, dt, isCall ? 1 : 0, numBranches, exerciseTimes.data(),
                            (int)exerciseTimes.size(), next_seed(), &price, nullptr, nullptr) != MCG_OK)
        raise_last();
    return price;
}

extern "C" {

int mcg_compat_branching_price(const double* row_major, int64_t n_paths, int n_cols, double r, double strike,
                               double maturity, double dt, int is_call, int num_branches, const int* exercise_times,
                               int n_exercise_times, double* price) {
    try {
        std::vector<std::vector<double>> m;
        if (row_major && n_paths > 0 && n_cols > 0) {
            m.resize((size_t)n_paths);
            for (int64_t i = 0; i < n_paths; ++i) m[i].assign(row_major + i * n_cols, row_major + (i + 1) * n_cols);
        }
        std::vector<int> ex;
        if (exercise_times && n_exercise_times > 0) ex.assign(exercise_times, exercise_times + n_exercise_times);
        BranchingProcesses bp;
        const double v = bp.PredictOptionPrice(m, r, strike, maturity, dt, is_call != 0, num_branches, ex);
        if (price) *price = v;
        return MCG_OK;
    } catch (const std::exception& e) {
        const std::string msg = e.what();
        const int code = msg == "BranchingProcesses: Empty pricePaths." ? MCG_ERR_EMPTY_PATHS : MCG_ERR_INVALID;
        return mcg::fail(code, "%s", msg.c_str());
    }
}

int mcg_compat_martingale_price(const double* row_major, int64_t n_paths, int n_cols, double r, double strike,
                                double maturity, double dt, int is_call, int poly_order, int max_iterations,
                                double* price) {
    try {
        std::vector<std::vector<double>> m;
        if (row_major && n_paths > 0 && n_cols > 0) {
            m.resize((size_t)n_paths);
            for (int64_t i = 0; i < n_paths; ++i) m[i].assign(row_major + i * n_cols, row_major + (i + 1) * n_cols);
        }
        MartingaleOptimization mo;
        const double v = mo.PredictOptionPrice(m, r, strike, maturity, dt, is_call != 0, poly_order, max_iterations);
        if (price) *price = v;
        return MCG_OK;
    } catch (const std::exception& e) {
        const std::string msg = e.what();
        const int code = msg == "MartingaleOptimization: Empty pricePaths." ? MCG_ERR_EMPTY_PATHS : MCG_ERR_INVALID;
        return mcg::fail(code, "%s", msg.c_str());
    }
}

int mcg_compat_asymptotic_price(const double* row_major, int64_t n_paths, int n_cols, double r, double strike,
                                double maturity, double dt, int is_call, double sigma, double dividend,
                                double* price) {
    try {
        std::vector<std::vector<double>> m;
        if (row_major && n_paths > 0 && n_cols > 0) {
            m.resize((size_t)n_paths);
            for (int64_t i = 0; i < n_paths; ++i) m[i].assign(row_major + i * n_cols, row_major + (i + 1) * n_cols);
        }
        AsymptoticAnalysis aa;
        const double v = aa.PredictOptionPrice(m, r, strike, maturity, dt, is_call != 0, sigma, dividend);
        if (price) *price = v;
        return MCG_OK;
    } catch (const std::exception& e) {
        return mcg::fail(MCG_ERR_INVALID, "%s", e.what());
    }
}

int mcg_debug_coalesce_selftest(int n_threads, int calls_per_thread, int* wrong) {
    if (n_threads < 1 || n_threads > 512 || calls_per_thread < 1 || !wrong) return mcg::fail(MCG_ERR_INVALID, "bad arguments");
    try {
        *wrong = mcg::co::selftest(n_threads, calls_per_thread);
    } catch (const std::exception& e) {
        return mcg::fail(MCG_ERR_OOM, "self-test: %s", e.what());
    }
    return MCG_OK;
}

int mcg_debug_coalesce_slots(int max_slots) {
    mcg::co::debug_max_slots(max_slots);
    return MCG_OK;
}

int mcg_compat_set_coalescing(int mode) {
    g_coalesce.store(mode < 0 || mode > 2 ? 1 : mode);
    return MCG_OK;
}

int mcg_compat_set_seed(uint64_t seed, int enabled) {
    g_seed.store(seed);
    g_seed_fixed.store(enabled != 0);
    return MCG_OK;
}

int mcg_compat_generate_paths(const double* hist, size_t n, int forward_steps, int path_num, double* out) {
    try {
        std::vector<double> h;
        if (hist && n) h.assign(hist, hist + n);
        RoughVolatility rv;
        auto m = rv.GenerateStockPricePaths(h, forward_steps, path_num);
        if (!m.empty() && !out) return mcg::fail(MCG_ERR_INVALID, "out is NULL");
        const size_t cols = (size_t)forward_steps + 1;
        for (size_t i = 0; i < m.size(); ++i) std::copy(m[i].begin(), m[i].end(), out + i * cols);
        return MCG_OK;
    } catch (const std::exception& e) {
        const std::string msg = e.what();
        const int code = msg == "Historical prices vector too small." ? MCG_ERR_HISTORY_TOO_SMALL : MCG_ERR_INVALID;
        return mcg::fail(code, "%s", msg.c_str());
    }
}

int mcg_compat_lsm_price(const double* row_major, int64_t n_paths, int n_cols, double r, double strike,
                         double maturity, double dt, int is_call, int poly_order, double* price) {
    try {
        std::vector<std::vector<double>> m;
        if (row_major && n_paths > 0 && n_cols > 0) {
            m.resize((size_t)n_paths);
            for (int64_t i = 0; i < n_paths; ++i) m[i].assign(row_major + i * n_cols, row_major + (i + 1) * n_cols);
        }
        LSM lsm;
        const double v = lsm.PredictOptionPrice(m, r, strike, maturity, dt, is_call != 0, poly_order);
        if (price) *price = v;
        return MCG_OK;
    } catch (const std::exception& e) {
        const std::string msg = e.what();
        const int code = msg == "LSM::PredictOptionPrice: Empty pricePaths." ? MCG_ERR_EMPTY_PATHS : MCG_ERR_INVALID;
        return mcg::fail(code, "%s", msg.c_str());
    }
}

}  // extern "C"
