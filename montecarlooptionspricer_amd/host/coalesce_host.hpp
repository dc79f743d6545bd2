// The calling thread's side of the combiner (host/coalesce.cpp) as host/dropin.cpp uses it.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../csrc/coalesce.hpp"

namespace mcg {
namespace co {

struct Waiter {  // a request in a lane's queue and the word its owner waits on
    Request* req = nullptr;
    std::atomic<int> state{0};
};

// A pricer call made AHEAD of the caller asking for it (dropin.cpp: co_price): the request, its waiter, and whether its answer can
// still be handed out (it can until the thread's matrix changes).
struct Prefetched {
    Request req;
    Waiter w;
    bool in_flight = false;  // enqueued and not yet waited for
    bool usable = false;     // belongs to the matrix the slot holds now
};

// One per host thread that has made a class-API call: its slot in the device arena (the matrix it generated or uploaded
// last stays there for the pricers that follow, PredictionGen.cpp:736-791) and its pinned, device-visible host buffer, which
// doubles as the host copy a later call's matrix is compared with -- element by element -- before the slot is trusted.
struct ThreadState {
    int slot = -1;
    int64_t slot_off = 0;
    double* pinned = nullptr;      // [n][m] path-major
    double* pinned_dev = nullptr;  // the device's address of it
    size_t pinned_cap = 0;         // doubles
    int n = 0, m = 0;
    bool valid = false;            // the slot holds pinned[n][m]
    Prefetched ahead[N_KINDS];     // by kind
    bool prefetched_for_this_matrix = false;
    ~ThreadState();
    bool have_slot();                      // this thread owns (or now gets) a slot of the arena; false: none left, or no combiner -- take the own-context route
    int prepare(int n_paths, int n_cols);  // slot + a pinned buffer of n_paths x n_cols doubles; drains what is in flight, invalidates `valid`; 0 or a status
    bool holds(const std::vector<std::vector<double>>& rows, size_t cols) const;
    int submit(Request& r);                // fills slot_off / host, blocks until answered; on failure mcg_last_error() holds r.err
    void prefetch(const Request& r);       // the same request, not waited for: its answer is taken later (or never)
    bool take_prefetched(int kind, double* price);  // waits for the kind's prefetched request if one is usable; false: make the call
    void drain();                          // wait for everything in flight (before the slot or the buffer changes hands)
    void forget_prefetched();
};
ThreadState& thread_state();
void debug_max_slots(int n);  // mcg_debug_coalesce_slots
int selftest(int n_threads, int calls_per_thread);  // mcg_debug_coalesce_selftest

}  // namespace co
}  // namespace mcg
