// The calling thread's side of the combiner (host/coalesce.cpp) as host/dropin.cpp uses it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../csrc/coalesce.hpp"

namespace mcg {
namespace co {

// One per host thread that has made a class-API call: its slot in the device arena (the matrix it generated or uploaded
// last stays there for the pricers that follow, PredictionGen.cpp:736-791) and its pinned, device-visible host buffer, which
// doubles as the host copy a later call's matrix is compared with -- element by element -- before the slot is trusted.
struct ThreadState {
    int slot = -1;
    int64_t slot_off = 0;
    double* pinned = nullptr;  // [n][m] path-major
    double* pinned_dev = nullptr;  // the device's address of it
    size_t pinned_cap = 0;     // doubles
    int n = 0, m = 0;
    bool valid = false;        // the slot holds pinned[n][m]
    ~ThreadState();
    int prepare(int n_paths, int n_cols);  // slot + a pinned buffer of n_paths x n_cols doubles (invalidates `valid`); 0 or a status
    bool holds(const std::vector<std::vector<double>>& rows, size_t cols) const;
    int submit(Request& r);                // fills slot_off / host, blocks until answered; on failure mcg_last_error() holds r.err
};
ThreadState& thread_state();

}  // namespace co
}  // namespace mcg
