// The combiner behind the reference's class API (csrc/coalesce.hpp has the why): request queue, leader election, the
// per-thread matrix slots of the device arena and the per-thread pinned host buffers.
//
// Protocol.  A caller pushes its request and, if nobody leads, becomes the leader: it takes EVERYTHING queued (its own
// request included), runs one round (co::execute_round: one upload, one launch per kind of call, one synchronisation),
// marks the requests done, hands the lead to the first caller that queued up meanwhile, and only then wakes the others --
// so the wake-ups overlap the next round.  While a round is on the device the other threads' calls pile up: that pile IS
// the next batch (group commit); nobody waits on a timer, and a lone caller is a round of one with no added latency.
// Waiting is a short spin, then a futex sleep on the waiter's own state word; the leader touches a waiter for the last
// time when it stores that word (a wake-up on an address whose owner has already left is harmless by futex semantics).
// Every request of a round that fails carries the failure; nothing throws across the leader.
//
// One such queue PER KIND of call (generate, AsymptoticAnalysis, BranchingProcesses, LSM, MartingaleOptimization), each with
// its own context (stream) and round buffers: the kinds' row kernels differ tenfold in latency (a row's LSM sweep walks its
// dates one after another: ~360 us at 126 steps; its MartingaleOptimization takes 20 us) and a caller of a short kind must
// not sit out a round of the long one -- with ONE queue a round cost the sum of its kinds' kernels (~570 us), a call waited
// one and a half rounds, a row five calls (measured, gpurun_out/r6e_unchanged.log).  The lanes' kernels overlap on the device.
#include "../csrc/coalesce.hpp"

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstring>
#include <mutex>
#include <vector>

#include "../csrc/mcg_internal.hpp"
#include "coalesce_host.hpp"

namespace mcg {
namespace co {
namespace {

enum : int { WAITING = 0, SLEEPING = 1, DONE = 2, LEAD = 3 };

struct Waiter {
    Request* req;
    std::atomic<int> state{WAITING};
};

void futex_wake(std::atomic<int>* w) { syscall(SYS_futex, reinterpret_cast<int*>(w), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }
void futex_sleep(std::atomic<int>* w, int expected) {
    syscall(SYS_futex, reinterpret_cast<int*>(w), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}

// Publish a state; wake the owner if it went to sleep.
void publish(Waiter* w, int s) {
    if (w->state.exchange(s, std::memory_order_acq_rel) == SLEEPING) futex_wake(&w->state);
}

constexpr int SLOTS_PER_CHUNK = 32;  // 32 x 2.09 MB = 67 MB of HBM per chunk, allocated when the 1st, 33rd, ... thread arrives
constexpr int MAX_CHUNKS = 16;       // 512 calling threads hold a slot; later ones take their own context

class Combiner {
public:
    int submit(Request& r);
    int acquire_slot(int64_t* off);
    void release_slot(int idx);
    int device() const { return device_; }
    int ready() {
        std::call_once(once_, [this] { init(); });
        if (init_rc_ != MCG_OK) return fail(init_rc_, "%s", init_err_.c_str());
        return MCG_OK;
    }

private:
    struct Lane {  // one kind of call: its queue, its lead, its stream
        mcg_ctx* ctx = nullptr;
        RoundBuffers rb;
        std::mutex mu;  // the queue and the lead
        std::vector<Waiter*> queue;
        bool leader_active = false;
    };
    void init();
    void lead(Lane& L, std::unique_lock<std::mutex>& lk, Waiter* self);

    std::once_flag once_;
    int init_rc_ = MCG_OK;
    std::string init_err_;
    int device_ = 0;
    Lane lanes_[N_KINDS];

    std::mutex slot_mu_;  // the arena
    std::vector<double*> chunks_;
    std::vector<int> free_slots_;
};

void Combiner::init() {
    if (const char* e = std::getenv("MCG_DEVICE")) device_ = std::atoi(e);
    for (Lane& L : lanes_) {
        if (mcg_init(&L.ctx, device_) != MCG_OK) {
            init_rc_ = MCG_ERR_NO_DEVICE;
            const char* m = mcg_last_error();
            init_err_ = m ? m : "mcg_init failed";
            L.ctx = nullptr;
            return;
        }
    }
}

int Combiner::acquire_slot(int64_t* off) {
    std::lock_guard<std::mutex> g(slot_mu_);
    if (free_slots_.empty()) {
        if ((int)chunks_.size() >= MAX_CHUNKS) return -1;
        double* p = nullptr;
        if (hipSetDevice(device_) != hipSuccess || hipMalloc((void**)&p, SLOT_DOUBLES * sizeof(double) * SLOTS_PER_CHUNK) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        const int first = (int)chunks_.size() * SLOTS_PER_CHUNK;
        chunks_.push_back(p);
        for (int k = SLOTS_PER_CHUNK - 1; k >= 0; --k) free_slots_.push_back(first + k);
    }
    const int idx = free_slots_.back();
    free_slots_.pop_back();
    // offsets are relative to the FIRST chunk (any two device allocations are a whole number of doubles apart)
    double* at = chunks_[(size_t)(idx / SLOTS_PER_CHUNK)] + (size_t)(idx % SLOTS_PER_CHUNK) * SLOT_DOUBLES;
    *off = (int64_t)(at - chunks_[0]);
    return idx;
}

void Combiner::release_slot(int idx) {
    std::lock_guard<std::mutex> g(slot_mu_);
    free_slots_.push_back(idx);
}

void Combiner::lead(Lane& L, std::unique_lock<std::mutex>& lk, Waiter* self) {
    // (this thread's own: the previous leader may still be walking ITS batch, waking callers, when this round starts)
    thread_local std::vector<Waiter*> batch_;
    thread_local std::vector<Request*> reqs_;
    batch_.clear();
    batch_.swap(L.queue);
    lk.unlock();
    reqs_.resize(batch_.size());
    for (size_t i = 0; i < batch_.size(); ++i) reqs_[i] = batch_[i]->req;
    double* base;
    {
        std::lock_guard<std::mutex> g(slot_mu_);
        base = chunks_.empty() ? nullptr : chunks_[0];
    }
    (void)execute_round(L.ctx, L.rb, base, reqs_.data(), (int)reqs_.size());  // every request now carries its status
    lk.lock();
    Waiter* next = nullptr;
    if (!L.queue.empty()) next = L.queue.front();
    else L.leader_active = false;
    lk.unlock();
    if (next) publish(next, LEAD);  // the next round starts while this thread wakes the answered callers
    const auto t0 = std::chrono::steady_clock::now();
    for (Waiter* w : batch_)
        if (w != self) publish(w, DONE);
    self->state.store(DONE, std::memory_order_release);
    g_stats.coalesced_wake_us.fetch_add((int64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(),
                                        std::memory_order_relaxed);
    lk.lock();
}

int Combiner::submit(Request& r) {
    int rc = ready();
    if (rc) {
        r.status = rc;
        std::snprintf(r.err, sizeof r.err, "%s", mcg_last_error());
        return rc;
    }
    Waiter w;
    w.req = &r;
    Lane& L = lanes_[r.kind >= 0 && r.kind < N_KINDS ? r.kind : 0];
    std::unique_lock<std::mutex> lk(L.mu);
    L.queue.push_back(&w);
    if (!L.leader_active) {
        L.leader_active = true;
        w.state.store(LEAD, std::memory_order_relaxed);
    }
    lk.unlock();
    for (int spins = 0;;) {
        int s = w.state.load(std::memory_order_acquire);
        if (s == DONE) break;
        if (s == LEAD) {
            lk.lock();
            lead(L, lk, &w);
            lk.unlock();
            continue;
        }
        if (++spins < 1000) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            continue;
        }
        int expect = WAITING;
        if (w.state.compare_exchange_strong(expect, SLEEPING, std::memory_order_acq_rel)) futex_sleep(&w.state, SLEEPING);
        else if (expect == SLEEPING) futex_sleep(&w.state, SLEEPING);  // (woken without a new state: sleep again)
    }
    return r.status;
}

Combiner& combiner() {
    static Combiner* c = new Combiner;  // never destroyed: threads may still be leaving when the process ends
    return *c;
}

}  // namespace

// ---- the calling thread's side --------------------------------------------------------------------------------------

ThreadState::~ThreadState() {
    if (pinned) (void)hipHostFree(pinned);
    if (slot >= 0) combiner().release_slot(slot);
}

int ThreadState::prepare(int n_paths, int n_cols) {
    valid = false;
    Combiner& c = combiner();
    int rc = c.ready();
    if (rc) return rc;
    if (slot < 0) {
        slot = c.acquire_slot(&slot_off);
        if (slot < 0) return fail(MCG_ERR_OOM, "no matrix slot left for this thread");
    }
    const size_t need = (size_t)n_paths * (size_t)n_cols;
    if (need > pinned_cap) {
        if (pinned) (void)hipHostFree(pinned);
        pinned = nullptr;
        pinned_cap = 0;
        const size_t cap = std::max<size_t>((need + 32767) & ~(size_t)32767, (size_t)1 << 15);  // whole 256 KiB
        MCG_HIP(hipSetDevice(c.device()));
        MCG_HIP(hipHostMalloc((void**)&pinned, cap * sizeof(double), hipHostMallocDefault));
        MCG_HIP(hipHostGetDevicePointer((void**)&pinned_dev, pinned, 0));
        pinned_cap = cap;
    }
    return MCG_OK;
}

bool ThreadState::holds(const std::vector<std::vector<double>>& rows, size_t cols) const {
    if (!valid || (size_t)n != rows.size() || (size_t)m != cols) return false;
    for (size_t i = 0; i < rows.size(); ++i)
        if (std::memcmp(rows[i].data(), pinned + i * cols, cols * sizeof(double)) != 0) return false;
    return true;
}

int ThreadState::submit(Request& r) {
    r.slot_off = slot_off;
    r.host = pinned;
    r.host_dev = pinned_dev;
    const int rc = combiner().submit(r);
    if (rc == MCG_OK) {
        valid = true;
        n = r.n_paths;
        m = r.n_steps + 1;
    } else {
        valid = false;
        set_error("%s", r.err[0] ? r.err : "coalesced call failed");
    }
    return rc;
}

ThreadState& thread_state() {
    thread_local ThreadState t;
    return t;
}

}  // namespace co
}  // namespace mcg
