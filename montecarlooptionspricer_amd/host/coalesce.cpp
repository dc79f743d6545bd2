// The combiner behind the reference's class API (csrc/coalesce.hpp has the why): one LANE per kind of call -- its request queue,
// its service thread, its context (stream) and round buffers --, the per-thread matrix slots of the device arena and the
// per-thread pinned host buffers.
//
// Protocol.  A caller pushes its request into the lane of its kind and waits (a short spin, then a futex sleep on the waiter's own
// state word).  The lane's SERVICE THREAD takes EVERYTHING queued, runs one round (co::execute_round: one upload, one launch per
// group of calls, one synchronisation), marks the requests done and wakes their owners; while a round is on the device the other
// threads' calls pile up: that pile IS the next batch (group commit) -- nobody waits on a timer, and a lone caller is a round of
// one.  The service thread touches a waiter for the last time when it stores its state word (a wake-up on an address whose owner
// has already left is harmless by futex semantics).  Every request of a round that fails carries the failure.
//
// One lane per kind (generate, AsymptoticAnalysis, BranchingProcesses, LSM, MartingaleOptimization): the kinds' row kernels differ
// tenfold in latency (a row's LSM sweep walks its dates one after another: ~360 us at 126 steps; its MartingaleOptimization
// takes 20 us) and a caller of a short kind must not sit out a round of the long one -- with ONE queue a round cost the sum of its
// kinds' kernels (~570 us) and a row five such waits (measured, gpurun_out/r6e_unchanged.log).  The lanes' kernels overlap on the
// device.  Service threads instead of "the first caller leads" (the first version of this file, 37-43 k rows/s at 128 threads): a
// caller can then have requests in SEVERAL lanes at once without owing any of them its attention -- which is what the prefetch of a
// row's other pricers needs (host/dropin.cpp: co_price) -- and no caller's own work waits behind the round it happened to lead.
#include "../csrc/coalesce.hpp"

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../csrc/mcg_internal.hpp"
#include "coalesce_host.hpp"

namespace mcg {
namespace co {
namespace {

enum : int { WAITING = 0, SLEEPING = 1, DONE = 2 };

void futex_wake(std::atomic<int>* w) { syscall(SYS_futex, reinterpret_cast<int*>(w), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }
void futex_sleep(std::atomic<int>* w, int expected) {
    // (bounded: a sleeper looks at the world again every 100 ms whatever happens -- cheap insurance against a wake-up lost to a
    //  process that is ending)
    const struct timespec ts = {0, 100 * 1000 * 1000};
    syscall(SYS_futex, reinterpret_cast<int*>(w), FUTEX_WAIT_PRIVATE, expected, &ts, nullptr, 0);
}
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// mcg_debug_coalesce_selftest: the protocol of this file -- queues, service threads, sleeps and wake-ups, prefetched requests,
// slots, thread exit -- on a host WITHOUT a GPU: rounds are answered by this function instead of co::execute_round (no context,
// no device memory, malloc'ed "pinned" buffers).  Set before the combiner's first use, for the life of the process.
typedef int (*RehearsalRound)(Request** reqs, int n);
std::atomic<RehearsalRound> g_rehearsal{nullptr};

constexpr int SLOTS_PER_CHUNK = 32;  // 32 x 2.09 MB = 67 MB of HBM per chunk, allocated when the 1st, 33rd, ... thread arrives
constexpr int MAX_CHUNKS = 16;       // 512 calling threads hold a slot; later ones take their own context

class Combiner {
public:
    void enqueue(Waiter* w);
    static void wait(Waiter* w);
    int acquire_slot(int64_t* off);
    void release_slot(int idx);
    int device() const { return device_; }
    int ready() {
        std::call_once(once_, [this] { init(); });
        if (init_rc_ != MCG_OK) return fail(init_rc_, "%s", init_err_.c_str());
        return MCG_OK;
    }
    void stop();

private:
    struct Lane {  // one kind of call: its queue, its service thread, its stream
        mcg_ctx* ctx = nullptr;
        RoundBuffers rb;
        std::mutex mu;
        std::vector<Waiter*> queue;
        std::atomic<int> idle{0};  // 1 while the service thread sleeps (futex word)
        std::thread th;
    };
    void init();
    void serve(Lane& L);

    std::once_flag once_;
    int init_rc_ = MCG_OK;
    std::string init_err_;
    int device_ = 0;
    Lane lanes_[N_KINDS];
    std::atomic<bool> stopping_{false};

public:
    std::atomic<bool> stopped_{false};  // the service threads have been joined (process exit)

private:

    std::mutex slot_mu_;  // the arena
    std::vector<double*> chunks_;
    std::vector<int> free_slots_;
    int slots_out_ = 0;

public:
    std::atomic<int> max_slots_{MAX_CHUNKS * SLOTS_PER_CHUNK};  // mcg_debug_coalesce_slots: fewer, so that a test can run out of them
};

Combiner& combiner() {
    static Combiner* c = new Combiner;  // never destroyed: threads may still be leaving when the process ends
    return *c;
}

void Combiner::init() {
    if (const char* e = std::getenv("MCG_DEVICE")) device_ = std::atoi(e);
    for (Lane& L : lanes_) {
        if (g_rehearsal.load()) break;
        if (mcg_init(&L.ctx, device_) != MCG_OK) {
            init_rc_ = MCG_ERR_NO_DEVICE;
            const char* m = mcg_last_error();
            init_err_ = m ? m : "mcg_init failed";
            L.ctx = nullptr;
            return;
        }
    }
    for (Lane& L : lanes_) L.th = std::thread([this, &L] { serve(L); });
    // the service threads leave before the HIP runtime's own tear-down (handlers run in reverse order of registration, and
    // the runtime registered its when the library was loaded)
    std::atexit([] { combiner().stop(); });
}

void Combiner::stop() {
    stopping_.store(true, std::memory_order_release);
    for (Lane& L : lanes_) {
        if (L.idle.exchange(0, std::memory_order_acq_rel) == 1) futex_wake(&L.idle);
        if (L.th.joinable()) L.th.join();
    }
    stopped_.store(true, std::memory_order_release);
}

void Combiner::serve(Lane& L) {
    const RehearsalRound rehearsal = g_rehearsal.load();
    if (!rehearsal) (void)hipSetDevice(device_);
    std::vector<Waiter*> batch;
    std::vector<Request*> reqs;
    for (;;) {
        for (int spins = 0;;) {  // wait for work: a short spin, then sleep until an enqueuer wakes us
            {
                std::lock_guard<std::mutex> g(L.mu);
                if (!L.queue.empty()) {
                    batch.swap(L.queue);
                    break;
                }
            }
            if (stopping_.load(std::memory_order_acquire)) return;
            if (++spins < 4000) {
                cpu_relax();
                continue;
            }
            L.idle.store(1, std::memory_order_release);
            bool work;
            {
                std::lock_guard<std::mutex> g(L.mu);  // (an enqueuer that pushed before this sees idle = 1 after it, or we see its push)
                work = !L.queue.empty();
            }
            if (!work && !stopping_.load(std::memory_order_acquire)) futex_sleep(&L.idle, 1);
            L.idle.store(0, std::memory_order_release);
            spins = 0;
        }
        reqs.resize(batch.size());
        for (size_t i = 0; i < batch.size(); ++i) reqs[i] = batch[i]->req;
        double* base;
        {
            std::lock_guard<std::mutex> g(slot_mu_);
            base = chunks_.empty() ? nullptr : chunks_[0];
        }
        if (rehearsal) (void)rehearsal(reqs.data(), (int)reqs.size());
        else (void)execute_round(L.ctx, L.rb, base, reqs.data(), (int)reqs.size());  // every request now carries its status
        const auto t0 = std::chrono::steady_clock::now();
        for (Waiter* w : batch)
            if (w->state.exchange(DONE, std::memory_order_acq_rel) == SLEEPING) futex_wake(&w->state);
        g_stats.coalesced_wake_us.fetch_add((int64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(),
                                            std::memory_order_relaxed);
        batch.clear();
    }
}

void Combiner::enqueue(Waiter* w) {
    Lane& L = lanes_[w->req->kind >= 0 && w->req->kind < N_KINDS ? w->req->kind : 0];
    w->state.store(WAITING, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> g(L.mu);
        L.queue.push_back(w);
    }
    if (L.idle.load(std::memory_order_acquire) == 1 && L.idle.exchange(0, std::memory_order_acq_rel) == 1) futex_wake(&L.idle);
}

void Combiner::wait(Waiter* w) {
    for (int spins = 0;;) {
        const int s = w->state.load(std::memory_order_acquire);
        if (s == DONE) return;
        // (the process is ending and the service threads have left: a thread that is only now running its destructors must not
        //  wait for an answer nobody will give -- its request is abandoned, unread)
        if (combiner().stopped_.load(std::memory_order_acquire)) {
            w->req->status = MCG_ERR_INVALID;
            return;
        }
        if (++spins < 1000) {
            cpu_relax();
            continue;
        }
        int expect = WAITING;
        if (w->state.compare_exchange_strong(expect, SLEEPING, std::memory_order_acq_rel) || expect == SLEEPING) futex_sleep(&w->state, SLEEPING);
    }
}

int Combiner::acquire_slot(int64_t* off) {
    std::lock_guard<std::mutex> g(slot_mu_);
    if (slots_out_ >= max_slots_.load(std::memory_order_relaxed)) return -1;
    if (free_slots_.empty()) {
        if ((int)chunks_.size() >= MAX_CHUNKS) return -1;
        double* p = nullptr;
        if (g_rehearsal.load()) {
            p = reinterpret_cast<double*>(((uintptr_t)chunks_.size() + 1) << 32);   // (never dereferenced: offsets only)
        } else if (hipSetDevice(device_) != hipSuccess || hipMalloc((void**)&p, SLOT_DOUBLES * sizeof(double) * SLOTS_PER_CHUNK) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        const int first = (int)chunks_.size() * SLOTS_PER_CHUNK;
        chunks_.push_back(p);
        for (int k = SLOTS_PER_CHUNK - 1; k >= 0; --k) free_slots_.push_back(first + k);
    }
    const int idx = free_slots_.back();
    free_slots_.pop_back();
    ++slots_out_;
    // offsets are relative to the FIRST chunk (any two device allocations are a whole number of doubles apart)
    double* at = chunks_[(size_t)(idx / SLOTS_PER_CHUNK)] + (size_t)(idx % SLOTS_PER_CHUNK) * SLOT_DOUBLES;
    *off = (int64_t)(at - chunks_[0]);
    return idx;
}

void Combiner::release_slot(int idx) {
    std::lock_guard<std::mutex> g(slot_mu_);
    free_slots_.push_back(idx);
    --slots_out_;
}

}  // namespace

// ---- the calling thread's side --------------------------------------------------------------------------------------

ThreadState::~ThreadState() {
    drain();  // nothing of ours may still be on the device when the buffers go
    if (pinned && g_rehearsal.load()) std::free(pinned);
    else if (pinned) (void)hipHostFree(pinned);
    if (slot >= 0) combiner().release_slot(slot);
}

bool ThreadState::have_slot() {
    if (slot >= 0) return true;
    Combiner& c = combiner();
    if (c.ready() != MCG_OK) return false;
    slot = c.acquire_slot(&slot_off);   // (512 threads hold one; the 513th prices on a context of its own, like every caller used to)
    return slot >= 0;
}

int ThreadState::prepare(int n_paths, int n_cols) {
    drain();
    forget_prefetched();
    valid = false;
    Combiner& c = combiner();
    int rc = c.ready();
    if (rc) return rc;
    if (!have_slot()) return fail(MCG_ERR_OOM, "no matrix slot left for this thread");
    const size_t need = (size_t)n_paths * (size_t)n_cols;
    if (need > pinned_cap) {
        const bool rehearsal = g_rehearsal.load() != nullptr;
        if (pinned && rehearsal) std::free(pinned);
        else if (pinned) (void)hipHostFree(pinned);
        pinned = nullptr;
        pinned_cap = 0;
        const size_t cap = std::max<size_t>((need + 32767) & ~(size_t)32767, (size_t)1 << 15);  // whole 256 KiB
        if (rehearsal) {
            pinned = pinned_dev = static_cast<double*>(std::malloc(cap * sizeof(double)));
            if (!pinned) return fail(MCG_ERR_OOM, "host allocation failed");
        } else {
            MCG_HIP(hipSetDevice(c.device()));
            MCG_HIP(hipHostMalloc((void**)&pinned, cap * sizeof(double), hipHostMallocDefault));
            MCG_HIP(hipHostGetDevicePointer((void**)&pinned_dev, pinned, 0));
        }
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
    Combiner& c = combiner();
    int rc = c.ready();
    if (rc) {
        r.status = rc;
        return rc;
    }
    r.slot_off = slot_off;
    r.host = pinned;
    r.host_dev = pinned_dev;
    Waiter w;
    w.req = &r;
    c.enqueue(&w);
    Combiner::wait(&w);
    rc = r.status;
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

void ThreadState::prefetch(const Request& r) {
    Combiner& c = combiner();
    if (r.kind < 0 || r.kind >= N_KINDS || c.ready() != MCG_OK) return;
    Prefetched& p = ahead[r.kind];
    if (p.in_flight) Combiner::wait(&p.w);
    p.req = r;
    p.req.slot_off = slot_off;
    p.req.host = pinned;
    p.req.host_dev = pinned_dev;
    p.req.upload = false;
    p.w.req = &p.req;
    p.in_flight = true;
    p.usable = true;
    c.enqueue(&p.w);
    g_stats.coalesced_prefetched.fetch_add(1, std::memory_order_relaxed);
}

bool ThreadState::take_prefetched(int kind, double* price) {
    Prefetched& p = ahead[kind];
    if (!p.usable) return false;
    if (p.in_flight) {
        Combiner::wait(&p.w);
        p.in_flight = false;
    }
    if (p.req.status != MCG_OK) {  // the ordinary call will say why
        p.usable = false;
        return false;
    }
    *price = p.req.price;
    g_stats.coalesced_prefetch_hits.fetch_add(1, std::memory_order_relaxed);
    return true;
}

void ThreadState::drain() {
    for (Prefetched& p : ahead) {
        if (p.in_flight) {
            Combiner::wait(&p.w);
            p.in_flight = false;
        }
    }
}

void ThreadState::forget_prefetched() {
    for (Prefetched& p : ahead) p.usable = false;
    prefetched_for_this_matrix = false;
}

ThreadState& thread_state() {
    thread_local ThreadState t;
    return t;
}

// ---- mcg_debug_coalesce_selftest -----------------------------------------------------------------------------------------------
namespace {

// the rehearsal's "device": a round takes ~50 us; a call's answer is a function of its arguments that no other call shares
double rehearsal_answer(const Request& q) { return q.strike * 2.0 + (double)q.kind + 0.25 * q.poly_order + 1e-3 * q.n_steps; }
int rehearsal_round(Request** reqs, int n) {
    usleep(50);
    for (int i = 0; i < n; ++i) {
        reqs[i]->price = rehearsal_answer(*reqs[i]);
        reqs[i]->status = MCG_OK;
    }
    g_stats.coalesced_rounds.fetch_add(1, std::memory_order_relaxed);
    g_stats.coalesced_calls.fetch_add(n, std::memory_order_relaxed);
    int64_t seen = g_stats.coalesced_peak_calls_per_round.load(std::memory_order_relaxed);
    while (n > seen && !g_stats.coalesced_peak_calls_per_round.compare_exchange_weak(seen, n, std::memory_order_relaxed)) {
    }
    return MCG_OK;
}

}  // namespace

// n_threads host threads, each making calls_per_thread calls of all five kinds through the combiner -- answered by
// rehearsal_round instead of the GPU --, every third one with two other kinds queued ahead and taken later (or left to be drained
// when the thread's matrix changes, or when the thread ends).  Returns the number of wrong or missing answers.
int selftest(int n_threads, int calls_per_thread) {
    g_rehearsal.store(rehearsal_round);
    std::atomic<int> wrong{0};
    auto work = [&](int tid) {
        ThreadState& t = thread_state();
        for (int c = 0; c < calls_per_thread; ++c) {
            Request q;
            q.kind = c % N_KINDS;
            q.n_paths = 250;
            q.n_steps = 5 + (tid + c) % 120;
            q.strike = 1000.0 * tid + c;
            q.poly_order = c % 5;
            if (q.kind == GEN || c % 7 == 0) {  // a new matrix: everything in flight is waited for first
                if (t.prepare(q.n_paths, q.n_steps + 1) != MCG_OK) {
                    ++wrong;
                    continue;
                }
            } else if (!t.have_slot()) {
                ++wrong;
                continue;
            }
            if (c % 3 == 0) {  // two requests ahead, in other lanes
                for (int k = 1; k <= 2; ++k) {
                    Request a = q;
                    a.kind = (q.kind + k) % N_KINDS;
                    a.strike = q.strike + 0.5 * k;
                    t.prefetch(a);
                }
            }
            if (t.submit(q) != MCG_OK || q.price != rehearsal_answer(q)) ++wrong;
            if (c % 6 == 0) {  // ... one of them taken, the other left for drain()
                const int k1 = (q.kind + 1) % N_KINDS;
                double price = -1.0;
                Request a = q;
                a.kind = k1;
                a.strike = q.strike + 0.5;
                if (!t.take_prefetched(k1, &price) || price != rehearsal_answer(a)) ++wrong;
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < n_threads; ++i) th.emplace_back(work, i);
    for (auto& x : th) x.join();  // (their ThreadState destructors have drained what they left in flight)
    return wrong.load();
}

void debug_max_slots(int n) { combiner().max_slots_.store(n < 0 ? MAX_CHUNKS * SLOTS_PER_CHUNK : n, std::memory_order_relaxed); }

}  // namespace co
}  // namespace mcg
