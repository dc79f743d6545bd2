// Node-local collective over POSIX shared memory: one process per GPU, all on one host (the north-star's shape: eight
// MI355X of one node).  Three services on one segment:
//
//   * a HOST all-reduce of a few doubles (payoff sums, the per-date moments of the per-date LSM kernels): copy down,
//     publish in the segment, barrier, sum in rank order, barrier, copy up -- the mcg_allreduce_fn of this ctx;
//   * a DEVICE mailbox for the one-launch LSM sweeps (kernels_lsm.hip): the per-date exchange between the GPUs happens
//     INSIDE the kernel -- no launch, no host round trip, no RCCL kernel competing for the CUs the sweep occupies.  One
//     slot per exchange round, one row per rank, rows written once per sweep (no recycling); each rank re-arms what it
//     owns with the reserved NaN before a sweep and a host barrier orders that against everybody's launch.  Two homes:
//       - HOST mailbox (always there): the segment itself, registered with HIP; a reducing workgroup writes its row with
//         system-scope stores and polls the other ranks' rows with system-scope loads -- every poll crosses PCIe;
//       - PEER-MEMORY mailbox (mcg_comm_shm_peer_mailbox, opt-in): every rank keeps a mailbox in its OWN HBM, exported
//         with hipIpcGetMemHandle through the segment and opened by the peers; a reducing workgroup PUSHES its row into
//         every peer's mailbox (one-way stores over xGMI) and polls only local memory.  Taken into use only if every
//         rank allocated, exported, opened AND passed an in-kernel ping over the mapping; otherwise all ranks stay on
//         the host mailbox together.
//   * the barrier and the small integer agreements (time-out flags) the two need.
//
// Every rank computes the same global moments (same published values, summed in rank order), hence bit-identical
// coefficients and the same refinement decisions, so the ranks stay in lock-step without further agreement.
//
// Failure containment: a barrier that times out, or a rank that hits a local error between two collective steps,
// poisons the segment (`abort`); every later barrier on it fails at once on every rank instead of drifting out of step.
// A stale segment of a crashed job under the same name cannot be joined: a peer is in only once the job's live rank 0 has
// echoed the random word it wrote into the segment, and while it waits for that it keeps checking that the inode it
// mapped is still the one the name leads to, re-opening if rank 0 has replaced it meanwhile.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <new>
#include <random>
#include <string>

#include "mcg_internal.hpp"

namespace mcg {

struct ShmHeader {
    std::atomic<uint32_t> magic;
    uint32_t n_ranks;
    std::atomic<uint32_t> arrive;
    std::atomic<uint32_t> sense;
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> abort;  // poisoned: every barrier fails from now on
    uint32_t pad[10];
    // joining: rank r > 0 writes a fresh random word into hello[r] and waits until rank 0 -- the LIVE rank 0 of this job,
    // serving these while it waits for everybody -- echoes it in ack[r]; nothing left behind by a crashed job can do that
    std::atomic<uint64_t> hello[SHM_MAX_RANKS], ack[SHM_MAX_RANKS];
    double host_slots[SHM_MAX_RANKS][64];
    unsigned char ipc_handle[SHM_MAX_RANKS][64];  // hipIpcMemHandle_t of each rank's peer-memory mailbox
    char pci_bus_id[SHM_MAX_RANKS][64];           // its GPU ("0000:c1:00.0"), for the peer-access check before the open
    // ranks that are THREADS of one process (one host thread per GPU, the reference's own OpenMP shape): an IPC handle
    // cannot be opened by the process that exported it, and need not be -- the owner's pointer is valid in the peer as it
    // stands (one address space), peer access between the two devices is all it takes
    uint64_t owner_pid[SHM_MAX_RANKS], owner_ptr[SHM_MAX_RANKS];
    // "the same process" = the same pid AND the same per-process random token (two containers sharing /dev/shm can hold
    // equal pids; a foreign pointer must never be dereferenced)
    uint64_t owner_token[SHM_MAX_RANKS];
    // ranks that currently hold rank r's OWN pointer (same-process borrowers), plus BORROW_ORPHANED once the owner has let
    // go of it.  An IPC mapping keeps the owner's memory alive until every opener has closed it; a borrowed pointer does
    // not -- so whoever brings this word to "orphaned, nobody left" frees the mailbox: the owner when no borrower is left,
    // else the LAST borrower to let go (one address space: any thread may hipFree it).  Nobody waits for anybody, nothing
    // leaks, whatever order the rank threads' contexts are closed in (round 5 waited up to 5 s and then kept the megabyte).
    std::atomic<uint32_t> borrowers[SHM_MAX_RANKS];
};
static_assert(sizeof(ShmHeader) % 64 == 0, "mailbox starts cache-line aligned");
static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle fits its slot");

struct ShmComm {
    std::string name;
    int fd = -1;
    void* base = nullptr;
    size_t bytes = 0;
    ShmHeader* hdr = nullptr;
    double* mbox_host = nullptr;
    double* mbox_dev = nullptr;   // device mapping of the host mailbox
    bool registered = false;      // hipHostRegister done (false: host-only attach of the CPU tests)
    uint32_t local_sense = 0;
    int n_ranks = 1, rank = 0;
    double timeout_s = 120.0;
    double* pinned = nullptr;     // 64 doubles
    // peer-memory mailbox
    bool peer_active = false;
    double* peer_own = nullptr;                    // this rank's mailbox in its own HBM
    double* peer_map[SHM_MAX_RANKS] = {nullptr};   // every rank's mailbox as mapped here ([rank] = peer_own)
    bool peer_borrowed[SHM_MAX_RANKS] = {false};   // [r]: peer_map[r] is rank r's own pointer (same process), not an IPC mapping
    const char* peer_memory_kind = "";
};

namespace {

constexpr uint32_t BORROW_ORPHANED = 0x80000000u;  // ShmHeader::borrowers: the owner has gone, the last borrower frees
constexpr uint32_t SHM_MAGIC = 0x4D434756u;  // "MCGV" (layout of round 5: owner_token, borrowers)

uint64_t process_token() {  // one random word per process, never written anywhere but into the segment's owner_token
    static const uint64_t t = [] {
        std::random_device rd;
        return (((uint64_t)rd() << 32) ^ (uint64_t)rd()) | 1ull;
    }();
    return t;
}
constexpr int SHM_FLAG_SLOT = 63;            // entry of a rank's host slot row that carries shm_sum_flag's integer

size_t mailbox_bytes() { return (size_t)SHM_MAX_ROUNDS * SHM_MAX_RANKS * SHM_ROW_DOUBLES * sizeof(double); }
size_t shm_bytes() { return sizeof(ShmHeader) + mailbox_bytes(); }

double since(const std::chrono::steady_clock::time_point& t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

bool shm_barrier(ShmComm* c) {
    if (c->hdr->abort.load(std::memory_order_acquire)) return false;
    c->local_sense ^= 1u;
    const uint32_t s = c->local_sense;
    if (c->hdr->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->n_ranks) {
        c->hdr->arrive.store(0, std::memory_order_relaxed);
        c->hdr->sense.store(s, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (c->hdr->sense.load(std::memory_order_acquire) != s) {
        if (++spins > 2000) {
            sched_yield();
            if (c->hdr->abort.load(std::memory_order_acquire)) return false;
            if (since(t0) > c->timeout_s) {
                c->hdr->abort.store(1, std::memory_order_release);  // the others fail fast instead of waiting their own 120 s
                return false;
            }
        }
    }
    return true;
}

const char* barrier_error(ShmComm* c) {
    g_stats.shm_barrier_failures.fetch_add(1, std::memory_order_relaxed);
    return c->hdr->abort.load() ? "shared-memory communicator is poisoned (a rank timed out or failed between two collective steps)"
                                : "shared-memory barrier failed";
}

int shm_allreduce(void* user, double* buf, int count, void* stream) {
    mcg_ctx* ctx = (mcg_ctx*)user;
    ShmComm* c = ctx->shm;
    if (!c || count < 0 || count >= SHM_FLAG_SLOT) {
        set_error("shared-memory all-reduce: bad count %d", count);
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(c->pinned, buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        c->hdr->abort.store(1, std::memory_order_release);
        set_error("shared-memory all-reduce: device copy failed");
        return 1;
    }
    std::memcpy(c->hdr->host_slots[c->rank], c->pinned, (size_t)count * sizeof(double));
    if (!shm_barrier(c)) {
        set_error("shared-memory all-reduce: %s", barrier_error(c));
        return 1;
    }
    for (int i = 0; i < count; ++i) {
        double s = 0.0;
        for (int r = 0; r < c->n_ranks; ++r) s += c->hdr->host_slots[r][i];  // rank order: the same bits on every rank
        c->pinned[i] = s;
    }
    if (!shm_barrier(c)) {  // nobody overwrites its slot before everyone has read it
        set_error("shared-memory all-reduce: %s", barrier_error(c));
        return 1;
    }
    if (hipMemcpyAsync(buf, c->pinned, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        c->hdr->abort.store(1, std::memory_order_release);
        set_error("shared-memory all-reduce: device copy failed");
        return 1;
    }
    return 0;
}

// Is the segment this rank mapped still the one `name` leads to?  (Rank 0 unlinks a stale segment before it creates the
// job's own; a peer that was quicker has mapped the stale one.)
bool still_linked(ShmComm* c) {
    struct stat mine, now;
    if (fstat(c->fd, &mine) != 0) return false;
    const int fd2 = shm_open(c->name.c_str(), O_RDWR, 0600);
    if (fd2 < 0) return false;
    const bool same = fstat(fd2, &now) == 0 && now.st_ino == mine.st_ino && now.st_dev == mine.st_dev;
    close(fd2);
    return same;
}

void unmap(ShmComm* c) {
    if (c->base) munmap(c->base, c->bytes);
    if (c->fd >= 0) close(c->fd);
    c->base = nullptr;
    c->hdr = nullptr;
    c->mbox_host = nullptr;
    c->fd = -1;
}

// Host part of joining the segment (no HIP call: the CPU tests drive it through mcg_debug_shm_*).  On MCG_OK every rank
// of the job has mapped the same, freshly initialised segment.
int shm_attach(ShmComm* c) {
    const auto t0 = std::chrono::steady_clock::now();
    const char* name = c->name.c_str();
    for (;;) {
        if (c->rank == 0) {
            shm_unlink(name);  // a stale segment of a crashed run
            c->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
            if (c->fd < 0 || ftruncate(c->fd, (off_t)c->bytes) != 0) return fail(MCG_ERR_COMM, "cannot create shared-memory segment %s", name);
        } else {
            for (;;) {
                c->fd = shm_open(name, O_RDWR, 0600);
                struct stat st;
                if (c->fd >= 0 && fstat(c->fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break;
                if (c->fd >= 0) close(c->fd);
                c->fd = -1;
                if (since(t0) > c->timeout_s) return fail(MCG_ERR_COMM, "shared-memory segment %s did not appear", name);
                usleep(2000);
            }
        }
        c->base = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
        if (c->base == MAP_FAILED) {
            c->base = nullptr;
            return fail(MCG_ERR_COMM, "mmap of %s failed", name);
        }
        c->hdr = reinterpret_cast<ShmHeader*>(c->base);
        c->mbox_host = reinterpret_cast<double*>(reinterpret_cast<char*>(c->base) + sizeof(ShmHeader));
        bool stale = false;
        if (c->rank == 0) {
            c->hdr->n_ranks = (uint32_t)c->n_ranks;
            c->hdr->arrive.store(0);
            c->hdr->sense.store(0);
            c->hdr->attached.store(0);
            c->hdr->abort.store(0);
            for (int r = 0; r < SHM_MAX_RANKS; ++r) {
                c->hdr->hello[r].store(0);
                c->hdr->ack[r].store(0);
            }
            c->hdr->magic.store(SHM_MAGIC, std::memory_order_release);
        } else {
            while (c->hdr->magic.load(std::memory_order_acquire) != SHM_MAGIC && !stale) {
                if (since(t0) > c->timeout_s) return fail(MCG_ERR_COMM, "shared-memory segment %s was never initialised", name);
                usleep(1000);
                stale = !still_linked(c);
            }
            if (!stale && c->hdr->n_ranks != (uint32_t)c->n_ranks && still_linked(c))
                return fail(MCG_ERR_COMM, "segment %s was created for %u ranks, not %d", name, c->hdr->n_ranks, c->n_ranks);
            if (!stale && c->hdr->n_ranks != (uint32_t)c->n_ranks) stale = true;
        }
        if (!stale && c->rank != 0) {  // is this the segment of a live job?  its rank 0 echoes a fresh random word
            std::random_device rd;
            const uint64_t word = (((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 17)) | 1u;
            c->hdr->hello[c->rank].store(word, std::memory_order_release);
            unsigned polls = 0;
            while (c->hdr->ack[c->rank].load(std::memory_order_acquire) != word) {
                if (since(t0) > c->timeout_s) return fail(MCG_ERR_COMM, "rank 0 never answered on segment %s", name);
                usleep(500);
                if ((++polls % 16) == 0 && !still_linked(c)) {  // rank 0 has replaced what this rank mapped
                    stale = true;
                    break;
                }
            }
        }
        // everybody attached (the creator may unlink the name only at release; a late rank still needs it until here)
        if (!stale) {
            c->hdr->attached.fetch_add(1);
            while (c->hdr->attached.load() < (uint32_t)c->n_ranks) {
                if (since(t0) > c->timeout_s)
                    return fail(MCG_ERR_COMM, "only %u of %d ranks attached to %s", c->hdr->attached.load(), c->n_ranks, name);
                if (c->rank == 0) {
                    for (int r = 1; r < c->n_ranks; ++r) {
                        const uint64_t h = c->hdr->hello[r].load(std::memory_order_acquire);
                        if (h != 0 && c->hdr->ack[r].load(std::memory_order_relaxed) != h) c->hdr->ack[r].store(h, std::memory_order_release);
                    }
                    usleep(200);
                } else {
                    usleep(1000);
                }
            }
        }
        if (!stale) return MCG_OK;
        unmap(c);
        usleep(1000);
    }
}

}  // namespace

// Re-arm what this rank owns of the first `rounds` slots and wait until every rank has done so (before a sweep's launch).
int shm_arm_mailbox(mcg_ctx* ctx, int rounds, uint64_t sentinel_bits) {
    ShmComm* c = ctx->shm;
    if (!c) return fail(MCG_ERR_COMM, "no shared-memory communicator");
    if (rounds > SHM_MAX_ROUNDS) return fail(MCG_ERR_INVALID, "too many exchange rounds for the mailbox");
    if (c->peer_active) {  // every rank's rows of this rank's own mailbox: the peers push into it
        if (peer_arm(ctx, c->peer_own, rounds, c->n_ranks, sentinel_bits) != 0) {
            shm_poison(ctx);
            return fail(MCG_ERR_HIP, "re-arming the peer-memory mailbox failed");
        }
    } else {  // this rank's rows of the shared host mailbox
        for (int q = 0; q < rounds; ++q) {
            uint64_t* row = reinterpret_cast<uint64_t*>(c->mbox_host + ((size_t)q * SHM_MAX_RANKS + c->rank) * SHM_ROW_DOUBLES);
            for (int t = 0; t < SHM_ROW_DOUBLES; ++t) row[t] = sentinel_bits;
        }
        std::atomic_thread_fence(std::memory_order_seq_cst);
    }
    if (!shm_barrier(c)) return fail(MCG_ERR_COMM, "%s", barrier_error(c));
    return MCG_OK;
}

// sum of one integer over the ranks (agreement on "did any rank's hand-shake time out")
int shm_sum_flag(mcg_ctx* ctx, int flag, int* total) {
    ShmComm* c = ctx->shm;
    if (!c) return fail(MCG_ERR_COMM, "no shared-memory communicator");
    c->hdr->host_slots[c->rank][SHM_FLAG_SLOT] = (double)flag;
    if (!shm_barrier(c)) return fail(MCG_ERR_COMM, "%s", barrier_error(c));
    double s = 0.0;
    for (int r = 0; r < c->n_ranks; ++r) s += c->hdr->host_slots[r][SHM_FLAG_SLOT];
    if (!shm_barrier(c)) return fail(MCG_ERR_COMM, "%s", barrier_error(c));
    *total = (int)s;
    return MCG_OK;
}

void shm_poison(mcg_ctx* ctx) {
    if (ctx->shm && ctx->shm->hdr) ctx->shm->hdr->abort.store(1, std::memory_order_release);
}

double* shm_mailbox_device(mcg_ctx* ctx) {
    if (!ctx->shm) return nullptr;
    return ctx->shm->peer_active ? ctx->shm->peer_own : ctx->shm->mbox_dev;
}
// Push targets of the in-kernel exchange: every rank's peer-memory mailbox as mapped in this process, or nullptr when
// the host mailbox is in use (everybody writes and polls the one shared copy).
double* const* shm_mailbox_peers(mcg_ctx* ctx) { return ctx->shm && ctx->shm->peer_active ? ctx->shm->peer_map : nullptr; }
int shm_rank(mcg_ctx* ctx) { return ctx->shm ? ctx->shm->rank : 0; }
int shm_n_ranks(mcg_ctx* ctx) { return ctx->shm ? ctx->shm->n_ranks : 1; }
int shm_attached(mcg_ctx* ctx) { return ctx->shm && ctx->shm->hdr ? (int)ctx->shm->hdr->attached.load() : 0; }
bool shm_peer_active(mcg_ctx* ctx) { return ctx->shm && ctx->shm->peer_active; }

// May this rank map a peer's mailbox?  (Pure decision, unit-tested on the CPU through mcg_debug_peer_decision.)
//   same process      : the owner's pointer is used as it is; across devices that needs peer access;
//   another process   : its GPU must resolve to a device THIS process can see (a rank confined to its own GPU by
//                       HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES cannot check reachability: a mapping that is not
//                       reachable faults the first kernel that touches it, so the answer is no and all ranks stay on the
//                       host mailbox together), and across devices peer access must be available.
bool peer_map_allowed(bool same_process, bool bus_id_resolves, bool same_device, bool can_access_peer) {
    (void)same_process;
    if (!bus_id_resolves) return false;
    return same_device || can_access_peer;
}

static void peer_release(ShmComm* c) {
    for (int r = 0; r < SHM_MAX_RANKS; ++r) {
        if (c->peer_map[r] && r != c->rank) {
            if (c->peer_borrowed[r]) {  // (our kernels are done: callers synchronise the stream first)
                if (c->hdr->borrowers[r].fetch_sub(1, std::memory_order_acq_rel) == (BORROW_ORPHANED | 1u)) {
                    (void)hipFree(c->peer_map[r]);  // the owner left before us and we were the last to hold its mailbox
                    c->hdr->borrowers[r].store(0, std::memory_order_release);
                }
            } else {
                (void)hipIpcCloseMemHandle(c->peer_map[r]);
            }
        }
        c->peer_map[r] = nullptr;
        c->peer_borrowed[r] = false;
    }
    if (c->peer_own) {
        // Same-process rank threads hold this pointer as it stands.  Lifetime rule (mcgpu.h, mcg_comm_init_shm): a rank's
        // mailbox outlives every peer that maps it -- so if a borrower is left, the mailbox is ITS to free when it lets go.
        if (c->hdr->borrowers[c->rank].fetch_or(BORROW_ORPHANED, std::memory_order_acq_rel) == 0) {
            (void)hipFree(c->peer_own);
            c->hdr->borrowers[c->rank].store(0, std::memory_order_release);
        } else {
            g_stats.peer_mailbox_kept.fetch_add(1, std::memory_order_relaxed);  // handed to the last borrower (mcg_stats)
        }
    }
    c->peer_own = nullptr;
    c->peer_active = false;
}

void shm_release(mcg_ctx* ctx) {
    ShmComm* c = ctx->shm;
    if (!c) return;
    if (ctx->allreduce == shm_allreduce) {
        ctx->allreduce = nullptr;
        ctx->allreduce_user = nullptr;
    }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);  // (this rank's own pushes into borrowed or mapped mailboxes are over)
    peer_release(c);
    if (c->registered) (void)hipHostUnregister(c->base);
    if (c->pinned) (void)hipHostFree(c->pinned);
    unmap(c);
    if (c->rank == 0) shm_unlink(c->name.c_str());
    delete c;
    ctx->shm = nullptr;
}

}  // namespace mcg

using namespace mcg;

extern "C" int mcg_comm_init_shm(mcg_ctx* ctx, const char* name, int n_ranks, int rank) {
    if (!ctx || !name || name[0] != '/') return fail(MCG_ERR_INVALID, "ctx is NULL or the segment name does not start with '/'");
    if (n_ranks < 1 || n_ranks > SHM_MAX_RANKS || rank < 0 || rank >= n_ranks)
        return fail(MCG_ERR_INVALID, "bad rank %d of %d (at most %d ranks share a segment)", rank, n_ranks, SHM_MAX_RANKS);
    MCG_HIP(hipSetDevice(ctx->device));
    shm_release(ctx);
    ShmComm* c = new (std::nothrow) ShmComm();
    if (!c) return fail(MCG_ERR_OOM, "host allocation failed");
    c->name = name;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->bytes = shm_bytes();
    ctx->shm = c;
    int rc = shm_attach(c);
    if (rc) {
        shm_release(ctx);
        return rc;
    }
    if (hipHostRegister(c->base, c->bytes, hipHostRegisterMapped) != hipSuccess) {
        (void)hipGetLastError();
        shm_poison(ctx);
        shm_release(ctx);
        return fail(MCG_ERR_HIP, "hipHostRegister of the shared segment failed");
    }
    c->registered = true;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, c->mbox_host, 0) != hipSuccess || hipHostMalloc((void**)&c->pinned, 64 * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        shm_poison(ctx);
        shm_release(ctx);
        return fail(MCG_ERR_HIP, "device mapping of the shared segment failed");
    }
    c->mbox_dev = (double*)dev;
    ctx->allreduce = shm_allreduce;
    ctx->allreduce_user = ctx;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return MCG_OK;
}

// Collective over the ranks of the segment: move the in-kernel mailbox into peer-mapped device memory, or (enable = 0)
// back into the host segment.  Every step is agreed on by all ranks, so they all end up in the same mode.
extern "C" int mcg_comm_shm_peer_mailbox(mcg_ctx* ctx, int enable, int* active) {
    if (active) *active = 0;
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    ShmComm* c = ctx->shm;
    if (!c || !c->registered) return fail(MCG_ERR_COMM, "no shared-memory communicator (call mcg_comm_init_shm first)");
    MCG_HIP(hipSetDevice(ctx->device));
    MCG_HIP(hipStreamSynchronize(ctx->stream));
    int all = 0, rc;
    if (!enable) {
        if ((rc = shm_sum_flag(ctx, 0, &all))) return rc;  // (everybody is here, nobody is inside a sweep)
        peer_release(c);
        return MCG_OK;
    }
    if (c->peer_active) {
        if (active) *active = 1;
        return MCG_OK;
    }
    if (c->peer_own) peer_release(c);  // (left behind by an earlier attempt that failed half-way)
    // (a mailbox of an earlier enable that a rank thread still holds: its count must not be mixed with the new one's)
    const bool earlier_still_borrowed = c->hdr->borrowers[c->rank].load(std::memory_order_acquire) != 0;
    // 1. allocate + export.  Uncached device memory first (every access goes to HBM: what a peer stores over xGMI is what
    // a local poll reads), fine-grained next, an ordinary allocation last; the in-kernel ping below is the judge.
    int ok = 0;
    {
        const unsigned kinds[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, hipDeviceMallocDefault};
        const char* names[3] = {"uncached device memory", "fine-grained device memory", "device memory"};
        for (int k = 0; k < 3 && !ok && !earlier_still_borrowed; ++k) {
            void* p = nullptr;
            if (hipExtMallocWithFlags(&p, mailbox_bytes(), kinds[k]) != hipSuccess) {
                (void)hipGetLastError();
                continue;
            }
            hipIpcMemHandle_t h;
            if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipFree(p);
                continue;
            }
            std::memcpy(c->hdr->ipc_handle[c->rank], &h, sizeof h);
            std::memset(c->hdr->pci_bus_id[c->rank], 0, sizeof c->hdr->pci_bus_id[0]);
            if (hipDeviceGetPCIBusId(c->hdr->pci_bus_id[c->rank], (int)sizeof c->hdr->pci_bus_id[0] - 1, ctx->device) != hipSuccess) {
                (void)hipGetLastError();
                c->hdr->pci_bus_id[c->rank][0] = 0;
            }
            c->hdr->owner_pid[c->rank] = (uint64_t)getpid();
            c->hdr->owner_token[c->rank] = process_token();
            c->hdr->owner_ptr[c->rank] = (uint64_t)(uintptr_t)p;
            c->peer_own = (double*)p;
            c->peer_memory_kind = names[k];
            ok = 1;
        }
    }
    if ((rc = shm_sum_flag(ctx, ok, &all))) {  // (its barriers also publish the handles)
        peer_release(c);
        return rc;
    }
    if (all != c->n_ranks) {
        peer_release(c);
        g_stats.peer_mailbox_refused.fetch_add(1, std::memory_order_relaxed);
        return MCG_OK;  // host mailbox stays
    }
    // 2. open the peers' mailboxes
    ok = 1;
    c->peer_map[c->rank] = c->peer_own;
    for (int r = 0; r < c->n_ranks && ok; ++r) {
        if (r == c->rank) continue;
        const bool same_process = c->hdr->owner_pid[r] == (uint64_t)getpid() && c->hdr->owner_token[r] == process_token();
        int peer_dev = -1, can = 0;
        const bool resolves = c->hdr->pci_bus_id[r][0] && hipDeviceGetByPCIBusId(&peer_dev, c->hdr->pci_bus_id[r]) == hipSuccess;
        if (!resolves) (void)hipGetLastError();
        const bool same_device = resolves && peer_dev == ctx->device;
        if (resolves && !same_device && (hipDeviceCanAccessPeer(&can, ctx->device, peer_dev) != hipSuccess || !can)) {
            (void)hipGetLastError();
            can = 0;
        }
        if (!peer_map_allowed(same_process, resolves, same_device, can != 0)) {
            ok = 0;
            break;
        }
        if (same_process) {  // a thread of this process: its pointer, plus peer access between the two devices
            if (!same_device) {
                const hipError_t e = hipDeviceEnablePeerAccess(peer_dev, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = 0;
                (void)hipGetLastError();
            }
            if (ok) {
                c->peer_map[r] = (double*)(uintptr_t)c->hdr->owner_ptr[r];
                c->peer_borrowed[r] = true;
                c->hdr->borrowers[r].fetch_add(1, std::memory_order_acq_rel);
            }
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, c->hdr->ipc_handle[r], sizeof h);
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            ok = 0;
        } else {
            c->peer_map[r] = (double*)p;
        }
    }
    if ((rc = shm_sum_flag(ctx, ok, &all))) {
        peer_release(c);
        return rc;
    }
    if (all != c->n_ranks) {
        peer_release(c);
        g_stats.peer_mailbox_refused.fetch_add(1, std::memory_order_relaxed);
        return MCG_OK;
    }
    // 3. ping: every rank arms its mailbox, then one wavefront per rank pushes a token into every peer's mailbox and
    // waits (bounded) for everybody's token in its own
    c->peer_active = true;
    const uint64_t sentinel = 0xFFF85EA7FFF85EA7ull;
    rc = shm_arm_mailbox(ctx, 1, sentinel);
    if (rc) {
        peer_release(c);
        return rc;
    }
    int failed = peer_ping(ctx, c->peer_map, c->n_ranks, c->rank);
    if ((rc = shm_sum_flag(ctx, failed ? 1 : 0, &all))) {
        peer_release(c);
        return rc;
    }
    if (all != 0) {
        peer_release(c);
        g_stats.peer_mailbox_refused.fetch_add(1, std::memory_order_relaxed);
        return MCG_OK;
    }
    g_stats.peer_mailbox_enabled.fetch_add(1, std::memory_order_relaxed);
    if (active) *active = 1;
    return MCG_OK;
}

extern "C" int mcg_debug_peer_decision(int same_process, int bus_id_resolves, int same_device, int can_access_peer) {
    return peer_map_allowed(same_process != 0, bus_id_resolves != 0, same_device != 0, can_access_peer != 0) ? 1 : 0;
}

extern "C" int mcg_comm_info(mcg_ctx* ctx, int* kind, int* n_ranks, int* rank, int* seen_ranks) {
    if (!ctx) return fail(MCG_ERR_INVALID, "ctx is NULL");
    int k = 0, seen = 0;
    if (ctx->shm) {
        k = shm_peer_active(ctx) ? 4 : 3;
        seen = shm_attached(ctx);
    } else if (ctx->rccl_comm) {
        k = 2;
        seen = comm_rccl_count(ctx);
    } else if (ctx->allreduce) {
        k = 1;
    }
    if (kind) *kind = k;
    if (n_ranks) *n_ranks = k ? ctx->n_ranks : 1;
    if (rank) *rank = k ? ctx->rank : 0;
    if (seen_ranks) *seen_ranks = seen;
    return MCG_OK;
}

// ---- host-only test hooks (no HIP call: usable without a GPU) ------------------------------------------------------
// The segment protocol on its own: join (with the stale-segment check), barrier, poison, leave.
extern "C" int mcg_debug_shm_attach(const char* name, int n_ranks, int rank, double timeout_s, void** handle) {
    if (!name || name[0] != '/' || !handle) return fail(MCG_ERR_INVALID, "bad arguments");
    if (n_ranks < 1 || n_ranks > SHM_MAX_RANKS || rank < 0 || rank >= n_ranks) return fail(MCG_ERR_INVALID, "bad rank");
    ShmComm* c = new (std::nothrow) ShmComm();
    if (!c) return fail(MCG_ERR_OOM, "host allocation failed");
    c->name = name;
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->bytes = shm_bytes();
    if (timeout_s > 0.0) c->timeout_s = timeout_s;
    int rc = shm_attach(c);
    if (rc) {
        unmap(c);
        if (rank == 0) shm_unlink(name);
        delete c;
        return rc;
    }
    *handle = c;
    return MCG_OK;
}

extern "C" int mcg_debug_shm_barrier(void* handle) {
    ShmComm* c = (ShmComm*)handle;
    if (!c) return fail(MCG_ERR_INVALID, "handle is NULL");
    if (!shm_barrier(c)) return fail(MCG_ERR_COMM, "%s", barrier_error(c));
    return MCG_OK;
}

extern "C" int mcg_debug_shm_poison(void* handle) {
    ShmComm* c = (ShmComm*)handle;
    if (!c) return fail(MCG_ERR_INVALID, "handle is NULL");
    c->hdr->abort.store(1, std::memory_order_release);
    return MCG_OK;
}

extern "C" int mcg_debug_shm_detach(void* handle) {
    ShmComm* c = (ShmComm*)handle;
    if (!c) return MCG_OK;
    unmap(c);
    if (c->rank == 0) shm_unlink(c->name.c_str());
    delete c;
    return MCG_OK;
}
