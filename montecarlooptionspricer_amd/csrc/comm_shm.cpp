// Node-local collective over POSIX shared memory: one process per GPU, all on one host (the north-star's shape: eight
// MI355X of one node).  Two services on one segment:
//
//   * a HOST all-reduce of a few doubles (payoff sums, the per-date moments of the per-date LSM kernels): copy down,
//     publish in the segment, barrier, sum in rank order, barrier, copy up -- the mcg_allreduce_fn of this ctx;
//   * a DEVICE mailbox for the one-launch LSM sweeps (kernels_lsm.hip): the segment is registered with HIP and every
//     GPU's reducing workgroup writes its local regression moments into its row of the round's slot and polls the other
//     ranks' rows with system-scope loads over PCIe -- the per-date exchange between GPUs happens INSIDE the kernel,
//     no launch, no host round trip, no RCCL kernel competing for the CUs the sweep occupies.  Rows are written once
//     per sweep (one slot per exchange round, no recycling); each rank re-arms its own rows with the reserved NaN
//     before a sweep and a host barrier orders that against everybody's launch.
//
// Every rank computes the same global moments (same published values, summed in rank order), hence bit-identical
// coefficients and the same refinement decisions, so the ranks stay in lock-step without further agreement.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <new>
#include <string>

#include "mcg_internal.hpp"

namespace mcg {

struct ShmHeader {
    std::atomic<uint32_t> magic;
    uint32_t n_ranks;
    std::atomic<uint32_t> arrive;
    std::atomic<uint32_t> sense;
    std::atomic<uint32_t> attached;
    uint32_t pad[11];
    double host_slots[SHM_MAX_RANKS][32];
};
static_assert(sizeof(ShmHeader) % 64 == 0, "mailbox starts cache-line aligned");

struct ShmComm {
    std::string name;
    int fd = -1;
    void* base = nullptr;
    size_t bytes = 0;
    ShmHeader* hdr = nullptr;
    double* mbox_host = nullptr;
    double* mbox_dev = nullptr;
    bool registered = false;
    uint32_t local_sense = 0;
    int n_ranks = 1, rank = 0;
    double* pinned = nullptr;  // 32 doubles
};

namespace {

constexpr uint32_t SHM_MAGIC = 0x4D434753u;  // "MCGS"
constexpr double SHM_TIMEOUT_S = 120.0;

size_t shm_bytes() { return sizeof(ShmHeader) + (size_t)SHM_MAX_ROUNDS * SHM_MAX_RANKS * SHM_ROW_DOUBLES * sizeof(double); }

bool shm_barrier(ShmComm* c) {
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
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > SHM_TIMEOUT_S) return false;
        }
    }
    return true;
}

int shm_allreduce(void* user, double* buf, int count, void* stream) {
    mcg_ctx* ctx = (mcg_ctx*)user;
    ShmComm* c = ctx->shm;
    if (!c || count < 0 || count > 31) {  // (entry 31 of a rank's slot row carries shm_sum_flag's integer)
        set_error("shared-memory all-reduce: bad count %d", count);
        return 1;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(c->pinned, buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        set_error("shared-memory all-reduce: device copy failed");
        return 1;
    }
    std::memcpy(c->hdr->host_slots[c->rank], c->pinned, (size_t)count * sizeof(double));
    if (!shm_barrier(c)) {
        set_error("shared-memory all-reduce: a rank did not arrive within %.0f s", SHM_TIMEOUT_S);
        return 1;
    }
    for (int i = 0; i < count; ++i) {
        double s = 0.0;
        for (int r = 0; r < c->n_ranks; ++r) s += c->hdr->host_slots[r][i];  // rank order: the same bits on every rank
        c->pinned[i] = s;
    }
    if (!shm_barrier(c)) {  // nobody overwrites its slot before everyone has read it
        set_error("shared-memory all-reduce: a rank did not arrive within %.0f s", SHM_TIMEOUT_S);
        return 1;
    }
    if (hipMemcpyAsync(buf, c->pinned, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        set_error("shared-memory all-reduce: device copy failed");
        return 1;
    }
    return 0;
}

}  // namespace

// Re-arm this rank's rows of the first `rounds` slots and wait until every rank has done so (before a sweep's launch).
int shm_arm_mailbox(mcg_ctx* ctx, int rounds, uint64_t sentinel_bits) {
    ShmComm* c = ctx->shm;
    if (!c) return fail(MCG_ERR_COMM, "no shared-memory communicator");
    if (rounds > SHM_MAX_ROUNDS) return fail(MCG_ERR_INVALID, "too many exchange rounds for the mailbox");
    for (int q = 0; q < rounds; ++q) {
        uint64_t* row = reinterpret_cast<uint64_t*>(c->mbox_host + ((size_t)q * SHM_MAX_RANKS + c->rank) * SHM_ROW_DOUBLES);
        for (int t = 0; t < SHM_ROW_DOUBLES; ++t) row[t] = sentinel_bits;
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (!shm_barrier(c)) return fail(MCG_ERR_COMM, "shared-memory barrier timed out");
    return MCG_OK;
}

// sum of one integer over the ranks (agreement on "did any rank's hand-shake time out")
int shm_sum_flag(mcg_ctx* ctx, int flag, int* total) {
    ShmComm* c = ctx->shm;
    if (!c) return fail(MCG_ERR_COMM, "no shared-memory communicator");
    c->hdr->host_slots[c->rank][31] = (double)flag;
    if (!shm_barrier(c)) return fail(MCG_ERR_COMM, "shared-memory barrier timed out");
    double s = 0.0;
    for (int r = 0; r < c->n_ranks; ++r) s += c->hdr->host_slots[r][31];
    if (!shm_barrier(c)) return fail(MCG_ERR_COMM, "shared-memory barrier timed out");
    *total = (int)s;
    return MCG_OK;
}

double* shm_mailbox_device(mcg_ctx* ctx) { return ctx->shm ? ctx->shm->mbox_dev : nullptr; }
int shm_rank(mcg_ctx* ctx) { return ctx->shm ? ctx->shm->rank : 0; }
int shm_n_ranks(mcg_ctx* ctx) { return ctx->shm ? ctx->shm->n_ranks : 1; }

void shm_release(mcg_ctx* ctx) {
    ShmComm* c = ctx->shm;
    if (!c) return;
    if (ctx->allreduce == shm_allreduce) {
        ctx->allreduce = nullptr;
        ctx->allreduce_user = nullptr;
    }
    if (c->registered) (void)hipHostUnregister(c->base);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->base) munmap(c->base, c->bytes);
    if (c->fd >= 0) close(c->fd);
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
    const auto t0 = std::chrono::steady_clock::now();
    auto waited = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (rank == 0) {
        shm_unlink(name);  // a stale segment of a crashed run
        c->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (c->fd < 0 || ftruncate(c->fd, (off_t)c->bytes) != 0) {
            shm_release(ctx);
            return fail(MCG_ERR_COMM, "cannot create shared-memory segment %s", name);
        }
    } else {
        for (;;) {
            c->fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (c->fd >= 0 && fstat(c->fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break;
            if (c->fd >= 0) close(c->fd);
            c->fd = -1;
            if (waited() > SHM_TIMEOUT_S) {
                shm_release(ctx);
                return fail(MCG_ERR_COMM, "shared-memory segment %s did not appear", name);
            }
            usleep(2000);
        }
    }
    c->base = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
    if (c->base == MAP_FAILED) {
        c->base = nullptr;
        shm_release(ctx);
        return fail(MCG_ERR_COMM, "mmap of %s failed", name);
    }
    c->hdr = reinterpret_cast<ShmHeader*>(c->base);
    c->mbox_host = reinterpret_cast<double*>(reinterpret_cast<char*>(c->base) + sizeof(ShmHeader));
    if (rank == 0) {
        c->hdr->n_ranks = (uint32_t)n_ranks;
        c->hdr->arrive.store(0);
        c->hdr->sense.store(0);
        c->hdr->attached.store(0);
        c->hdr->magic.store(SHM_MAGIC, std::memory_order_release);
    } else {
        while (c->hdr->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
            if (waited() > SHM_TIMEOUT_S) {
                shm_release(ctx);
                return fail(MCG_ERR_COMM, "shared-memory segment %s was never initialised", name);
            }
            usleep(1000);
        }
        if (c->hdr->n_ranks != (uint32_t)n_ranks) {
            shm_release(ctx);
            return fail(MCG_ERR_COMM, "segment %s was created for %u ranks, not %d", name, c->hdr->n_ranks, n_ranks);
        }
    }
    if (hipHostRegister(c->base, c->bytes, hipHostRegisterMapped) != hipSuccess) {
        (void)hipGetLastError();
        shm_release(ctx);
        return fail(MCG_ERR_HIP, "hipHostRegister of the shared segment failed");
    }
    c->registered = true;
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, c->mbox_host, 0) != hipSuccess || hipHostMalloc((void**)&c->pinned, 32 * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        shm_release(ctx);
        return fail(MCG_ERR_HIP, "device mapping of the shared segment failed");
    }
    c->mbox_dev = (double*)dev;
    ctx->allreduce = shm_allreduce;
    ctx->allreduce_user = ctx;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    // everybody attached (the creator may unlink the name only at finalize; a late rank still needs it until here)
    c->hdr->attached.fetch_add(1);
    while (c->hdr->attached.load() < (uint32_t)n_ranks) {
        if (waited() > SHM_TIMEOUT_S) {
            shm_release(ctx);
            return fail(MCG_ERR_COMM, "only %u of %d ranks attached to %s", c->hdr->attached.load(), n_ranks, name);
        }
        usleep(1000);
    }
    return MCG_OK;
}
