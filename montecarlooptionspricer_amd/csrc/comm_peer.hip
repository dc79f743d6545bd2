// In-kernel ping over the peer-memory mailbox (comm_shm.cpp: mcg_comm_shm_peer_mailbox): before the one-launch LSM
// sweeps are allowed to exchange their per-date moments through mailboxes in the peers' HBM, every rank proves -- with
// the very access pattern the sweeps use -- that a system-scope store into a peer's mapping becomes visible to that
// peer's system-scope polls of its own memory while both kernels are running.  One wavefront per rank; lane r < n pushes
// this rank's token into rank r's mailbox and polls this rank's own mailbox for rank r's token.  Every spin is bounded:
// a mapping that does not behave costs 0.2 s and sends all ranks back to the host mailbox, never a hang.
#include "mcg_internal.hpp"

namespace mcg {

struct PeerPingArgs {
    double* peer[SHM_MAX_RANKS];
    int n, rank;
    unsigned spin_limit;
    unsigned* failed;
};

__global__ __launch_bounds__(64) void k_peer_ping(PeerPingArgs a) {
    const int r = threadIdx.x;
    if (r >= a.n) return;
    const uint64_t sentinel = 0xFFF85EA7FFF85EA7ull;
    // slot 0 of the mailbox, row = sender, entry 0
    double* dst = a.peer[r] + (size_t)a.rank * SHM_ROW_DOUBLES;
    const double* src = a.peer[a.rank] + (size_t)r * SHM_ROW_DOUBLES;
    __hip_atomic_store(dst, 1000.0 + a.rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned spins = 0;
    for (;;) {
        const double v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((uint64_t)__double_as_longlong(v) != sentinel) {
            if (v != 1000.0 + r) __hip_atomic_store(a.failed, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
        if (++spins > a.spin_limit) {
            __hip_atomic_store(a.failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// Re-arm the rows of ranks 0 .. n-1 in the first `rounds` slots of a mailbox with the reserved NaN: one small launch
// (a hipMemsetD32Async over the slots' full width -- 16 ranks' rows -- is split by the runtime into ~80 fill kernels).
__global__ __launch_bounds__(256) void k_peer_arm(unsigned long long* mbox, int rounds, int n, unsigned long long sentinel) {
    const int per_round = n * SHM_ROW_DOUBLES;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rounds * per_round) return;
    const int64_t q = i / per_round, e = i % per_round;
    mbox[q * (SHM_MAX_RANKS * SHM_ROW_DOUBLES) + e] = sentinel;
}

int peer_arm(mcg_ctx* ctx, double* mbox, int rounds, int n, uint64_t sentinel) {
    const int64_t total = (int64_t)rounds * n * SHM_ROW_DOUBLES;
    hipLaunchKernelGGL(k_peer_arm, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<unsigned long long*>(mbox), rounds, n, (unsigned long long)sentinel);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    return 0;
}

// returns 0 when every peer's token arrived, non-zero otherwise (time-out, wrong token, or a HIP error)
int peer_ping(mcg_ctx* ctx, double* const* peers, int n, int rank) {
    PeerPingArgs a;
    for (int r = 0; r < SHM_MAX_RANKS; ++r) a.peer[r] = r < n ? peers[r] : nullptr;
    a.n = n;
    a.rank = rank;
    a.spin_limit = 400000;  // ~0.2 s of 0.5-us polls
    a.failed = reinterpret_cast<unsigned*>(ctx->scalars + SC_BARRIER);
    if (hipMemsetAsync(a.failed, 0, sizeof(unsigned), ctx->stream) != hipSuccess) return 3;
    hipLaunchKernelGGL(k_peer_ping, dim3(1), dim3(64), 0, ctx->stream, a);
    if (hipGetLastError() != hipSuccess) return 3;
    if (hipMemcpyAsync(ctx->h_scalars + SC_BARRIER, a.failed, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        return 3;
    }
    return (int)reinterpret_cast<const unsigned*>(ctx->h_scalars + SC_BARRIER)[0];
}

}  // namespace mcg
