// One process, one host thread per rank -- the shape of the reference's own driver (an OpenMP parallel region,
// src/core/PredictionGen.cpp:542-546) carried over to a multi-GPU node: every thread owns an mcg_ctx on its GPU
// (rank % device count: on a one-GPU box all ranks share GPU 0), the ranks join one node segment (mcg_comm_init_shm; with
// argv[2] = "ipc" the in-kernel mailbox moves into device memory, same-process peers exchanging pointers instead of HIP
// IPC handles), each generates its shard of ONE Philox stream and prices it: European (one 3-double sum over the ranks)
// and American by Longstaff-Schwartz (the per-date moments exchanged INSIDE the one launch each rank's sweep is).
// Every rank must hold the price a single context computes on all paths.  Plain g++ against include/mcgpu.h.
//
//   GPU_MAX_HW_QUEUES=16 build/thread_ranks_driver <ranks> [shm|ipc]
// (ranks that SHARE a device need a hardware queue each: HIP maps a process's streams onto 4 by default, and two sweeps
//  in one queue would run one after the other, each waiting for the other's moments; one rank per GPU needs nothing.)
#include <omp.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mcgpu.h"

static void shard(long n, int rank, int world, int align, long* begin, long* count) {   // sharding.shard_range
    const long units = (n + align - 1) / align, base = units / world, rem = units % world;
    const long u0 = rank * base + (rank < rem ? rank : rem), u1 = u0 + base + (rank < rem ? 1 : 0);
    *begin = u0 * align < n ? u0 * align : n;
    *count = (u1 * align < n ? u1 * align : n) - *begin;
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 8;
    const bool ipc = argc > 2 && std::strcmp(argv[2], "ipc") == 0;
    const long n_gbm = 400001, n_rb = 120003;
    const int steps_gbm = 50, steps_rb = 64;
    const double dt = 1.0 / 252.0;
    int n_dev = 0;
    if (mcg_device_count(&n_dev) != MCG_OK || n_dev < 1) {
        std::printf("FAILED no device: %s\n", mcg_last_error());
        return 1;
    }
    // the single-context answers
    double want[3][2];
    {
        mcg_ctx* c = nullptr;
        mcg_paths* P = nullptr;
        if (mcg_init(&c, 0) != MCG_OK) return 1;
        mcg_paths_gbm_payoff(c, 7, 100.0, 0.04, 0.2, dt, 252, 0, n_gbm, 100.0, 1, &P);
        mcg_price_european(c, P, 100.0, 0.04, 1.0, 1, &want[0][0], &want[0][1]);
        mcg_paths_free(P);
        mcg_paths_gbm(c, 7, 100.0, 0.04, 0.2, 0.02, steps_gbm, 0, n_gbm, &P);
        mcg_price_lsm(c, P, 0.04, 100.0, 1.0, 0.02, 0, 2, &want[1][0], &want[1][1]);
        mcg_paths_free(P);
        mcg_paths_rbergomi(c, 7, 100.0, 0.04, 0.04, 0.1, 1.9, -0.9, dt, steps_rb, 0, n_rb, &P);
        mcg_price_lsm(c, P, 0.04, 100.0, steps_rb * dt, dt, 0, 2, &want[2][0], &want[2][1]);
        mcg_paths_free(P);
        mcg_finalize(c);
    }
    const std::string seg = "/mcg_cpp_threads_" + std::to_string((long)getpid());
    std::vector<double> got(6 * world, 0.0);
    std::vector<int> kind(world, -1), seen(world, 0), one_launch(world, 0);
    int failures = 0;
#pragma omp parallel num_threads(world) reduction(+ : failures)
    {
        const int rank = omp_get_thread_num();
        mcg_ctx* c = nullptr;
        mcg_paths* P = nullptr;
        int active = 0;
        long b, n;
        double* g = &got[6 * rank];
        bool ok = mcg_init(&c, rank % n_dev) == MCG_OK && mcg_comm_init_shm(c, seg.c_str(), world, rank) == MCG_OK;
        if (ok && ipc) ok = mcg_comm_shm_peer_mailbox(c, 1, &active) == MCG_OK;
        if (ok) {
            shard(n_gbm, rank, world, 1, &b, &n);
            ok = mcg_paths_gbm_payoff(c, 7, 100.0, 0.04, 0.2, dt, 252, (uint64_t)b, n, 100.0, 1, &P) == MCG_OK &&
                 mcg_price_european(c, P, 100.0, 0.04, 1.0, 1, &g[0], &g[1]) == MCG_OK;
            mcg_paths_free(P);
        }
        if (ok) {
            ok = mcg_paths_gbm(c, 7, 100.0, 0.04, 0.2, 0.02, steps_gbm, (uint64_t)b, n, &P) == MCG_OK &&
                 mcg_price_lsm(c, P, 0.04, 100.0, 1.0, 0.02, 0, 2, &g[2], &g[3]) == MCG_OK;
            mcg_paths_free(P);
        }
        if (ok) {
            shard(n_rb, rank, world, 2, &b, &n);   // rBergomi paths come in pairs: even-aligned shards
            ok = mcg_paths_rbergomi(c, 7, 100.0, 0.04, 0.04, 0.1, 1.9, -0.9, dt, steps_rb, (uint64_t)b, n, &P) == MCG_OK &&
                 mcg_price_lsm(c, P, 0.04, 100.0, steps_rb * dt, dt, 0, 2, &g[4], &g[5]) == MCG_OK;
            mcg_paths_free(P);
        }
        if (!ok) {
            std::printf("rank %d FAILED: %s\n", rank, mcg_last_error());
            ++failures;
        } else {
            int nr = 0, rk = 0;
            mcg_comm_info(c, &kind[rank], &nr, &rk, &seen[rank]);
            mcg_lsm_one_launch_enabled(c, &one_launch[rank]);
            if (nr != world || rk != rank) ++failures;
        }
#pragma omp barrier   // nobody frees its mailbox while a peer may still push into it
        if (c) mcg_finalize(c);
    }
    for (int r = 0; r < world && !failures; ++r) {
        const double tol[3] = {1e-12, 1e-9, 1e-9};
        for (int k = 0; k < 3; ++k) {
            const double p = got[6 * r + 2 * k], se = got[6 * r + 2 * k + 1];
            if (!(std::fabs(p - want[k][0]) <= tol[k] * want[k][0]) || !(std::fabs(se - want[k][1]) <= 1e-9 * want[k][1])) {
                std::printf("rank %d job %d: %.15g +- %.15g, single context %.15g +- %.15g\n", r, k, p, se, want[k][0], want[k][1]);
                ++failures;
            }
            if (p != got[2 * k]) ++failures;   // the same bits on every rank
        }
        if (kind[r] != (ipc ? 4 : 3) || seen[r] != world || !one_launch[r]) {
            std::printf("rank %d: comm kind %d (want %d), seen %d, one-launch %d\n", r, kind[r], ipc ? 4 : 3, seen[r], one_launch[r]);
            ++failures;
        }
    }
    mcg_stats_t st;
    mcg_stats(&st, 0);
    std::printf("european %.10f lsm %.10f rbergomi_lsm %.10f | one-launch sweeps %lld time-outs %lld barrier failures %lld peer mailboxes %lld\n",
                got[0], got[2], got[4], (long long)st.lsm_one_launch_sweeps, (long long)st.lsm_one_launch_timeouts,
                (long long)st.shm_barrier_failures, (long long)st.peer_mailbox_enabled);
    if (st.lsm_one_launch_timeouts != 0 || st.shm_barrier_failures != 0) ++failures;
    if (failures) {
        std::printf("FAILED %d\n", failures);
        return 1;
    }
    std::printf("OK ranks=%d mailbox=%s\n", world, ipc ? "peer memory" : "host");
    return 0;
}
