"""EIGHT ranks of one sharded job as eight THREADS of one process on GPU 0 (started by tests/test_gpu_round4.py in a fresh
child process, so that the HIP runtime can be given one hardware queue per rank: HIP maps a process's streams onto
GPU_MAX_HW_QUEUES = 4 hardware queues by default, and two one-launch sweeps that share a queue run one after the other --
each waiting for the other's moments until the hand-shake times out.  On a node every rank thread drives its own GPU and
has that GPU's queues to itself.)

    GPU_MAX_HW_QUEUES=16 python tests/thread_ranks_worker.py <world> <shm|ipc|callback> <out.json> [c5full]

c5full: ONLY the rBergomi American put, at BASELINE.json configs[4]'s full size -- 64M paths x 252 steps over the ranks (8 x
16.2 GB of matrices resident on the one GPU together), per-date route with the all-reduce over the rank threads.
"""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED, DT = 20251031, 1.0 / 252.0
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)
JOBS = dict(euro_paths=300_001, lsm_paths=200_001, lsm_steps=50, rb_paths=100_003, rb_steps=64)


C5_FULL = dict(rb_paths=64_000_000, rb_steps=252)


def run_rank_threads(world, mode, tag, c5_full=False, keep_engines=None):
    """Every rank a thread with a ctx of its own; returns (per-rank result dicts, per-rank all-reduce counts) or raises
    the first rank's error."""
    import numpy as np
    import torch

    import montecarlooptionspricer_amd as mc
    from montecarlooptionspricer_amd import _native as N
    from montecarlooptionspricer_amd.engine import _DevView
    from montecarlooptionspricer_amd.sharding import shard_range

    res, errs = [None] * world, []
    bar = threading.Barrier(world)
    parts = {}
    lock = threading.Lock()
    calls = [[] for _ in range(world)]

    def allreduce_over_threads(rank):
        def fn(ptr, count, _stream):
            t = torch.as_tensor(_DevView(ptr, count), device="cuda:0")
            h = t.cpu().numpy().copy()                   # (waits for the producing kernel: the ctx runs on torch's stream)
            with lock:
                parts[rank] = h
            bar.wait()                                    # all parts are in
            tot = np.zeros(count)
            for r in range(world):
                tot += parts[r]                           # rank order: the same bits on every rank
            bar.wait()                                    # everybody has summed before anybody overwrites its part
            t.copy_(torch.from_numpy(tot))
            calls[rank].append(count)
        return fn

    def work(rank):
        try:
            if mode == "callback":
                torch.cuda.set_device(0)
                e = mc.PathEngine(0, stream=torch.cuda.current_stream().cuda_stream)
                e.set_allreduce(allreduce_over_threads(rank))
                peer = False
            else:
                e = mc.PathEngine(0)
                peer = e.init_shm(f"/mcg_threads_{tag}_{os.getpid()}", rank, world, peer_mailbox=(mode == "ipc"))
            e.timing_enable(True)
            out = {}
            jobs = dict(JOBS, **C5_FULL) if c5_full else JOBS
            if c5_full:
                # C2 at the driver's N = 8 size first: 8 x 10M paths of one stream, one 3-double all-reduce
                b, c = shard_range(80_000_000, rank, world)
                P = e.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, c, path_begin=b, payoff=(100.0, True))
                out["euro"] = e.price_european(P, 100.0, 0.04, 1.0, True)
                P.free()
                e.trim()                                  # (the pool would keep the 20 GB beside the next job's 16)
                b, c = shard_range(jobs["rb_paths"], rank, world, align=2)
                out["shard"] = (b, c)
                T = jobs["rb_steps"] * DT
                P = e.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, jobs["rb_steps"], c, path_begin=b)
                e.timing_reset()
                out["rb_lsm"] = e.price_lsm(P, RB["r"], 100.0, T, DT, False, 2)
                out["rb_lsm_sweep_launches"] = e.timing_get(N.K_LSM_SWEEP)[1]
                out["rb_euro_put"] = e.price_european(P, 100.0, RB["r"], T, False)
                P.free()
                out["comm"] = e.comm_info()
                res[rank] = out
                e.synchronize()
                bar.wait()
                e.close()
                return
            b, c = shard_range(JOBS["euro_paths"], rank, world)
            P = e.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, c, path_begin=b, payoff=(100.0, True))
            out["euro"] = e.price_european(P, 100.0, 0.04, 1.0, True)
            P.free()
            b, c = shard_range(JOBS["lsm_paths"], rank, world)
            P = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, JOBS["lsm_steps"], c, path_begin=b)
            e.timing_reset()
            out["gbm_lsm"] = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
            out["gbm_lsm_sweep_launches"] = e.timing_get(N.K_LSM_SWEEP)[1]
            P.free()
            b, c = shard_range(JOBS["rb_paths"], rank, world, align=2)
            out["shard"] = (b, c)
            T = JOBS["rb_steps"] * DT
            P = e.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, JOBS["rb_steps"], c, path_begin=b)
            e.timing_reset()
            out["rb_lsm"] = e.price_lsm(P, RB["r"], 100.0, T, DT, False, 2)
            out["rb_lsm_sweep_launches"] = e.timing_get(N.K_LSM_SWEEP)[1]
            out["rb_euro_put"] = e.price_european(P, 100.0, RB["r"], T, False)
            P.free()
            out["one_launch_enabled"] = e.lsm_one_launch_enabled()
            out["comm"] = e.comm_info()
            out["peer_mailbox"] = bool(peer)
            res[rank] = out
            e.synchronize()
            bar.wait()                                    # nobody leaves (and frees its mailbox) while a peer may still push into it
            if keep_engines is not None:                  # seqclose: the caller closes the contexts itself, one after another
                keep_engines[rank] = e
                return
            e.close()
        except BaseException as ex:   # noqa: BLE001
            errs.append((rank, ex))
            bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(900 if c5_full else 300)
    if any(t.is_alive() for t in th):
        raise RuntimeError("a rank thread hangs")
    if errs:
        raise errs[0][1]
    return res, calls


def main():
    world, mode, out_path = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    import time

    import montecarlooptionspricer_amd as mc
    what = sys.argv[4] if len(sys.argv) > 4 else ""
    engines = [None] * world if what == "seqclose" else None
    ranks, calls = run_rank_threads(world, mode, f"{mode}{world}", c5_full=what == "c5full", keep_engines=engines)
    close_seconds = None
    if engines is not None:   # ONE thread closes the rank threads' contexts one after another (a loop over engines, the garbage
        close_seconds = []    # collector, atexit): every owner but the last still has borrowers when it goes
        for e in engines:
            t0 = time.perf_counter()
            e.close()
            close_seconds.append(time.perf_counter() - t0)
    with open(out_path, "w") as f:
        json.dump({"ranks": ranks, "calls": [{"1": c.count(1), "3": c.count(3), "8": c.count(8)} for c in calls], "stats": mc.stats(),
                   "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "close_seconds": close_seconds}, f)


if __name__ == "__main__":
    main()
