"""One rank of a sharded pricing job on the PRODUCT path (libmcgpu through the C ABI): started as a fresh child
process by tests/test_gpu_multirank.py, once per rank, all ranks on GPU 0.

    python tests/mp_rank_worker.py <rank> <world> <port> <out.json> [gloo|shm|shm_timeout|ipc]

gloo (two ranks cannot share one device under RCCL): the callback installed with mcg_set_allreduce copies the handful
of doubles to the host, all-reduces there and copies back, stream-ordered on the ctx's stream (= torch's current
stream).  Everything else -- shard ranges, kernels, the per-date reduce / all-reduce / solve sequence of
csrc/kernels_lsm.hip -- is exactly what an N-GPU run executes.
shm: the library's node-local shared-memory communicator (mcg_comm_init_shm; no torch.distributed at all): host
all-reduce of the sums, and the LSM sweeps run as ONE launch per rank whose reducing workgroups exchange the per-date
moments through the device-mapped mailbox -- here with both ranks' persistent kernels resident on the same GPU.
ipc: the same with the mailbox in device memory, each rank's copy mapped into the peers by HIP IPC
(mcg_comm_shm_peer_mailbox): a rank pushes its moments into every copy and polls its own.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED, DT = 20251031, 1.0 / 252.0
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)
JOBS = dict(euro_paths=300_001, lsm_paths=200_001, lsm_steps=50, rb_paths=100_003, rb_steps=64)


def near_degenerate_matrix(seed, n_total, n_itm, base, spread, K=100.0):
    """[n_total][3] price matrix (path-major) whose middle date has exactly n_itm in-the-money paths (put), within a
    relative spread `spread` of `base`: the reference's rank threshold decides that date's fit (tests/test_gpu_parity.py
    has the single-GPU versions of this case)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    m = np.empty((n_total, 3))
    m[:, 0] = 1.05 * K
    m[:, 1] = K * (1.02 + 0.2 * rs.rand(n_total))
    m[:, 2] = K * (0.7 + 0.5 * rs.rand(n_total))
    idx = rs.choice(n_total, n_itm, replace=False)
    m[idx, 1] = base * (1.0 + spread * rs.uniform(-1.0, 1.0, n_itm))
    return m


DEGENERATE = [(11, 5000, 3, 90.0, 1e-4), (12, 5000, 4, 99.9, 1e-6), (13, 5000, 2, 60.0, 1e-3)]   # (seed, paths, itm, base, spread)


def main() -> None:
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    import torch

    import montecarlooptionspricer_amd as mc
    from montecarlooptionspricer_amd import _native as N
    from montecarlooptionspricer_amd.engine import _DevView
    from montecarlooptionspricer_amd.sharding import shard_range

    calls = []
    dist = None
    if mode == "gloo":
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        eng = mc.PathEngine(0, stream=torch.cuda.current_stream().cuda_stream)

        def allreduce(ptr, count, _stream):
            t = torch.as_tensor(_DevView(ptr, count), device="cuda:0")
            h = t.cpu()                      # waits for the producing kernel on the shared stream
            dist.all_reduce(h)
            t.copy_(h)
            calls.append(count)

        eng.set_allreduce(allreduce)
    else:
        eng = mc.PathEngine(0)
        if mode == "shm_timeout":       # every hand-shake gives up at once: the ranks must agree to fall back together
            eng.debug_lsm_hooks(spin_limit=0)
        peer = eng.init_shm(f"/mcg_test_{port}", rank, world, peer_mailbox=(mode == "ipc"))
    eng.timing_enable(True)
    res = {}

    b, c = shard_range(JOBS["euro_paths"], rank, world)
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, c, path_begin=b, payoff=(100.0, True))
    res["euro"] = eng.price_european(P, 100.0, 0.04, 1.0, True)
    P.free()

    b, c = shard_range(JOBS["lsm_paths"], rank, world)
    P = eng.gbm(SEED, 100.0, 0.04, 0.2, 0.02, JOBS["lsm_steps"], c, path_begin=b)
    eng.timing_reset()
    res["gbm_lsm"] = eng.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
    res["gbm_lsm_sweep_launches"] = eng.timing_get(N.K_LSM_SWEEP)[1]
    P.free()

    b, c = shard_range(JOBS["rb_paths"], rank, world, align=2)
    T = JOBS["rb_steps"] * DT
    P = eng.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, JOBS["rb_steps"], c, path_begin=b)
    eng.timing_reset()
    res["rb_lsm"] = eng.price_lsm(P, RB["r"], 100.0, T, DT, False, 2)
    res["rb_lsm_sweep_launches"] = eng.timing_get(N.K_LSM_SWEEP)[1]
    res["one_launch_enabled"] = eng.lsm_one_launch_enabled()
    res["rb_euro_put"] = eng.price_european(P, 100.0, RB["r"], T, False)
    P.free()

    # dates the first solve does not trust, sharded: the re-fit needs a second, data-dependent reduction over the ranks
    # (one more mailbox round inside the one launch; one more launch + all-reduce on the per-date route)
    res["degenerate_lsm"] = []
    for seed, n_total, n_itm, base, spread in DEGENERATE:
        m = near_degenerate_matrix(seed, n_total, n_itm, base, spread)
        b, c = shard_range(n_total, rank, world)
        P = eng.from_host(m[b:b + c])
        res["degenerate_lsm"].append(eng.price_lsm(P, 0.04, 100.0, 1.0, 0.5, False, 2)[0])
        P.free()
    b, c = shard_range(JOBS["rb_paths"], rank, world, align=2)

    res["comm"] = eng.comm_info()
    res["peer_mailbox"] = bool(peer) if mode != "gloo" else False
    if mode == "ipc":   # back to the host mailbox (collective over the ranks) and once more: the same bits
        assert eng.shm_peer_mailbox(False) is False
        res["comm_after_disable"] = eng.comm_info()
        b, c = shard_range(JOBS["lsm_paths"], rank, world)
        P = eng.gbm(SEED, 100.0, 0.04, 0.2, 0.02, JOBS["lsm_steps"], c, path_begin=b)
        res["gbm_lsm_host_mailbox"] = eng.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
        P.free()
        b, c = shard_range(JOBS["rb_paths"], rank, world, align=2)
    res["allreduce_calls"] = {"3": calls.count(3), "8": calls.count(8)}
    res["stats"] = mc.stats()   # this process's event counters: time-outs, fall-backs, re-fits
    res["shard"] = [b, c]
    eng.close()
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(res, f)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
