"""bench.py's N > 1 CONTROL FLOW at world size 8 -- BASELINE.json configs[4]'s, the only N north_star names -- rehearsed
on CPU ranks (`bench.py --rehearsal`, gloo; tests/bench_rehearsal.py stands in for the GPU and the engine).  Eight rank
PROCESSES cannot share the GPU box's card (the pool allows six GPU processes), and each rank of the driver's command
starts a child job on top: what a one-GPU box can show of world size 8 is (a) the kernels and mailboxes with eight rank
THREADS (tests/test_gpu_round4.py) and (b) this: the driver's very command line with eight processes, every broadcast,
agreement and barrier of the script, the collective cascade, the child job and its rendezvous, the node segment joined by
eight processes -- and, injected, the failures that used to leave ranks in different collectives."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, fail="", extra=(), timeout=420):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--rehearsal", "--paths", "250000", "--c5-paths", "100001", *extra]
    env = dict(os.environ, MCG_REHEARSAL_FAIL=fail, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def _ids(n):
    return [float(n), float(n) * (n - 1) / 2.0]


def test_eight_ranks_run_the_drivers_command_end_to_end():
    world = 8
    out, _ = _run(world)
    assert out["rehearsal"] is True and out["value"] is None and out["n_gpus"] == world
    # C2: every rank's shard of 8 x 250 000 ids went through the collective exactly once (here: torch's, the built-in RCCL
    # communicator cannot form without a device -- and every rank took that step down together)
    assert [out["ids_counted"], out["ids_summed"]] == _ids(world * 250_000)
    assert out["config"]["collective"].startswith("torch (built-in RCCL init failed")
    rows = out["extra"]["configs"]
    assert [r["collective_requested"] for r in rows] == ["none", "shm", "ipc", "rccl"]
    total = world * 100_001          # odd per-rank count: unequal, even-aligned shards
    for r in rows:
        assert "error" not in r, r
        assert r["global_paths"] == total
    alone, shm, ipc, rccl = rows
    assert alone["comm"]["kind"] == "none" and alone["price"] < total          # a local count only
    for r in (shm, ipc, rccl):
        assert [r["price"], r["std_err"]] == _ids(total), r                      # all eight shards, each once, no overlap
    assert shm["collective"] == "shm" and shm["comm"]["n_ranks"] == world and shm["comm"]["seen_ranks_min_over_ranks"] == world
    assert ipc["collective"].startswith("shm (peer-memory mailbox unavailable") and ipc["comm"]["seen_ranks_min_over_ranks"] == world
    assert rccl["collective"].startswith("torch (built-in RCCL init failed")


def test_injected_set_up_failures_keep_the_ranks_in_step():
    """The set-up races VERDICT r3 named (weak #4): one rank cannot join the segment AND rank 0 cannot create the RCCL id.
    Rank 2's failure sends ALL ranks from shm to the RCCL route together; there rank 0 still enters the id broadcast (with
    an empty id), every rank raises, all fall to torch.distributed -- and the job ENDS within seconds with every shard
    counted once, not after a 300-s process-group time-out."""
    import time
    world, t0 = 4, time.time()
    out, _ = _run(world, fail="shm_init:2,rccl_id", extra=("--c5-collectives", "shm,rccl"), timeout=240)
    assert time.time() - t0 < 200
    rows = out["extra"]["configs"]
    assert [r["collective_requested"] for r in rows] == ["shm", "rccl"]
    for r in rows:
        assert "error" not in r and r["collective"].startswith("torch (built-in RCCL init failed"), r
        assert [r["price"], r["std_err"]] == _ids(world * 100_001)
    assert out["config"]["collective"].startswith("torch (built-in RCCL init failed")      # the C2 loop's collective too


def test_injected_row_failure_and_missing_librccl_keep_the_ranks_in_step():
    """ADVICE r3's row-level race and the librccl probe: rank 3 raises inside the passes of the shm row -- the row is
    recorded as failed on ALL ranks at once and the next row runs; rank 1 cannot load librccl -- the ranks agree BEFORE
    anybody enters ncclCommInitRank (where a rank whose peer never arrives would wait) and take torch together."""
    import time
    world, t0 = 4, time.time()
    out, _ = _run(world, fail="pass:3:shm,rccl_probe:1", extra=("--c5-collectives", "shm,rccl"), timeout=240)
    assert time.time() - t0 < 200
    rows = out["extra"]["configs"]
    assert "error" in rows[0] and "error" not in rows[1], rows
    assert rows[1]["collective"].startswith("torch (built-in RCCL init failed")
    assert [rows[1]["price"], rows[1]["std_err"]] == _ids(world * 100_001)
