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


def _run(world, fail="", extra=(), timeout=420, env_extra=None, want_rc=0):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--rehearsal", "--paths", "250000", "--c5-paths", "100001", *extra]
    env = dict(os.environ, MCG_REHEARSAL_FAIL=fail, OMP_NUM_THREADS="1", **(env_extra or {}))
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert (p.returncode == 0) == (want_rc == 0), p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def _ids(n):
    return [float(n), float(n) * (n - 1) / 2.0]


@pytest.mark.parametrize("mode", ["auto", "child", "off"])
def test_eight_ranks_run_the_drivers_command_end_to_end(mode):
    """auto at world size 8 = inline (16 GPU processes would be over the guard measured): the rows run in the rank
    processes, rank 0's guardian holds the headline meanwhile; child = a child process per rank, watched by the parents."""
    world = 8
    out, _ = _run(world, extra=("--c5-rows", mode))
    assert out["rehearsal"] is True and out["value"] is None and out["n_gpus"] == world
    assert out["config"]["c5_rows"] == {"auto": "inline"}.get(mode, mode) and "aborted" not in out
    if mode == "off":
        assert out["extra"]["configs"] is None
        assert [out["ids_counted"], out["ids_summed"]] == _ids(world * 250_000)
        return
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


@pytest.mark.parametrize("mode", ["inline", "child"])
def test_a_rank_failing_inside_an_unbounded_collective_ends_the_rows_not_the_line(mode):
    """a rank that raises inside a pass over the RCCL / torch route cannot meet its peers in an agreement -- they
    sit in the data all-reduce, and its own agreement all-reduce would pair with theirs.  It ends the job instead (exit code
    17).  inline: the launcher takes the peers down, rank 0's guardian prints the line with the rows that were finished
    and says so; child: the parents see the child die, kill the other children and carry on -- exit code 0, an error row."""
    import time
    world, t0 = 4, time.time()
    out, _ = _run(world, fail="pass:2:rccl", extra=("--c5-rows", mode, "--c5-collectives", "shm,rccl,none"), timeout=240,
                  want_rc=0 if mode == "child" else 1)
    assert time.time() - t0 < 120
    assert [out["ids_counted"], out["ids_summed"]] == _ids(world * 250_000)          # the headline is whole
    rows = out["extra"]["configs"]
    assert rows[0]["collective_requested"] == "shm" and [rows[0]["price"], rows[0]["std_err"]] == _ids(world * 100_001)
    if mode == "inline":
        assert len(rows) == 1 and "rank 0 ended before the line was finished" in out["aborted"]
    else:
        assert len(rows) == 2 and "child job" in rows[1]["error"] and "aborted" not in out


def test_a_refused_child_process_costs_the_rows_not_the_line():
    """The pool's process guard (or any limit) refusing a rank's child: that rank reports it, every parent kills its own
    child at the next watch round, the line carries the headline and one row that says what happened."""
    import time
    world, t0 = 4, time.time()
    out, err = _run(world, extra=("--c5-rows", "child"), env_extra={"MCG_BENCH_SPAWN_FAIL": "3"}, timeout=240)
    assert time.time() - t0 < 120
    assert [out["ids_counted"], out["ids_summed"]] == _ids(world * 250_000)
    rows = out["extra"]["configs"]
    assert len(rows) == 1 and "child job ended by its parents" in rows[0]["error"]
    assert "could not be started" in err


def test_the_row_budget_is_kept_by_all_ranks_together():
    """inline rows past their wall-clock budget: every agreement carries "is there time left", so the ranks abandon the
    remaining rows at the same point; the rows finished before stay in the line."""
    out, _ = _run(4, extra=("--c5-rows", "inline", "--c5-budget", "0", "--c5-collectives", "none,shm"))
    rows = out["extra"]["configs"]
    assert len(rows) == 1 and "budget" in rows[0]["error"] and "aborted" not in out


def test_a_communicator_that_never_forms_costs_the_collective_not_the_run():
    """ncclCommInitRank with N > 1 has never run in this repo's history.  bench.py forms the built-in RCCL communicator on a
    scratch context inside a time box first: a rank whose formation never returns (injected) reports so when the box closes,
    every rank takes torch.distributed together, and the job finishes -- the hung scratch thread is a daemon."""
    import time
    world, t0 = 4, time.time()
    out, err = _run(world, fail="rccl_hang:2", extra=("--c5-rows", "off"), env_extra={"MCG_BENCH_RCCL_INIT_LIMIT": "3"}, timeout=120)
    assert time.time() - t0 < 60
    assert out["config"]["collective"].startswith("torch (built-in RCCL init failed")
    assert [out["ids_counted"], out["ids_summed"]] == _ids(world * 250_000)
    assert "did not form within 3 s" in err
    # ... and when it is RANK 0 whose formation hangs: the line says so, and the rank leaves through os._exit after the
    # barrier instead of a tear-down that could block on the stuck thread -- the job still ends with exit code 0
    t0 = time.time()
    out, err = _run(world, fail="rccl_hang:0", extra=("--c5-rows", "off"), env_extra={"MCG_BENCH_RCCL_INIT_LIMIT": "3"}, timeout=120)
    assert time.time() - t0 < 60
    assert out["config"].get("rccl_init_abandoned") is True and out["config"]["collective"].startswith("torch (built-in RCCL init failed")
