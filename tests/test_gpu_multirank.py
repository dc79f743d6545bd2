"""C5 at its real shard size, and the N>1 PRODUCT path with two (and four) real ranks (run with -m gpu on an MI355X).

  * test_c5_shard_full_size: BASELINE.json configs[4] as one of its eight shards -- rBergomi (H = 0.1, eta = 1.9)
    8M paths x 252 steps, American put, LSM order 2 -- through the per-date kernels every sharded run takes, once
    without and once with a collective installed (a no-op: world size 1).
  * test_two_rank_processes_equal_single_rank: two fresh child processes on GPU 0 (tests/mp_rank_worker.py), gloo
    between them, mcg_set_allreduce on each ctx; European, GBM-LSM and rBergomi-LSM prices of the sharded job must
    equal the single-rank run on the same global path ids (European 1e-12: only the order of three additions
    differs; LSM 1e-9: the regression moments are summed in a different order).
"""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

import montecarlooptionspricer_amd as mc

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SEED, DT = 20251031, 1.0 / 252.0
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)


def test_c5_shard_full_size():
    n, steps = 8_000_000, 252
    T = steps * DT
    a = mc.PathEngine(0)
    P = a.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, steps, n, payoff=(100.0, False))
    eu, ese = a.price_european(P, 100.0, RB["r"], T, False)
    am, ase = a.price_lsm(P, RB["r"], 100.0, T, DT, False, 2)
    assert math.isfinite(am) and math.isfinite(ase) and ase > 0
    assert eu - 3.0 * ese < am < 100.0                      # American put >= European put on the same paths
    c, cse = a.price_european(P, 100.0, RB["r"], T, True)    # martingale through put-call parity on the same paths
    fwd = math.exp(RB["r"] * T) * (c - eu) + 100.0
    assert abs(fwd - 100.0 * math.exp(RB["r"] * T)) <= 2.5 * math.exp(RB["r"] * T) * math.hypot(cse, ese)
    P.free()
    a.trim()

    b = mc.PathEngine(0)
    calls = []
    b.set_allreduce(lambda ptr, count, stream: calls.append(count))   # world size 1: the sum over ranks is the identity
    Q = b.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, steps, n, payoff=(100.0, False))
    eu2, ese2 = b.price_european(Q, 100.0, RB["r"], T, False)
    am2, ase2 = b.price_lsm(Q, RB["r"], 100.0, T, DT, False, 2)
    Q.free()
    b.close()
    a.close()
    assert eu2 == eu and ese2 == ese
    assert abs(am2 - am) <= 1e-9 * am and abs(ase2 - ase) <= 1e-9 * ase
    assert calls.count(8) == steps and calls.count(3) == 2   # 3p+2 moments on each of the 252 dates; payoff + final sums


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["gloo", "shm", "shm_timeout", "shm4"])
def test_two_rank_processes_equal_single_rank(tmp_path, mode):
    """mode "gloo": a host all-reduce callback, the per-date LSM kernels (what RCCL runs use).  mode "shm": the
    library's node-local shared-memory communicator -- each rank's LSM sweep is ONE launch, and the two persistent
    kernels exchange their per-date moments through the device-mapped mailbox while both are resident on the GPU.
    mode "shm_timeout": the same with every hand-shake forced to give up: the ranks agree (sum of their time-out flags)
    to discard the sweep and answer from the per-date kernels over the segment's host all-reduce.  mode "shm4": FOUR ranks
    on the one GPU through the shared-memory communicator (four mailbox rows per round, four persistent grids resident
    together), unequal shards."""
    sys.path.insert(0, HERE)
    from mp_rank_worker import JOBS

    world, port, out = (4 if mode == "shm4" else 2), _free_port(), str(tmp_path / "res.json")
    worker_mode = "shm" if mode == "shm4" else mode
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_rank_worker.py"), str(r), str(world), str(port), out, worker_mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    ranks = [json.load(open(f"{out}.{r}")) for r in range(world)]

    e = mc.PathEngine(0)
    P = e.gbm(SEED, 100.0, 0.04, 0.2, DT, 252, JOBS["euro_paths"], payoff=(100.0, True))
    want_euro = e.price_european(P, 100.0, 0.04, 1.0, True)
    P.free()
    P = e.gbm(SEED, 100.0, 0.04, 0.2, 0.02, JOBS["lsm_steps"], JOBS["lsm_paths"])
    want_lsm = e.price_lsm(P, 0.04, 100.0, 1.0, 0.02, False, 2)
    P.free()
    T = JOBS["rb_steps"] * DT
    P = e.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, JOBS["rb_steps"], JOBS["rb_paths"])
    want_rb = e.price_lsm(P, RB["r"], 100.0, T, DT, False, 2)
    want_rb_eu = e.price_european(P, 100.0, RB["r"], T, False)
    P.free()
    e.close()

    for r, res in enumerate(ranks):                          # every rank holds the GLOBAL price
        for got, want, tol in ((res["euro"], want_euro, 1e-12), (res["rb_euro_put"], want_rb_eu, 1e-12),
                               (res["gbm_lsm"], want_lsm, 1e-9), (res["rb_lsm"], want_rb, 1e-9)):
            assert abs(got[0] - want[0]) <= tol * abs(want[0]), (r, got, want)
            assert abs(got[1] - want[1]) <= max(tol, 1e-9) * abs(want[1]), (r, got, want)
        if mode == "gloo":
            assert res["allreduce_calls"] == {"3": 4, "8": JOBS["lsm_steps"] + JOBS["rb_steps"]}
            assert res["gbm_lsm_sweep_launches"] == JOBS["lsm_steps"] + 2          # one per date + terminal + final sums
        elif mode in ("shm", "shm4"):
            assert res["one_launch_enabled"], "\n".join(logs)                       # no hand-shake ever timed out
            assert res["gbm_lsm_sweep_launches"] == 1 and res["rb_lsm_sweep_launches"] == 1
        else:   # forced time-out: the void sweep (1 launch) is discarded on BOTH ranks, the per-date kernels answer
            assert not res["one_launch_enabled"]
            assert res["gbm_lsm_sweep_launches"] == 1 + JOBS["lsm_steps"] + 2
            assert res["rb_lsm_sweep_launches"] == JOBS["rb_steps"] + 2              # sticky: no second attempt
    assert ranks[0]["shard"][0] == 0 and all(r["shard"][0] % 2 == 0 for r in ranks)
    assert sum(r["shard"][1] for r in ranks) == JOBS["rb_paths"]
