"""TEST INFRASTRUCTURE ONLY.  Stand-ins that let bench.py's MULTI-RANK CONTROL FLOW run on CPU ranks (gloo), where no GPU is:
`bench.py --rehearsal` (tests/test_bench_rehearsal.py starts it with eight ranks -- BASELINE.json configs[4]'s world size,
which the pool's limit of six GPU processes per card keeps off the GPU box as processes).  What is real: the process
group, every broadcast / all-reduce / barrier bench.py issues, the collective cascade (ipc -> shm -> rccl -> torch) with
its agreement steps, the child job with its own rendezvous, the node segment's join and barrier protocol (through the
library's host-only hooks, mcg_debug_shm_*).  What is not: there is no device, so nothing is computed or timed --
"prices" are counts of path ids summed over the ranks through whatever collective got installed (so a row proves that
every rank's shard went through that collective exactly once), and the line bench.py prints says "rehearsal": true and
"value": null.  Never imported by the product or by a GPU run.

Failure injection (MCG_REHEARSAL_FAIL = comma-separated):
  shm_init:<rank>     that rank's init_shm raises            -> all ranks fall to rccl together
  rccl_id             rank 0 cannot create the RCCL id       -> all ranks raise together, fall to torch
  rccl_probe:<rank>   that rank cannot load librccl          -> nobody enters ncclCommInitRank, all fall to torch
  rccl_hang:<rank>    that rank's ncclCommInitRank never returns -> the time box (MCG_BENCH_RCCL_INIT_LIMIT) closes, all fall to torch
  pass:<rank>:<want>  that rank raises in the passes of the C5 row <want> -> over the node mailbox (shm, ipc): the row fails on
                      ALL ranks, the next row runs; over the RCCL / torch route (unbounded: the peers cannot leave their
                      all-reduce) the rank ends the job with exit code 17 and the line survives with rank 0's guardian
MCG_BENCH_SPAWN_FAIL = <rank>: that rank's child process of `--c5-rows child` cannot be started (bench.py's own hook)
"""
import ctypes as C
import os
import time

import torch
import torch.distributed as dist

import montecarlooptionspricer_amd as mc


def _fail(tag):
    return tag in [t.strip() for t in os.environ.get("MCG_REHEARSAL_FAIL", "").split(",") if t.strip()]


class Cpu:
    """What bench.py asks of torch.cuda, on a host without one."""
    name = "cpu"

    @staticmethod
    def set_device(_d):
        pass

    @staticmethod
    def synchronize():
        pass

    @staticmethod
    def current_stream_handle():
        return 0

    @staticmethod
    def device_count():
        return 1


class _Matrix:
    def __init__(self, begin, count):
        self.begin, self.count = begin, count

    def free(self):
        pass


class RehearsalEngine:
    """The PathEngine surface bench.py uses.  A 'price' is (number of path ids, sum of path ids) of the shard, summed over
    the ranks by the installed collective."""

    def __init__(self, device=0, stream=None):
        self._L = mc.load_library()
        self._rank = int(os.environ.get("RANK", "0"))
        self._world = 1
        self._kind, self._shm, self._sum = "none", None, None
        self._launches = 0

    # -- collectives -------------------------------------------------------------------------------
    def set_allreduce(self, fn):
        self._detach()
        self._kind, self._sum = ("none", None) if fn is None else ("callback", fn)

    def use_torch_distributed(self, group=None):
        self._detach()

        def s(v):
            t = torch.tensor(v, dtype=torch.float64)
            dist.all_reduce(t, group=group)
            return t.tolist()
        self._kind, self._sum = "callback", s
        self._world = dist.get_world_size()

    def rccl_probe(self):
        if _fail(f"rccl_probe:{self._rank}"):
            raise mc.McgError("injected: cannot dlopen librccl on this rank", 7)

    def init_rccl(self, rank, world, broadcast_bytes):
        uid = None
        if rank == 0:
            uid = b"" if _fail("rccl_id") else b"\x01" * 128
        uid = broadcast_bytes(uid)           # rank 0 ALWAYS enters the broadcast (PathEngine.init_rccl's contract)
        if not uid:
            raise mc.McgError("rank 0 could not create the RCCL id: injected", 7)
        if _fail(f"rccl_hang:{rank}"):       # a communicator that never forms: what bench.py's time box is for
            time.sleep(3600)
        raise mc.McgError("no HIP device: the built-in RCCL communicator cannot form on CPU ranks", 7)

    def init_shm(self, name, rank, world, peer_mailbox=False):
        if _fail(f"shm_init:{rank}"):
            raise mc.McgError("injected: shared-memory communicator unavailable on this rank", 7)
        h = C.c_void_p()
        rc = self._L.mcg_debug_shm_attach(name.encode(), world, rank, 60.0, C.byref(h))   # the REAL join protocol
        if rc:
            raise mc.McgError(self._L.mcg_last_error().decode(), rc)
        self._shm, self._world, self._kind = h, world, "shm"

        def s(v):   # segment barrier on both sides (the real one), the sum itself over gloo
            if self._L.mcg_debug_shm_barrier(self._shm):
                raise mc.McgError(self._L.mcg_last_error().decode(), 7)
            t = torch.tensor(v, dtype=torch.float64)
            dist.all_reduce(t)
            if self._L.mcg_debug_shm_barrier(self._shm):
                raise mc.McgError(self._L.mcg_last_error().decode(), 7)
            return t.tolist()
        self._sum = s
        return False

    def shm_peer_mailbox(self, enable=True):
        return False        # no device memory to put a mailbox in: every rank answers alike, the host mailbox stays

    def comm_info(self):
        seen = self._world if self._kind == "shm" else 0
        return {"kind": self._kind, "n_ranks": self._world if self._kind != "none" else 1,
                "rank": self._rank if self._kind != "none" else 0, "seen_ranks": seen}

    # -- "pricing" ---------------------------------------------------------------------------------
    def gbm(self, seed, S0, r, sigma, dt, n_steps, n_paths, path_begin=0, payoff=None):
        return _Matrix(path_begin, n_paths)

    def rbergomi(self, seed, S0, r, xi, H, eta, rho, dt, n_steps, n_paths, path_begin=0, payoff=None):
        if path_begin % 2:
            raise mc.McgError("rBergomi path_begin must be even", 1)
        return _Matrix(path_begin, n_paths)

    def _price(self, P):
        v = [float(P.count), float(P.count) * (2.0 * P.begin + P.count - 1.0) / 2.0]    # how many ids, their sum
        self._launches += 1
        return tuple(self._sum(v)) if self._sum else tuple(v)

    def price_european(self, P, K, r, T, is_call):
        return self._price(P)

    def price_lsm(self, P, r, K, maturity, dt, is_call, poly_order):
        if _fail(f"pass:{self._rank}:{os.environ.get('MCG_REHEARSAL_ROW', '')}"):
            raise RuntimeError("injected: this rank fails inside the passes of this row")
        return self._price(P)

    # -- the rest of the surface: nothing to do without a device -------------------------------------
    def lsm_one_launch_enabled(self):
        return self._kind in ("none", "shm")

    def timing_enable(self, on=True):
        pass

    def timing_select(self, kernels=None):
        pass

    def timing_reset(self):
        self._launches = 0

    def timing_get(self, kernel):
        return 0.0, self._launches

    def synchronize(self):
        pass

    def trim(self):
        pass

    def generator_clock_arm(self, on=True):
        pass

    def generator_clock(self):
        return None

    def probe_write_ceiling(self, *a, **k):
        return None

    def _detach(self):
        if self._shm is not None:
            self._L.mcg_debug_shm_detach(self._shm)
            self._shm = None

    def close(self):
        self._detach()
