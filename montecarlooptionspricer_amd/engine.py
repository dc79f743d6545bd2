"""PathEngine: explicit-parameter host API over the C ABI (include/mcgpu.h).

This is the (seed, S0, K, r, sigma | xi, H, eta, T, N_paths, N_steps) interface the north-star
names; the reference hard-codes these as literals (SURVEY.md section 5 "Config / flags").
Device memory is owned by libmcgpu; torch is only used (optionally) for the stream and for the
cross-rank all-reduce.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Callable, Optional, Tuple

import numpy as np

from . import _native as N
from ._native import McgError, check


class _DevView:
    """Raw device pointer exposed through __cuda_array_interface__ (zero-copy torch view)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 3, "strides": None}


class PathMatrix:
    """Device-resident (n_steps+1) x n_paths price matrix, step-major (mcg_paths*)."""

    def __init__(self, engine: "PathEngine", handle: C.c_void_p):
        self._engine = engine
        self._h = handle
        n_paths, n_steps, ld, ptr = C.c_int64(), C.c_int(), C.c_int64(), C.c_void_p()
        check(engine._L.mcg_paths_info(handle, C.byref(n_paths), C.byref(n_steps), C.byref(ld), C.byref(ptr)))
        self.n_paths, self.n_steps, self.ld = n_paths.value, n_steps.value, ld.value
        self.device_ptr = ptr.value or 0
        engine._live.add(self)

    @property
    def nbytes_algorithmic(self) -> int:
        """8*(n_steps+1) bytes per path: the figure the HBM-write roofline is computed from."""
        return 8 * (self.n_steps + 1) * self.n_paths

    def to_host(self) -> np.ndarray:
        """Reference layout: [n_paths][n_steps+1] (what GenerateStockPricePaths returns)."""
        self._alive()
        out = np.empty((self.n_paths, self.n_steps + 1), dtype=np.float64)
        check(self._engine._L.mcg_paths_to_host(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def to_host_step_major(self) -> np.ndarray:
        """As stored: [n_steps+1][n_paths]."""
        self._alive()
        out = np.empty((self.n_steps + 1, self.n_paths), dtype=np.float64)
        check(self._engine._L.mcg_paths_to_host_step_major(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def free(self) -> None:
        """Give the device buffer back to the engine's pool.  Safe in any order with PathEngine.close(): close()
        frees the matrices that are still alive first, so a late free() (or the garbage collector) finds nothing
        left to do."""
        h, self._h = self._h, None
        if h is not None:
            self._engine._live.discard(self)
            self._engine._L.mcg_paths_free(h)

    def _alive(self):
        if self._h is None:
            raise McgError("PathMatrix already freed")

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PathEngine:
    """One context (device + stream + workspace); create one per process/GPU or per host thread."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._L = N.load_library()
        self._ctx = C.c_void_p()
        if stream is None:
            check(self._L.mcg_init(C.byref(self._ctx), int(device)))
        else:  # adopt the caller's stream; 0 is the legacy default stream (torch's default)
            check(self._L.mcg_init_on_stream(C.byref(self._ctx), int(device), C.c_void_p(int(stream))))
        self._cb = None  # keep the ctypes callback alive
        self.device = device
        self._live = weakref.WeakSet()  # PathMatrix objects whose device buffer belongs to this ctx

    # -- lifecycle ------------------------------------------------------------------------------
    def close(self) -> None:
        if self._ctx:
            for m in list(self._live):  # matrices nobody freed (an exception skipped P.free()): before the ctx goes
                m.free()
            self._L.mcg_finalize(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self) -> None:
        check(self._L.mcg_synchronize(self._ctx))

    def trim(self) -> None:
        check(self._L.mcg_trim(self._ctx))

    # -- collectives ----------------------------------------------------------------------------
    def set_allreduce(self, fn: Optional[Callable[[int, int, int], None]]) -> None:
        """fn(device_ptr, count, stream_handle) must sum `count` doubles in place over all ranks."""
        if fn is None:
            self._cb = None
            check(self._L.mcg_set_allreduce(self._ctx, C.cast(None, N.ALLREDUCE_FN), None))
            return

        def _tramp(_user, buf, count, stream):
            try:
                fn(int(buf), int(count), int(stream or 0))
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._cb = N.ALLREDUCE_FN(_tramp)
        check(self._L.mcg_set_allreduce(self._ctx, self._cb, None))

    def use_torch_distributed(self, group=None) -> None:
        """All-reduce through torch.distributed (backend "nccl" is RCCL on ROCm).  The ctx must run
        on torch's current stream (PathEngine(stream=torch.cuda.current_stream().cuda_stream))."""
        import torch
        import torch.distributed as dist

        views = {}  # (ptr, count) -> tensor aliasing the context's workspace (a handful of fixed addresses)

        def _ar(ptr: int, count: int, _stream: int) -> None:
            t = views.get((ptr, count))
            if t is None:
                t = views[(ptr, count)] = torch.as_tensor(_DevView(ptr, count), device=torch.device("cuda", self.device))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

        self.set_allreduce(_ar)

    def init_rccl(self, rank: int, world: int, broadcast_bytes: Callable[[Optional[bytes]], bytes]) -> None:
        """Built-in RCCL communicator.  broadcast_bytes(id_or_None) returns rank 0's 128-byte id on
        every rank (e.g. via torch.distributed.broadcast_object_list).  Rank 0 ALWAYS enters the broadcast: if it cannot
        create the id it sends an empty one, and every rank raises -- no rank is left alone in a collective."""
        uid, err = None, None
        if rank == 0:
            buf = C.create_string_buffer(128)
            if self._L.mcg_comm_unique_id(buf) != 0:
                msg = self._L.mcg_last_error()
                err = msg.decode() if msg else "mcg_comm_unique_id failed"
                uid = b""
            else:
                uid = buf.raw
        uid = broadcast_bytes(uid)
        if not uid:
            raise McgError("rank 0 could not create the RCCL id" + (f": {err}" if err else ""), 7)
        check(self._L.mcg_comm_init_rank(self._ctx, uid, int(world), int(rank)))

    def rccl_probe(self) -> None:
        """Raises McgError unless librccl loads and hands out an id on THIS rank.  A launcher calls it on every rank and
        lets the ranks agree BEFORE init_rccl: a rank that cannot load the library would otherwise leave its peers waiting
        inside ncclCommInitRank."""
        buf = C.create_string_buffer(128)
        check(self._L.mcg_comm_unique_id(buf))

    def init_shm(self, name: str, rank: int, world: int, peer_mailbox: bool = False) -> bool:
        """Node-local shared-memory collective (mcg_comm_init_shm): `name` starts with '/', is the same on every rank
        and unique to the job.  Host all-reduce for the sums; the one-launch LSM sweeps exchange their per-date moments
        between the GPUs inside the kernel.  peer_mailbox: ask for the in-kernel mailbox in the GPUs' own HBM, mapped
        into the peers by HIP IPC (mcg_comm_shm_peer_mailbox; every rank or none); returns whether it is in use."""
        check(self._L.mcg_comm_init_shm(self._ctx, name.encode(), int(world), int(rank)))
        return self.shm_peer_mailbox(True) if peer_mailbox else False

    def shm_peer_mailbox(self, enable: bool = True) -> bool:
        """Collective over the ranks of the segment: switch the in-kernel mailbox between peer-mapped device memory and
        the host segment.  True when the peer-memory mailbox is in use afterwards (all ranks agree)."""
        active = C.c_int()
        check(self._L.mcg_comm_shm_peer_mailbox(self._ctx, int(bool(enable)), C.byref(active)))
        return bool(active.value)

    # -- generation -----------------------------------------------------------------------------
    def gbm(self, seed: int, S0: float, r: float, sigma: float, dt: float, n_steps: int, n_paths: int,
            path_begin: int = 0, payoff: Optional[Tuple[float, bool]] = None) -> PathMatrix:
        h = C.c_void_p()
        if payoff is None:
            check(self._L.mcg_paths_gbm(self._ctx, seed, S0, r, sigma, dt, n_steps, path_begin, n_paths, C.byref(h)))
        else:
            K, is_call = payoff
            check(self._L.mcg_paths_gbm_payoff(self._ctx, seed, S0, r, sigma, dt, n_steps, path_begin, n_paths,
                                               K, int(bool(is_call)), C.byref(h)))
        return PathMatrix(self, h)

    def rbergomi(self, seed: int, S0: float, r: float, xi: float, H: float, eta: float, rho: float, dt: float,
                 n_steps: int, n_paths: int, path_begin: int = 0,
                 payoff: Optional[Tuple[float, bool]] = None) -> PathMatrix:
        h = C.c_void_p()
        if payoff is None:
            check(self._L.mcg_paths_rbergomi(self._ctx, seed, S0, r, xi, H, eta, rho, dt, n_steps, path_begin,
                                             n_paths, C.byref(h)))
        else:
            K, is_call = payoff
            check(self._L.mcg_paths_rbergomi_payoff(self._ctx, seed, S0, r, xi, H, eta, rho, dt, n_steps,
                                                    path_begin, n_paths, K, int(bool(is_call)), C.byref(h)))
        return PathMatrix(self, h)

    def from_host(self, row_major: np.ndarray) -> PathMatrix:
        """Upload [n_paths][n_steps+1] (the reference's pricePaths layout)."""
        a = np.ascontiguousarray(row_major, dtype=np.float64)
        if a.ndim != 2 or a.size == 0:
            raise McgError("LSM::PredictOptionPrice: Empty pricePaths.", 6)
        h = C.c_void_p()
        check(self._L.mcg_paths_from_host(self._ctx, a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0],
                                          a.shape[1], C.byref(h)))
        return PathMatrix(self, h)

    # -- pricing --------------------------------------------------------------------------------
    def price_european(self, paths: PathMatrix, K: float, r: float, T: float, is_call: bool) -> Tuple[float, float]:
        paths._alive()
        m, se = C.c_double(), C.c_double()
        check(self._L.mcg_price_european(self._ctx, paths._h, K, r, T, int(bool(is_call)), C.byref(m), C.byref(se)))
        return m.value, se.value

    def price_lsm(self, paths: PathMatrix, r: float, K: float, maturity: float, dt: float, is_call: bool,
                  poly_order: int) -> Tuple[float, float]:
        paths._alive()
        m, se = C.c_double(), C.c_double()
        check(self._L.mcg_price_lsm(self._ctx, paths._h, r, K, maturity, dt, int(bool(is_call)), int(poly_order),
                                    C.byref(m), C.byref(se)))
        return m.value, se.value

    def lsm_one_launch_enabled(self) -> bool:
        """False once the one-launch LSM sweep's hand-shake has timed out on this ctx (mcg_lsm_one_launch_enabled)."""
        v = C.c_int()
        check(self._L.mcg_lsm_one_launch_enabled(self._ctx, C.byref(v)))
        return bool(v.value)

    def lsm_one_launch_reset(self) -> None:
        """Allow the one-launch sweep again at once after a time-out (mcg_lsm_one_launch_reset)."""
        check(self._L.mcg_lsm_one_launch_reset(self._ctx))

    def debug_lsm_hooks(self, spin_limit: int = -1, poll_delay: int = 0) -> None:
        """Test hooks of the one-launch sweeps' hand-shake (mcg_debug_lsm_hooks)."""
        check(self._L.mcg_debug_lsm_hooks(self._ctx, int(spin_limit), int(poll_delay)))

    def comm_info(self) -> dict:
        """Collective held by this ctx and the ranks it has seen (mcg_comm_info)."""
        k, n, r, seen = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(self._L.mcg_comm_info(self._ctx, C.byref(k), C.byref(n), C.byref(r), C.byref(seen)))
        kinds = {0: "none", 1: "callback", 2: "rccl", 3: "shm", 4: "shm+peer-memory mailbox"}
        return {"kind": kinds.get(k.value, str(k.value)), "n_ranks": n.value, "rank": r.value, "seen_ranks": seen.value}

    def price_asymptotic(self, paths: PathMatrix, r: float, K: float, maturity: float, dt: float, is_call: bool,
                         sigma: float, dividend: float) -> float:
        paths._alive()
        p = C.c_double()
        check(self._L.mcg_price_asymptotic(self._ctx, paths._h, r, K, maturity, dt, int(bool(is_call)), sigma, dividend,
                                           C.byref(p)))
        return p.value

    def price_martingale(self, paths: PathMatrix, r: float, K: float, maturity: float, dt: float, is_call: bool,
                         poly_order: int, max_iterations: int = 5) -> Tuple[float, float, float]:
        """(price, lower, upper) of MartingaleOptimization::PredictOptionPrice."""
        paths._alive()
        p, lo, up = C.c_double(), C.c_double(), C.c_double()
        check(self._L.mcg_price_martingale(self._ctx, paths._h, r, K, maturity, dt, int(bool(is_call)), int(poly_order),
                                           int(max_iterations), C.byref(p), C.byref(lo), C.byref(up)))
        return p.value, lo.value, up.value

    def price_branching(self, paths: PathMatrix, r: float, K: float, maturity: float, dt: float, is_call: bool,
                        num_branches: int, exercise_times, seed: int) -> Tuple[float, float, float]:
        """(price, lower, upper) of BranchingProcesses::PredictOptionPrice; resampling = Philox stream 2 of seed."""
        paths._alive()
        ex = np.ascontiguousarray(exercise_times, dtype=np.int32)
        p, lo, up = C.c_double(), C.c_double(), C.c_double()
        check(self._L.mcg_price_branching(self._ctx, paths._h, r, K, maturity, dt, int(bool(is_call)), int(num_branches),
                                          ex.ctypes.data_as(C.POINTER(C.c_int)), len(ex), int(seed), C.byref(p),
                                          C.byref(lo), C.byref(up)))
        return p.value, lo.value, up.value

    def batch_price_rows(self, rows, n_paths: int = 250, r: float = 0.04, dt: float = 1.0 / 252.0,
                         num_branches: int = 10, poly_order: int = 2, max_iterations: int = 5,
                         seed: int = 0, features=None) -> np.ndarray:
        """mcg_batch_price_rows: rows = sequence of dicts with the mcg_row fields, or the array make_rows() built from
        one (what a caller that prices the same rows repeatedly -- or times the entry point -- holds on to); returns
        [n_rows][4] (asymptotic, branching, lsm, martingale), the driver's four model columns.  features = [n_rows][2]
        (twenty_day_vol, twenty_day_momentum of mcg_row_build): mcg_batch_price_rows6, [n_rows][6] -- the driver's six."""
        arr = rows if isinstance(rows, C.Array) else make_rows(rows)
        n = len(arr)
        dp = C.POINTER(C.c_double)
        if features is None:
            out = np.zeros((n, 4), dtype=np.float64)
            check(self._L.mcg_batch_price_rows(self._ctx, arr, n, int(n_paths), r, dt, int(num_branches),
                                               int(poly_order), int(max_iterations), int(seed), out.ctypes.data_as(dp)))
            return out
        f = np.ascontiguousarray(features, dtype=np.float64)
        if f.shape != (n, 2):
            raise McgError("features must be [n_rows][2]", 1)
        out = np.zeros((n, 6), dtype=np.float64)
        check(self._L.mcg_batch_price_rows6(self._ctx, arr, f.ctypes.data_as(dp), n, int(n_paths), r, dt, int(num_branches),
                                            int(poly_order), int(max_iterations), int(seed), out.ctypes.data_as(dp)))
        return out

    def debug_batch_budget(self, nbytes: int) -> None:
        """Test hook (mcg_debug_batch_budget): workspace bytes of one chunk of the batched rows (0: the default)."""
        check(self._L.mcg_debug_batch_budget(self._ctx, int(nbytes)))

    def debug_lsm_date_fault(self, mode: int = 0, date: int = 0, workgroup: int = 0, delay: int = 0, spin_limit: int = -1) -> None:
        """Test hook of the per-date LSM route's exchange (mcg_debug_lsm_date_fault)."""
        check(self._L.mcg_debug_lsm_date_fault(self._ctx, int(mode), int(date), int(workgroup), int(delay), int(spin_limit)))

    def probe_write_ceiling(self, n_paths: int = 10_000_000, n_steps: int = 252, reps: int = 5) -> Tuple[float, float]:
        """(GB/s, ms per launch) this board writes with the path matrix's store pattern and no arithmetic
        (mcg_probe_write_ceiling)."""
        g, ms = C.c_double(), C.c_double()
        check(self._L.mcg_probe_write_ceiling(self._ctx, int(n_paths), int(n_steps), int(reps), C.byref(g), C.byref(ms)))
        return g.value, ms.value

    def generator_clock_arm(self, on: bool = True) -> None:
        """The next GBM generator launches stamp their shader clock (off by default; mcg_generator_clock_arm)."""
        check(self._L.mcg_generator_clock_arm(self._ctx, 1 if on else 0))

    def generator_clock(self) -> dict:
        """Shader clock of the last ARMED GBM generator launch, stamped in-kernel (mcg_generator_clock)."""
        med, lo, hi, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        check(self._L.mcg_generator_clock(self._ctx, C.byref(med), C.byref(n), C.byref(lo), C.byref(hi)))
        return {"GHz_median": med.value, "GHz_min": lo.value, "GHz_max": hi.value, "stamping_workgroups": n.value}

    def debug_eval(self, fn: int, x: np.ndarray) -> np.ndarray:
        """Test hook (mcg_debug_eval): one device math routine elementwise; returns [n][4]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty((len(x), 4), dtype=np.float64)
        check(self._L.mcg_debug_eval(self._ctx, int(fn), x.ctypes.data_as(C.POINTER(C.c_double)),
                                     y.ctypes.data_as(C.POINTER(C.c_double)), len(x)))
        return y

    # -- measurement ----------------------------------------------------------------------------
    def timing_enable(self, on: bool = True) -> None:
        check(self._L.mcg_timing_enable(self._ctx, int(on)))

    def timing_select(self, kernels=None) -> None:
        """Bracket only these kernels (_native.K_*) while timing is enabled; None: all of them (mcg_timing_select)."""
        mask = 0xFFFFFFFF if kernels is None else sum(1 << int(k) for k in kernels)
        check(self._L.mcg_timing_select(self._ctx, mask))

    def timing_reset(self) -> None:
        check(self._L.mcg_timing_reset(self._ctx))

    def timing_get(self, kernel: int) -> Tuple[float, int]:
        """(total device ms, launches) for one of _native.K_* since the last reset."""
        ms, n = C.c_double(), C.c_int64()
        check(self._L.mcg_timing_get(self._ctx, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def make_rows(rows):
    """ctypes array of mcg_row from a sequence of dicts with its fields (built once, passed to batch_price_rows as is)."""
    arr = (N.Row * len(rows))()
    names = [k for k, _ in N.Row._fields_]
    for i, d in enumerate(rows):
        r = arr[i]
        for k in names:
            setattr(r, k, d[k])
    return arr


def row_features(hist) -> Tuple[float, float]:
    """(twenty_day_vol, twenty_day_momentum) of a spot history (mcg_row_features; PredictionGen.cpp:313-347)."""
    L = N.load_library()
    h = np.ascontiguousarray(hist, dtype=np.float64)
    v, m = C.c_double(), C.c_double()
    check(L.mcg_row_features(h.ctypes.data_as(C.POINTER(C.c_double)), len(h), C.byref(v), C.byref(m)))
    return v.value, m.value


def row_build(hist, underlying_last: float, dte: float, strike_dist_pct: float, option_type: int, dividend: float = 0.08):
    """(row dict with the mcg_row fields, (twenty_day_vol, twenty_day_momentum)) from the driver's own inputs of one
    option row (mcg_row_build; PredictionGen.cpp:664-719)."""
    L = N.load_library()
    h = np.ascontiguousarray(hist, dtype=np.float64)
    row, f = N.Row(), (C.c_double * 2)()
    check(L.mcg_row_build(h.ctypes.data_as(C.POINTER(C.c_double)), len(h), underlying_last, dte, strike_dist_pct, int(option_type),
                          dividend, C.byref(row), f))
    return {k: getattr(row, k) for k, _ in N.Row._fields_}, (f[0], f[1])


def stats(reset: bool = False) -> dict:
    """Process-wide event counters of libmcgpu (mcg_stats): what ran and what fell back."""
    L = N.load_library()
    s = N.Stats()
    check(L.mcg_stats(C.byref(s), int(bool(reset))))
    return {k: int(getattr(s, k)) for k, _ in N.Stats._fields_}


def estimate_params(hist) -> dict:
    """Host-side estimators of the class API (RoughVolatility.cpp:324-331)."""
    L = N.load_library()
    h = np.ascontiguousarray(hist, dtype=np.float64)
    out = np.empty(5)
    check(L.mcg_estimate_params(h.ctypes.data_as(C.POINTER(C.c_double)), len(h), out.ctypes.data_as(C.POINTER(C.c_double))))
    return dict(xi=out[0], H=out[1], eta=out[2], rho=out[3], S0=out[4])


def rbergomi_spectrum(H: float, eta: float, dt: float, n_steps: int):
    """(amp[Mz], comp[n_steps]) -- the LDS-staged spectral amplitudes and compensator."""
    L = N.load_library()
    M = 1
    while M < n_steps:
        M *= 2
    amp, comp, mz = np.empty(M), np.empty(n_steps), C.c_int()
    check(L.mcg_rbergomi_spectrum(H, eta, dt, n_steps, amp.ctypes.data_as(C.POINTER(C.c_double)),
                                  comp.ctypes.data_as(C.POINTER(C.c_double)), C.byref(mz)))
    assert mz.value == M
    return amp, comp
