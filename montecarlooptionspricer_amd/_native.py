"""ctypes binding of libmcgpu.so (the C ABI declared in include/mcgpu.h)."""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("MCG_LIB") or os.path.join(_HERE, "lib", "libmcgpu.so")  # MCG_LIB: A/B experiments with another build

MCG_OK = 0
K_GBM, K_RBERGOMI, K_PAYOFF, K_LSM_SWEEP, K_LSM_SOLVE, K_TRANSPOSE, K_ASYM, K_MARTINGALE, K_BRANCHING, K_BATCH = range(10)
KERNEL_NAMES = {K_GBM: "gbm", K_RBERGOMI: "rbergomi", K_PAYOFF: "payoff", K_LSM_SWEEP: "lsm_sweep",
                K_LSM_SOLVE: "lsm_solve", K_TRANSPOSE: "transpose", K_ASYM: "asymptotic", K_MARTINGALE: "martingale", K_BRANCHING: "branching", K_BATCH: "batch_rows"}

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)


class Row(C.Structure):
    """mcg_row (include/mcgpu.h): one option row of the reference's driver."""
    _fields_ = [("S0", C.c_double), ("xi", C.c_double), ("H", C.c_double), ("eta", C.c_double), ("rho", C.c_double),
                ("strike", C.c_double), ("maturity", C.c_double), ("sigma", C.c_double), ("dividend", C.c_double),
                ("n_steps", C.c_int), ("is_call", C.c_int)]


class Stats(C.Structure):
    """mcg_stats_t (include/mcgpu.h): process-wide event counters."""
    _fields_ = [(k, C.c_int64) for k in (
        "lsm_one_launch_sweeps", "lsm_one_launch_timeouts", "lsm_per_date_sweeps", "lsm_per_date_launches",
        "lsm_per_date_refits", "lsm_per_date_faults", "shm_barrier_failures", "peer_mailbox_enabled", "peer_mailbox_refused",
        "batch_calls", "batch_chunks", "batch_rows", "batch_rows_singly", "batch_peak_workspace_bytes",
        "peer_mailbox_kept", "coalesced_rounds", "coalesced_calls", "coalesced_peak_calls_per_round", "coalesced_fallbacks",
        "coalesced_round_us", "coalesced_device_wait_us", "coalesced_wake_us", "coalesced_prefetched",
        "coalesced_prefetch_hits")]


class McgError(RuntimeError):
    """A non-zero status from libmcgpu (message = mcg_last_error()) or a missing library."""

    def __init__(self, msg: str, status: int = -1):
        super().__init__(msg)
        self.status = status


_lib = None


def lib_path() -> str:
    return _LIB


def load_library():
    """Load libmcgpu.so; fail loudly if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise McgError(f"{_LIB} not found: build it with `make lib` (or __graft_entry__.build()); "
                       "this package has no CPU fallback")
    # One process can host only ONE HIP runtime.  torch wheels bundle their own libamdhip64 (same
    # SONAME as /opt/rocm's): if torch is imported first, libmcgpu binds to that copy and both
    # coexist; the other order gives torch "No HIP GPUs are available".  So when torch is
    # installed, import it before dlopen'ing libmcgpu (MCG_NO_TORCH_PRELOAD=1 skips this).
    if "torch" not in sys.modules and not os.environ.get("MCG_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(_LIB)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.mcg_last_error.restype = C.c_char_p
    L.mcg_version.restype = C.c_char_p
    L.mcg_device_count.argtypes = [C.POINTER(C.c_int)]
    L.mcg_init.argtypes = [C.POINTER(vp), C.c_int]
    L.mcg_init_on_stream.argtypes = [C.POINTER(vp), C.c_int, vp]
    L.mcg_finalize.argtypes = [vp]
    L.mcg_synchronize.argtypes = [vp]
    L.mcg_trim.argtypes = [vp]
    L.mcg_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.mcg_comm_unique_id.argtypes = [C.c_char_p]
    L.mcg_comm_init_rank.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.mcg_comm_init_shm.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]

    def newer(name, argtypes):
        """Entry points added after round 2: an older build loaded through MCG_LIB for an A/B run lacks them."""
        try:
            getattr(L, name).argtypes = argtypes
        except AttributeError:
            if not os.environ.get("MCG_LIB"):
                raise

    newer("mcg_comm_shm_peer_mailbox", [vp, C.c_int, C.POINTER(C.c_int)])
    newer("mcg_debug_shm_attach", [C.c_char_p, C.c_int, C.c_int, C.c_double, C.POINTER(vp)])
    newer("mcg_debug_shm_barrier", [vp])
    newer("mcg_debug_shm_poison", [vp])
    newer("mcg_debug_shm_detach", [vp])
    newer("mcg_lsm_one_launch_reset", [vp])
    newer("mcg_debug_lsm_hooks", [vp, C.c_longlong, C.c_int])
    newer("mcg_comm_info", [vp] + [C.POINTER(C.c_int)] * 4)
    newer("mcg_timing_select", [vp, C.c_uint])
    # round 4
    newer("mcg_stats", [C.POINTER(Stats), C.c_int])
    newer("mcg_debug_lsm_date_fault", [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong])
    newer("mcg_debug_batch_budget", [vp, C.c_size_t])
    newer("mcg_debug_peer_decision", [C.c_int] * 4)
    newer("mcg_probe_write_ceiling", [vp, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)])
    newer("mcg_generator_clock_arm", [vp, C.c_int])
    newer("mcg_generator_clock", [vp, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)])
    newer("mcg_row_features", [C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)])
    newer("mcg_row_build", [C.POINTER(C.c_double), C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double,
                            C.POINTER(Row), C.POINTER(C.c_double)])
    newer("mcg_batch_price_rows6", [vp, C.POINTER(Row), C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int,
                                    C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_double)])
    L.mcg_paths_gbm.argtypes = [vp, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                C.c_uint64, C.c_int64, C.POINTER(vp)]
    L.mcg_paths_gbm_payoff.argtypes = [vp, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                       C.c_uint64, C.c_int64, C.c_double, C.c_int, C.POINTER(vp)]
    L.mcg_paths_rbergomi.argtypes = [vp, C.c_uint64] + [C.c_double] * 7 + [C.c_int, C.c_uint64, C.c_int64,
                                                                          C.POINTER(vp)]
    L.mcg_paths_rbergomi_payoff.argtypes = [vp, C.c_uint64] + [C.c_double] * 7 + [C.c_int, C.c_uint64, C.c_int64,
                                                                                 C.c_double, C.c_int, C.POINTER(vp)]
    L.mcg_paths_from_host.argtypes = [vp, dp, C.c_int64, C.c_int, C.POINTER(vp)]
    L.mcg_paths_to_host.argtypes = [vp, dp]
    L.mcg_paths_to_host_step_major.argtypes = [vp, dp]
    L.mcg_paths_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(vp)]
    L.mcg_paths_free.argtypes = [vp]
    L.mcg_price_european.argtypes = [vp, vp, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp]
    L.mcg_price_lsm.argtypes = [vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, dp, dp]
    L.mcg_lsm_one_launch_enabled.argtypes = [vp, C.POINTER(C.c_int)]
    L.mcg_price_asymptotic.argtypes = [vp, vp] + [C.c_double] * 4 + [C.c_int, C.c_double, C.c_double, dp]
    L.mcg_compat_asymptotic_price.argtypes = [dp, C.c_int64, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_double, C.c_double, dp]
    L.mcg_price_martingale.argtypes = [vp, vp] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_int, dp, dp, dp]
    L.mcg_compat_martingale_price.argtypes = [dp, C.c_int64, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_int, dp]
    ip = C.POINTER(C.c_int)
    L.mcg_price_branching.argtypes = [vp, vp] + [C.c_double] * 4 + [C.c_int, C.c_int, ip, C.c_int, C.c_uint64, dp, dp, dp]
    L.mcg_compat_branching_price.argtypes = [dp, C.c_int64, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, ip, C.c_int, dp]
    L.mcg_batch_price_rows.argtypes = [vp, C.POINTER(Row), C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                       C.c_int, C.c_uint64, dp]
    L.mcg_estimate_params.argtypes = [dp, C.c_size_t, dp]
    L.mcg_rbergomi_spectrum.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, C.POINTER(C.c_int)]
    L.mcg_compat_set_seed.argtypes = [C.c_uint64, C.c_int]
    L.mcg_compat_set_coalescing.argtypes = [C.c_int]
    L.mcg_debug_coalesce_slots.argtypes = [C.c_int]
    L.mcg_debug_coalesce_selftest.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.mcg_compat_generate_paths.argtypes = [dp, C.c_size_t, C.c_int, C.c_int, dp]
    L.mcg_compat_lsm_price.argtypes = [dp, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_int, C.c_int, dp]
    L.mcg_debug_eval.argtypes = [vp, C.c_int, dp, dp, C.c_int64]
    L.mcg_timing_enable.argtypes = [vp, C.c_int]
    L.mcg_timing_reset.argtypes = [vp]
    L.mcg_timing_get.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_int64)]
    _lib = L
    return L


def check(status: int) -> None:
    if status != MCG_OK:
        msg = load_library().mcg_last_error()
        raise McgError(msg.decode() if msg else f"libmcgpu status {status}", status)
