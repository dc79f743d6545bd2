"""montecarlooptionspricer_amd -- MI355X-native Monte Carlo path engine.

Host-side mirror (Python) of the reference's pricing-method interface for the hot path of
bcosm/MonteCarloOptionsPricer, over the C ABI in include/mcgpu.h (libmcgpu.so: hand-written HIP for
gfx950).  There is no CPU fallback: importing works anywhere, but every compute entry point raises
McgError when the library or a GPU is missing.

    from montecarlooptionspricer_amd import PathEngine, RoughVolatility, LSM
"""
from ._native import McgError, lib_path, load_library  # noqa: F401
from .engine import PathEngine, PathMatrix, estimate_params, make_rows, rbergomi_spectrum, row_build, row_features, stats  # noqa: F401
from .compat import LSM, AsymptoticAnalysis, BranchingProcesses, MartingaleOptimization, PayoffFunction, RoughVolatility, set_compat_coalescing, set_compat_seed  # noqa: F401
from .sharding import combine_sums, price_from_sums, shard_range  # noqa: F401

__all__ = ["McgError", "PathEngine", "PathMatrix", "RoughVolatility", "LSM", "AsymptoticAnalysis", "MartingaleOptimization", "BranchingProcesses",
           "PayoffFunction", "estimate_params", "rbergomi_spectrum", "make_rows", "row_build", "row_features", "stats",
           "set_compat_seed", "set_compat_coalescing", "shard_range", "combine_sums", "price_from_sums", "load_library", "lib_path"]
