"""How one pricing job is split over the GPUs of a node (one process per GPU).

Paths are independent (/root/reference/src/models/RoughVolatility.cpp:346-365 has no cross-path
state) and the Philox counter is the GLOBAL path id, so rank g of G simply owns a contiguous id
range; the path matrix is never exchanged.  What is exchanged is tiny: {sum, sum^2, n} of the
payoffs (European) or 3p+2 regression moments per exercise date (LSM), summed over ranks.
These helpers are pure host logic and are exercised with gloo on CPU in tests/.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple


def shard_range(n_paths: int, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """[begin, count) of global path ids owned by `rank`.  Contiguous, balanced to within one unit of
    `align` paths, covers [0, n_paths) exactly once over all ranks; count may be 0 when there are fewer
    units than ranks.  Every begin is a multiple of `align`: the rBergomi generator makes paths in pairs
    (one complex transform serves ids 2q and 2q+1), so its shards use align=2."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    if n_paths < 0:
        raise ValueError("n_paths must be >= 0")
    if align < 1:
        raise ValueError("align must be >= 1")
    units = -(-n_paths // align)  # the last unit may be partial
    base, rem = divmod(units, world)
    u0 = rank * base + min(rank, rem)
    u1 = u0 + base + (1 if rank < rem else 0)
    begin = min(u0 * align, n_paths)
    return begin, min(u1 * align, n_paths) - begin


def combine_sums(parts: Sequence[Sequence[float]]) -> Tuple[float, float, float]:
    """Sum per-shard {sum, sum^2, n} triples (what the all-reduce does)."""
    s = s2 = n = 0.0
    for p in parts:
        s += p[0]
        s2 += p[1]
        n += p[2]
    return s, s2, n


def price_from_sums(sums: Sequence[float], discount: float = 1.0) -> Tuple[float, float]:
    """(mean, std-err) of discount * X from {sum X, sum X^2, n}."""
    s, s2, n = sums
    if not n >= 1:
        raise ValueError("no paths")
    m = s / n
    var = max(0.0, (s2 - n * m * m) / (n - 1)) if n > 1 else 0.0
    return discount * m, discount * math.sqrt(var / n)
