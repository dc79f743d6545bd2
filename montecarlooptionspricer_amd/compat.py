"""Python mirror of the reference's class API for the hot path, same names and argument meaning.

    RoughVolatility().GenerateStockPricePaths(historical_prices, forward_steps, path_num)
        <-> /root/reference/include/models/RoughVolatility.h:15-19
    LSM().PredictOptionPrice(pricePaths, r, strike, maturity, dt, isCall, polyOrder)
        <-> /root/reference/include/models/LSMPricer.h:8-14
    PayoffFunction(isCall, stockPrice, strike)
        <-> /root/reference/include/core/common.h:8-14

Both classes call the C++ drop-in classes inside libmcgpu.so through mcg_compat_*; errors carry the
reference's messages ("Historical prices vector too small.", "LSM::PredictOptionPrice: Empty
pricePaths.") as RuntimeError (McgError), the Python analogue of the reference's std::runtime_error.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from ._native import McgError, check

_dp = C.POINTER(C.c_double)


def PayoffFunction(isCall: bool, stockPrice: float, strike: float) -> float:
    return max(0.0, stockPrice - strike) if isCall else max(0.0, strike - stockPrice)


def set_compat_seed(seed: int | None) -> None:
    """Fix the Philox seed used by the class API (None: std::random_device per call, like the reference)."""
    L = N.load_library()
    check(L.mcg_compat_set_seed(0 if seed is None else int(seed), 0 if seed is None else 1))


def set_compat_coalescing(enabled: bool) -> None:
    """Answer class-API calls of different host threads together (mcg_compat_set_coalescing; on by default)."""
    L = N.load_library()
    check(L.mcg_compat_set_coalescing(1 if enabled else 0))


class RoughVolatility:
    def GenerateStockPricePaths(self, historical_prices, forward_steps: int, path_num: int):
        L = N.load_library()
        h = np.ascontiguousarray(historical_prices, dtype=np.float64).ravel()
        if forward_steps < 0 or path_num < 0:
            raise McgError("RoughVolatility: negative forward_steps or path_num", 1)
        out = np.empty((path_num, forward_steps + 1), dtype=np.float64)
        check(L.mcg_compat_generate_paths(h.ctypes.data_as(_dp), len(h), int(forward_steps), int(path_num),
                                          out.ctypes.data_as(_dp)))
        return out


class LSM:
    def PredictOptionPrice(self, pricePaths, r: float, strike: float, maturity: float, dt: float, isCall: bool,
                           polyOrder: int) -> float:
        L = N.load_library()
        a = np.ascontiguousarray(pricePaths, dtype=np.float64)
        if a.ndim != 2:
            a = a.reshape(0, 0)
        price = C.c_double()
        check(L.mcg_compat_lsm_price(a.ctypes.data_as(_dp), a.shape[0], a.shape[1], r, strike, maturity, dt,
                                     int(bool(isCall)), int(polyOrder), C.byref(price)))
        return price.value


class AsymptoticAnalysis:
    """<-> /root/reference/include/models/AsymptoticAnalysisPricer.h:5-16."""

    def PredictOptionPrice(self, pricePaths, r: float, strike: float, maturity: float, dt: float, isCall: bool,
                           sigma: float, dividend: float) -> float:
        L = N.load_library()
        a = np.ascontiguousarray(pricePaths, dtype=np.float64)
        if a.ndim != 2:
            a = a.reshape(0, 0)
        price = C.c_double()
        check(L.mcg_compat_asymptotic_price(a.ctypes.data_as(_dp), a.shape[0], a.shape[1], r, strike, maturity, dt,
                                            int(bool(isCall)), sigma, dividend, C.byref(price)))
        return price.value


class MartingaleOptimization:
    """<-> /root/reference/include/models/MartingaleOptimizationPricer.h:7-18."""

    def PredictOptionPrice(self, pricePaths, r: float, strike: float, maturity: float, dt: float, isCall: bool,
                           polyOrder: int, maxIterations: int = 5) -> float:
        L = N.load_library()
        a = np.ascontiguousarray(pricePaths, dtype=np.float64)
        if a.ndim != 2:
            a = a.reshape(0, 0)
        price = C.c_double()
        check(L.mcg_compat_martingale_price(a.ctypes.data_as(_dp), a.shape[0], a.shape[1], r, strike, maturity, dt,
                                            int(bool(isCall)), int(polyOrder), int(maxIterations), C.byref(price)))
        return price.value


class BranchingProcesses:
    """<-> /root/reference/include/models/BranchingProcessPricer.h:5-16."""

    def PredictOptionPrice(self, pricePaths, r: float, strike: float, maturity: float, dt: float, isCall: bool,
                           numBranches: int, exerciseTimes) -> float:
        L = N.load_library()
        a = np.ascontiguousarray(pricePaths, dtype=np.float64)
        if a.ndim != 2:
            a = a.reshape(0, 0)
        ex = np.ascontiguousarray(exerciseTimes, dtype=np.int32).ravel()
        price = C.c_double()
        check(L.mcg_compat_branching_price(a.ctypes.data_as(_dp), a.shape[0], a.shape[1], r, strike, maturity, dt,
                                           int(bool(isCall)), int(numBranches), ex.ctypes.data_as(C.POINTER(C.c_int)),
                                           len(ex), C.byref(price)))
        return price.value
