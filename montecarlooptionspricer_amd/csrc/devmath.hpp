// Device math for the path kernels: Philox block -> normal pair, and the wave/block reductions
// used for payoff and regression moments.  gfx950 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace mcg {

// One Philox block -> two N(0,1) deviates (Box-Muller; philox.hpp states the contract).
__device__ __forceinline__ void normal_pair(uint32_t k0, uint32_t k1, uint64_t path, uint32_t block,
                                            uint32_t stream, double& z0, double& z1) {
    const Philox4 w = philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), block, stream, k0, k1);
    const double u1 = u01_from_bits(w.w0, w.w1);
    const double u2 = u01_from_bits(w.w2, w.w3);
    const double rad = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    z0 = rad * c;
    z1 = rad * s;
}

// include/core/common.h:8-14
__device__ __forceinline__ double payoff_of(bool is_call, double s, double k) {
    return is_call ? fmax(0.0, s - k) : fmax(0.0, k - s);
}

// Butterfly sum over the 64 lanes of a wave; every lane ends with the total.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum NV values per thread over a block of NW waves.  Result valid in thread 0.
// Deterministic: fixed butterfly inside the wave, fixed wave order across waves.
template <int NV, int NW>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* NV*NW doubles */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = lds[i];
            for (int w = 1; w < NW; ++w) s += lds[w * NV + i];
            v[i] = s;
        }
    }
}

}  // namespace mcg
