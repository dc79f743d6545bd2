// Device math for the path kernels: Philox block -> normal pair, and the wave/block reductions
// used for payoff and regression moments.  gfx950 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace mcg {

// One Philox block -> four N(0,1) deviates through the device library's log/sincospi/sqrt
// (philox.hpp states the contract).  Reference-grade twin of fm::normal_quad_fast; used by the
// debug hook only.
__device__ __forceinline__ void normal_quad_ref(uint32_t k0, uint32_t k1, uint64_t path, uint32_t block,
                                                uint32_t stream, double (&z)[4]) {
    const Philox4 w = philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), block, stream, k0, k1);
    const uint32_t wa[2] = {w.w0, w.w2}, wb[2] = {w.w1, w.w3};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint64_t a40 = ((uint64_t)(wb[h] & 0xFFu) << 32) | wa[h];
        const double u = ((double)a40 + 0.5) * 0x1p-40;
        const double f = ((double)(wb[h] >> 8) + 0.5) * 0x1p-24;
        const double rad = sqrt(-2.0 * log(u));
        double s, c;
        sincospi(2.0 * f, &s, &c);
        z[2 * h] = rad * c;
        z[2 * h + 1] = rad * s;
    }
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
