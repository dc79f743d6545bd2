// Device math for the path kernels: Philox block -> normal pair, and the wave/block reductions
// used for payoff and regression moments.  gfx950 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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

// Butterfly sum over the 64 lanes of a wave (strides 32, 16, 8, 4, 2, 1); every lane ends with the total.
// No LDS: strides 32 and 16 swap halves / rows between two copies of the value (v_permlane32_swap, v_permlane16_swap:
// afterwards the two registers of a lane hold both addends of its butterfly), strides 8 .. 1 are DPP moves.  The stride-4
// stage reads lane + 4 (mod 16) instead of lane ^ 4: after the stride-8 stage lanes l and l ^ 8 hold the same value, so
// the addend is the same.  Bit-identical to the __shfl_xor butterfly it replaces (22 vector instructions per value
// instead of six ~100-cycle trips through ds_bpermute: the per-date exchange of the LSM sweeps waits on this).
// Must be called with all 64 lanes active.
__device__ __forceinline__ double wave_sum(double v) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const v2u l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const v2u h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const v2u l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const v2u h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    auto dpp = [](double x, auto ctrl_tag) {
        constexpr int ctrl = decltype(ctrl_tag)::value;
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), ctrl, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), ctrl, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    };
    v += dpp(v, std::integral_constant<int, 0x128>{});  // row_ror:8
    v += dpp(v, std::integral_constant<int, 0x124>{});  // row_ror:4
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
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
