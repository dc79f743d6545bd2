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

// NV sums at once, FOLDED: the two swap stages exchange halves / rows between TWO registers, so applied to two DIFFERENT
// values they leave one register that holds value a's partial sums in one half (or row) of the wave and value b's in the
// other -- each of the two stages halves the number of live registers instead of costing 7 instructions per value:
//   stride 32: (v[2i], v[2i+1]) -> w[i]:  lanes 0..31 hold v[2i]'s partials, lanes 32..63 v[2i+1]'s      (3 instructions)
//   stride 16: (w[2k], w[2k+1]) -> u[k]:  rows 0..3 of 16 lanes hold v[4k], v[4k+2], v[4k+1], v[4k+3]     (3 instructions)
//   strides 8, 4, 2, 1: the DPP stages of wave_sum on the ceil(NV / 4) registers that are left.
// Eight values: 12 + 6 + 24 = 42 instructions instead of 8 x 22.  Every lane adds exactly the pairs it adds in wave_sum
// (own + partner: commutative), so each total is BIT-IDENTICAL to wave_sum's; what differs is where it ends up: the total
// of v[4k + i] sits in u[k], row {0, 2, 1, 3}[i] (all 16 lanes of the row).  Must be called with all 64 lanes active.
template <int NV>
struct WaveSums {
    static constexpr int N2 = (NV + 1) / 2, N4 = (N2 + 1) / 2;
    double u[N4];
    static __device__ __forceinline__ int home_lane(int i) { return 16 * (((i & 3) >> 1) + 2 * (i & 1)); }
    // the total of v[i] as a wave-uniform value (two v_readlane_b32: it lands in scalar registers)
    __device__ __forceinline__ double get(int i) const {
        const double x = u[i >> 2];
        const int lane = home_lane(i);
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
    }
};

template <int NV>
__device__ __forceinline__ WaveSums<NV> wave_sums_folded(const double (&v)[NV]) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    constexpr int N2 = WaveSums<NV>::N2, N4 = WaveSums<NV>::N4;
    double w[N2];
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const double a = v[2 * i], b = v[2 * i + 1 < NV ? 2 * i + 1 : 2 * i];
        const v2u l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
        const v2u h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
        w[i] = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    WaveSums<NV> r;
#pragma unroll
    for (int k = 0; k < N4; ++k) {
        const double a = w[2 * k], b = w[2 * k + 1 < N2 ? 2 * k + 1 : 2 * k];
        const v2u l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
        const v2u h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
        r.u[k] = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    auto dpp = [](double x, auto ctrl_tag) {
        constexpr int ctrl = decltype(ctrl_tag)::value;
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), ctrl, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), ctrl, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    };
#pragma unroll
    for (int k = 0; k < N4; ++k) {
        double x = r.u[k];
        x += dpp(x, std::integral_constant<int, 0x128>{});  // row_ror:8
        x += dpp(x, std::integral_constant<int, 0x124>{});  // row_ror:4
        x += dpp(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
        x += dpp(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
        r.u[k] = x;
    }
    return r;
}

// v[i] <- its sum over the wave, in every lane (wave-uniform: the totals come back through scalar registers)
template <int NV>
__device__ __forceinline__ void wave_sum_all(double (&v)[NV]) {
    if constexpr (NV < 2) {
        v[0] = wave_sum(v[0]);
    } else {
        const WaveSums<NV> r = wave_sums_folded<NV>(v);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = r.get(i);
    }
}

// Sum NV values per thread over a block of NW waves.  Result valid in thread 0.
// Deterministic: fixed butterfly inside the wave, fixed wave order across waves.
template <int NV, int NW>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* NV*NW doubles */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (NV < 2) {
        v[0] = wave_sum(v[0]);
        if (lane == 0) lds[wave * NV] = v[0];
    } else {
        // folded (wave_sums_folded): value i's total sits in the 16 lanes from WaveSums::home_lane(i) on; that lane stores it
        const WaveSums<NV> r = wave_sums_folded<NV>(v);
        if ((lane & 15) == 0) {
            const int row = lane >> 4;                       // row {0,1,2,3} holds value 4k + {0,2,1,3}
            const int sub = ((row & 1) << 1) | (row >> 1);
#pragma unroll
            for (int k = 0; k < WaveSums<NV>::N4; ++k)
                if (4 * k + sub < NV) lds[wave * NV + 4 * k + sub] = r.u[k];
        }
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
