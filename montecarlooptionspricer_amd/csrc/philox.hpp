// Counter-based RNG for the path engine: Philox4x32-10 (Salmon et al., SC'11) and the
// block -> two-normals map shared by every kernel.  Device-only code for gfx950.
//
// RNG contract (DESIGN.md "RNG contract"; the oracle's normal_quad mirrors it bit for bit up to
// libm-vs-device rounding of log/sincos):
//   key     = (seed_lo, seed_hi)
//   counter = (path_lo, path_hi, block, stream)       stream 0: price driver, 1: volatility driver
//   words   = philox4x32_10(counter, key) = (w0, w1, w2, w3)
//   one block -> FOUR standard normals, two Box-Muller pairs, 64 bits each:
//     pair A from (w0, w1), pair B from (w2, w3); for a pair (wa, wb):
//       radius uniform  u = ((wb & 0xFF) * 2^32 + wa + 1/2) * 2^-40      40 bits, in (0,1)
//       angle fraction  f = ((wb >> 8) + 1/2) * 2^-24                     24 bits, in (0,1)
//       z_even = sqrt(-2 ln u) cos(2 pi f),  z_odd = sqrt(-2 ln u) sin(2 pi f)
//   element e of block b is draw number 4b + e of its (path, stream): step n of the price driver uses
//   block n >> 2, element n & 3.  (40 radius bits reach 7.5 sigma; 2^24 equally spaced angles leave
//   the marginal law of z exact to far below fp64 resolution -- the angle average is a trapezoid
//   rule on a smooth periodic integrand.)
// A path's draws depend only on (seed, global path id): shards of one job reproduce the
// single-GPU stream exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcg {

enum : uint32_t { STREAM_PRICE = 0u, STREAM_VOL = 1u };

struct Philox4 {
    uint32_t w0, w1, w2, w3;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // 32x32 -> 64 products; hipcc selects v_mad_u64_u32 / v_mul_hi_u32 + v_mul_lo_u32
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input xor in one v_bitop3_b32 (truth table 0x96); the round key is wave-uniform (SGPR)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// Radius uniform of a pair: 40 bits a = (wb & 0xFF):wa -> (a + 1/2) * 2^-40 in (0,1).
// Pasted into the mantissa of a double in [1,2) with the half as the next bit: one exact subtract.
__device__ __forceinline__ double radius_u01(uint32_t wa, uint32_t wb) {
    const uint32_t mhi = 0x3FF00000u | ((wb & 0xFFu) << 12) | (wa >> 20);
    const uint32_t mlo = (wa << 12) | 0x800u;
    return __hiloint2double((int)mhi, (int)mlo) - 1.0;
}

}  // namespace mcg
