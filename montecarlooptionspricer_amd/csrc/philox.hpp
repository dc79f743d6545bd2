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

// UNIFORM_C23: the caller guarantees that c2, c3 and the key are wave-uniform (every kernel's price stream: block and
// stream number do not depend on the lane).
template <bool UNIFORM_C23 = false>
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // 32x32 -> 64 products; hipcc selects v_mad_u64_u32 / v_mul_hi_u32 + v_mul_lo_u32
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input xor in one v_bitop3_b32 (truth table 0x96); the round key is wave-uniform (SGPR)
        uint32_t n0, n2;
        if (UNIFORM_C23 && r == 0) {
            // c2 (block) and c3 (stream) are wave-uniform, so two of the three xor inputs are scalars: fold them on
            // the scalar unit (a VALU instruction reads one scalar; hipcc would copy the other into a VGPR first)
            n0 = __builtin_amdgcn_readfirstlane((uint32_t)(p1 >> 32) ^ k0) ^ c1;
            n2 = (uint32_t)(p0 >> 32) ^ __builtin_amdgcn_readfirstlane(c3 ^ k1);
        } else if (UNIFORM_C23 && r == 1) {
            n0 = (uint32_t)(p1 >> 32) ^ __builtin_amdgcn_readfirstlane(c1 ^ k0);  // c1 = low word of M1 * block: uniform
            n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        } else {
            n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
            n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        }
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// The same function for a lane that walks through the blocks of ONE (path, stream) with a wave-uniform block number:
// everything in rounds 1 and 2 that depends on the path and the key only is computed once (philox_lane_setup), and
// the inputs that depend on the block only are combined on the scalar unit.  Per block this saves two 32x32
// multiplies and two xors of the 40 + 40 that ten rounds cost.  Bit-identical to philox4x32_10.
struct PhiloxLane {
    uint32_t c1;    // path_hi
    uint32_t lo0;   // low word of M0 * path_lo              (c3 after round 1)
    uint32_t hi1b;  // high word of M1 * c2', c2' = hi(M0 * path_lo) ^ stream ^ k1   (round 2)
    uint32_t lo1b;  // low word of the same product          (c1 after round 2)
};

__device__ __forceinline__ PhiloxLane philox_lane_setup(uint64_t path, uint32_t stream, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint32_t)path;
    const uint32_t c2b = (uint32_t)(p0 >> 32) ^ (stream ^ k1);
    const uint64_t p1b = (uint64_t)0xCD9E8D57u * c2b;
    return PhiloxLane{(uint32_t)(path >> 32), (uint32_t)p0, (uint32_t)(p1b >> 32), (uint32_t)p1b};
}

// block, k0, k1: wave-uniform
__device__ __forceinline__ Philox4 philox4x32_10_lane(const PhiloxLane& L, uint32_t block, uint32_t k0, uint32_t k1) {
    // round 1: only M1 * block is new, and it is scalar
    const uint64_t p1a = (uint64_t)0xCD9E8D57u * block;
    uint32_t c0 = __builtin_amdgcn_readfirstlane((uint32_t)(p1a >> 32) ^ k0) ^ L.c1;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
    // round 2
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        c0 = L.hi1b ^ __builtin_amdgcn_readfirstlane((uint32_t)p1a ^ k0);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), L.lo0, k1, 0x96);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
        uint32_t c1 = L.lo1b, c2 = n2, c3 = (uint32_t)p0;
#pragma unroll
        for (int r = 2; r < 10; ++r) {
            const uint64_t q0 = (uint64_t)0xD2511F53u * c0;
            const uint64_t q1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(q1 >> 32), c1, k0, 0x96);
            const uint32_t m2 = __builtin_amdgcn_bitop3_b32((uint32_t)(q0 >> 32), c3, k1, 0x96);
            c1 = (uint32_t)q1;
            c3 = (uint32_t)q0;
            c0 = n0;
            c2 = m2;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        return Philox4{c0, c1, c2, c3};
    }
}

// ... and for a lane whose block number is its own (the rBergomi generator: the block depends on the lane's place in
// the transform): the same per-path products, no scalar folding.  Saves two 32x32 multiplies and one xor per block.
__device__ __forceinline__ Philox4 philox4x32_10_path(const PhiloxLane& L, uint32_t block, uint32_t k0, uint32_t k1) {
    const uint64_t p1a = (uint64_t)0xCD9E8D57u * block;
    uint32_t c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1a >> 32), L.c1, k0, 0x96);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    c0 = __builtin_amdgcn_bitop3_b32(L.hi1b, (uint32_t)p1a, k0, 0x96);
    uint32_t c2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), L.lo0, k1, 0x96);
    uint32_t c1 = L.lo1b, c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
#pragma unroll
    for (int r = 2; r < 10; ++r) {
        const uint64_t q0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t q1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(q1 >> 32), c1, k0, 0x96);
        const uint32_t m2 = __builtin_amdgcn_bitop3_b32((uint32_t)(q0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)q1;
        c3 = (uint32_t)q0;
        c0 = n0;
        c2 = m2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// Radius uniform of a pair: 40 bits a = (wb & 0xFF):wa -> (a + 1/2) * 2^-40 in (0,1).
// Pasted into the mantissa of a double in [1,2) with the half as the next bit: one exact subtract.
__device__ __forceinline__ double radius_u01(uint32_t wa, uint32_t wb) {
    // (wb:wa >> 20) = (wb << 12) | (wa >> 20); mask to the 20 mantissa bits and set the exponent in ONE v_and_or_b32
    // (hipcc emits v_and + v_or: gfx950 has no VOP3 literals and one constant-bus read per instruction, so one
    // constant sits in a vector register, the other in a scalar one)
    uint32_t mhi;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(mhi) : "v"(__builtin_amdgcn_alignbit(wb, wa, 20)), "v"(0xFFFFFu), "s"(0x3FF00000u));
    const uint32_t mlo = (wa << 12) | 0x800u;
    return __hiloint2double((int)mhi, (int)mlo) - 1.0;
}

}  // namespace mcg
