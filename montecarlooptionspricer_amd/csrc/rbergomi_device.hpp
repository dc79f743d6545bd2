// Device-side body of the rBergomi generator, shared by the single-contract kernels
// (kernels_rbergomi.hip) and the batched driver-row kernel (kernels_batch.hip).
//
// Reference behaviour per path (/root/reference/src/models/RoughVolatility.cpp:346-365):
//   X = sqrt(2H) eta Re(FFT^-(phi (.) Z)/Mz)        (:347-348, :264-292)   a stationary circular Gaussian sequence
//   v_n = xi exp(X_n - 0.5 eta^2 t_n^{2H})          (:349, :294-309)
//   S_{n+1} = S_n exp((r - v_n/2) dt + sqrt(max(0,v_n)) dW_n),  dW_n ~ N(0, dt)   (:354-364)
//
// Device algorithm: spectral synthesis, one complex transform per PAIR of paths.  With the symmetric
// amplitudes a_k of host/volterra.cpp and Y_k = a_k (g_k + i h_k),
//   x_n = sum_{k<Mz} Y_k e^{+2 pi i k n/Mz},   X(path 2q) = Re x,  X(path 2q+1) = Im x
// (two independent copies of the reference's X).  The transform is an in-register radix-2
// decimation-in-time FFT of length Mz = 16 G spread over G lanes (16 complex points per lane; 32 only at
// Mz = 2048): the spectrum is drawn directly in bit-reversed order (each lane's 16 inputs are four runs of 4
// consecutive k, i.e. 8 Philox blocks with no draw wasted), the 2 lowest and 2 (3) highest index bits are
// butterflies between registers, the log2(G) middle bits are butterflies between lanes (DPP moves and
// v_permlane16/32_swap, no LDS), twiddles come from an LDS table.  O(Mz log Mz) instead of the O(Mz steps) contraction of the Volterra form (which ran at the
// fp64-FMA roofline on MFMA in an earlier version of this file: 64 cycles per v_mfma_f64_16x16x4_f64 and no
// overlap with fp64 VALU work on gfx950, tools/ubench_mfma_f64.hip).
// The natural-order output leaves lane g of a pair with steps n = 4(tG + g) + v (t = 0..3, v = 0..3): four
// CONSECUTIVE steps per 4G-step tile.  The price then advances as a product: step factors from a short
// polynomial, in-lane running products, an exclusive product scan over the G lanes, and the tile's rows are
// staged through LDS so that they leave as (pairs x 16)-byte runs of the step-major row.
#pragma once
#include <type_traits>
#include "devmath.hpp"
#include "fastmath.hpp"

namespace mcg {

struct RbArgs {
    double* out;
    int64_t ld;
    int64_t n_paths;
    int n_steps;
    int M;  // Mz = nextpow2(n_steps)
    uint64_t path_begin;  // even
    uint32_t k0, k1;
    double S0, logS0, r, xi, dt, sqdt;
    const double* amp;   // [M] spectral amplitudes a_k
    const double* comp;  // [n_steps]
    const double2* log_tab;
    double K;
    int is_call;
    double* partials;
    unsigned long long* ticket;  // shares handed out beyond the first gridDim.x (zero at launch)
    int64_t n_blocks;  // workgroup shares of the launch (rb_pairs_per_block pairs each); a workgroup takes every gridDim.x-th
};

// Transform points per lane = 4 * 2^LT.  16 points (64 data VGPRs) keep the kernel at 2 waves/SIMD with no
// spills; 32 points only where 64 lanes x 16 points do not cover the transform (Mz = 2048).
#ifndef RB_LT_DEFAULT
#define RB_LT_DEFAULT 2
#endif
__host__ __device__ inline int rb_log_tiles(int M) { return M > (64 * 4 << RB_LT_DEFAULT) ? 3 : RB_LT_DEFAULT; }
// pairs handled by one 256-thread workgroup of the variant chosen for Mz
__host__ __device__ inline int rb_pairs_per_block(int M) { return M < 32 ? 256 : 4 * (64 / (M >> (2 + rb_log_tiles(M)))); }

// Output staging (FFT variants): one 4G-step tile of the workgroup's pairs, [4G rows][pairs + 1 pad] double2,
// double-buffered while LDS allows (LT = 2), so that the step-major rows leave as (pairs * 16)-byte runs
// instead of one 16-byte piece per lane.
#ifndef RB_NBUF
#define RB_NBUF 2
#endif
// the normals of a Philox block request their four table entries together, then compute (fm::normal_quad_fast<true>;
// false = one lookup at a time, 12 registers fewer)
#ifndef RB_EAGER_SPECTRUM
#define RB_EAGER_SPECTRUM true
#endif
#ifndef RB_EAGER_PRICE
#define RB_EAGER_PRICE true
#endif
// (one buffer from Mz = 512 on: with the 34 KiB of generator tables next to it, two would leave room for one workgroup
// per CU only -- since round 3, when the sin/cos table went from 512 to 1024 entries; A/B on one box: C4 9.48 -> 9.31 ms
// with the larger tables and one buffer, 9.33 with the exponential's larger table alone and two buffers)
#define RB_NBUF_MAX_M 512
__host__ __device__ inline int rb_stage_bufs(int M) { return rb_log_tiles(M) == 2 && M < RB_NBUF_MAX_M ? RB_NBUF : 1; }
__host__ __device__ inline size_t rb_stage_units(int M) {  // double2 units per buffer
    return M < 32 ? 0 : (size_t)(M >> rb_log_tiles(M)) * (size_t)(rb_pairs_per_block(M) + 1);
}

// LDS carve-up shared by all variants: amp[M] | comp[max(M, n_steps)], zero beyond n_steps | twiddle (cos, sin)(2 pi q / M), q < max(M/2, 1) | staging
__host__ __device__ inline size_t rb_smem_bytes(int M, int n_steps) {
    const size_t nc = (size_t)(n_steps > M ? n_steps : M);
    const size_t head = (size_t)M + nc + ((M + nc) & 1);
    return (head + 2 * (size_t)(M / 2 + 1) + 2 * rb_stage_bufs(M) * rb_stage_units(M)) * sizeof(double);
}

struct RbLds {
    const double* amp;
    const double* comp;
    const double2* tw;
    double2* stage;
};

// comp_scale: the FFT variants keep the compensator as 128 log2(e) comp -- their variance factor is
// e^{(X + comp)/2} = 2^{(128 log2(e) X + table)/256} (fm::exp2_pair) --, the direct variant as it is.
constexpr double RB_HALF_LOG2E = 0x1.71547652b82fep+7;  // 256 log2(e) / 2: fm::exp2_pair takes its argument in units of (ln 2)/256
__device__ __forceinline__ RbLds rb_stage_lds(const RbArgs& a, double* smem, fm::Tables* tabs, double comp_scale) {
    const int M = a.M;
    double* amp = smem;
    double* comp = smem + M;
    const int nc = a.n_steps > M ? a.n_steps : M;  // (n_steps <= M except in the direct variant, M < 32)
    double2* tw = reinterpret_cast<double2*>(smem + M + nc + ((M + nc) & 1));  // 16-B aligned
    for (int i = threadIdx.x; i < M; i += blockDim.x) amp[i] = a.amp[i];
    // zero beyond the grid: the FFT variants step through whole tiles, and a step that does not exist must stay finite
    for (int i = threadIdx.x; i < nc; i += blockDim.x) comp[i] = i < a.n_steps ? comp_scale * a.comp[i] : 0.0;
    for (int q = threadIdx.x; q < M / 2; q += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)q / (double)M, &s, &c);
        tw[q] = make_double2(c, s);
    }
    fm::load_tables(tabs, a.log_tab);
    __syncthreads();
    return RbLds{amp, comp, tw, tw + (M / 2 + 1)};
}

// {S_A, S_B} -> out[col_a], out[col_a + 1] (col_a even, rows 16-byte aligned)
__device__ __forceinline__ void rb_store_pair(double* row_a, double sa, double sb, bool live_a, bool live_b) {
    typedef double v2d __attribute__((ext_vector_type(2)));
#ifdef RB_NO_STORE
    if (sa != 12345.678) return;
#endif
    if (live_b) {
        v2d v = {sa, sb};
        __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(row_a));
    } else if (live_a) {
        __builtin_nontemporal_store(sa, row_a);
    }
}

// x from lane (l ^ DELTA).  Strides 1, 2 (inside a quad) and 8 (half a 16-lane row) are DPP moves on the vector ALU --
// a few cycles instead of the ~100-cycle round trip of ds_bpermute through LDS that every other stride takes.
#ifndef RB_SWAP4
#define RB_SWAP4 4  // 4: stride-4 stage as complete butterflies through DPP swaps; 0: as a DPP exchange
#endif
template <int DELTA>
__device__ __forceinline__ double lane_xor(double x) {
    if constexpr (DELTA == 1 || DELTA == 2 || DELTA == 8) {
        constexpr int ctrl = DELTA == 1 ? 0xB1      // quad_perm [1,0,3,2]
                             : DELTA == 2 ? 0x4E    // quad_perm [2,3,0,1]
                                          : 0x128;  // row_ror:8
        // (mov_dpp: no `old` operand -- every lane has a valid source here, and update_dpp(0, ...) costs a v_mov of
        // the zero into the destination before every DPP move)
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), ctrl, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), ctrl, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
    } else if constexpr (DELTA == 4) {
        // banks 1 and 3 (lanes 4-7, 12-15 of a row) read lane - 4, then banks 0 and 2 read lane + 4 into the same register
        int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x114, 0xF, 0xA, true);
        int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x114, 0xF, 0xA, true);
        lo = __builtin_amdgcn_update_dpp(lo, __double2loint(x), 0x104, 0xF, 0x5, false);
        hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(x), 0x104, 0xF, 0x5, false);
        return __hiloint2double(hi, lo);
    } else {
        return __shfl_xor(x, DELTA, 64);
    }
}

// Strides 16 and 32: gfx950's v_permlane16_swap / v_permlane32_swap exchange halves BETWEEN TWO registers (odd 16-lane
// rows of a with even rows of b; upper 32 lanes of a with lower 32 lanes of b).  Applied to two values a, b of one
// lane, the lane whose stride bit is 0 ends up with both ends of a's butterfly and its partner with both ends of
// b's: every lane then computes one complete butterfly (one complex multiply instead of two half ones) and a
// second swap sends the results home.  No LDS, no address registers, a few cycles of latency.
// Stride 4 has no swap instruction; two bank-masked DPP moves per 32-bit word do the same (row_shr:4 into banks 1 and
// 3, row_shl:4 into banks 0 and 2).
template <int DELTA>
__device__ __forceinline__ void lane_swap(double& a, double& b) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    static_assert(DELTA == 4 || DELTA == 16 || DELTA == 32, "swap strides: 4 (DPP), 16 and 32 (permlane swaps)");
    if constexpr (DELTA == 4) {
        const int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
        // a' : lanes with stride bit 1 (banks 1, 3) take b from lane - 4; b' : lanes with stride bit 0 take a from lane + 4
        const int nalo = __builtin_amdgcn_update_dpp(alo, blo, 0x114, 0xF, 0xA, false);
        const int nahi = __builtin_amdgcn_update_dpp(ahi, bhi, 0x114, 0xF, 0xA, false);
        const int nblo = __builtin_amdgcn_update_dpp(blo, alo, 0x104, 0xF, 0x5, false);
        const int nbhi = __builtin_amdgcn_update_dpp(bhi, ahi, 0x104, 0xF, 0x5, false);
        a = __hiloint2double(nahi, nalo);
        b = __hiloint2double(nbhi, nblo);
        return;
    }
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    v2u lo, hi;
    if constexpr (DELTA == 4) {
    } else if constexpr (DELTA == 16) {
        lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    } else {
        lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    }
    a = __hiloint2double((int)hi.x, (int)lo.x);
    b = __hiloint2double((int)hi.y, (int)lo.y);
}

// (a, b) -> (a + w b, a - w b) in six FMAs: lo = a + w b as two chained FMAs per component, hi = 2a - lo.
// (The textbook form -- w b, then a +- it -- is eight instructions.)  Outputs may alias the inputs.
__device__ __forceinline__ void rb_butterfly(const double2 w, double ar, double ai, double br, double bi, double& lo_r,
                                             double& lo_i, double& hi_r, double& hi_i) {
    const double lr = fma(w.x, br, fma(-w.y, bi, ar));
    const double li = fma(w.x, bi, fma(w.y, br, ai));
    hi_r = fma(2.0, ar, -lr);
    hi_i = fma(2.0, ai, -li);
    lo_r = lr;
    lo_i = li;
}

// bit reversal of the low LT bits of t (LT = 2 or 3)
template <int LT>
__device__ __forceinline__ constexpr int rb_rev(int t) {
    return LT == 3 ? (((t & 1) << 2) | (t & 2) | ((t >> 2) & 1)) : (((t & 1) << 1) | ((t >> 1) & 1));
}

// One workgroup's share: 4 waves x (64 >> LG) path pairs, 4 * 2^LT transform points per lane
// (Mz = 2^(2 + LG + LT)).  Returns through end_a/end_b the final prices S_T of this lane's pair (the stored values);
// lead = this lane is the one lane (g == 0) that reports the pair's payoff.
// The LDS tables (L, tabs) are staged by the caller: once per workgroup, however many shares it works through.
// tid = threadIdx.x (a parameter so that a looping caller can keep the lane-derived indices out of its loop-invariant set).
template <int LG, int LT>
__device__ __forceinline__ void rb_fft_block(const RbArgs& a, const RbLds& L, int64_t block_index, int tid,
                                             fm::Tables* tabs, double& end_a, double& end_b, bool& live_a, bool& live_b, bool& lead) {
    constexpr int G = 1 << LG;   // lanes per pair
    constexpr int P = 64 >> LG;  // pairs per wave
    constexpr int NT = 1 << LT;  // 4-step tiles per lane
    constexpr int PW = 4 * P;    // pairs per workgroup
    constexpr int RS = PW + 1;   // staging row stride in 16-byte units (one unit of padding)
    constexpr int NBUF = (LT == 2 && (16 << LG) < RB_NBUF_MAX_M) ? RB_NBUF : 1;  // = rb_stage_bufs(Mz), Mz = 16 G
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> (6 - LG), c = lane & (P - 1);
    const int64_t q = block_index * (4 * P) + wave * P + c;  // pair index within the launch
    const int64_t col_a = 2 * q;
    live_a = col_a < a.n_paths;
    live_b = col_a + 1 < a.n_paths;
    const uint64_t id_a = a.path_begin + (uint64_t)col_a, id_b = id_a + 1;
    const uint64_t pair_id = id_a >> 1;
    // the parts of Philox rounds 1-2 that depend on the path and the key only (the block numbers below depend on the lane)
    const PhiloxLane rng_vol = philox_lane_setup(pair_id, STREAM_VOL, a.k1);
    const PhiloxLane rng_a = philox_lane_setup(id_a, STREAM_PRICE, a.k1), rng_b = philox_lane_setup(id_b, STREAM_PRICE, a.k1);

    // ---- spectrum, in bit-reversed order: slot (t, v) of lane g holds Y_k, k = rev(4(tG + g) + v) ----
    double xr[4 * NT], xi[4 * NT];
    const int g_rev = LG ? (int)(__brev((unsigned)g) >> (32 - LG)) : 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int v_rev = ((v & 1) << 1) | (v >> 1);
        const int k_base = (v_rev << (LT + LG)) | (g_rev << LT);  // NT consecutive k: k_base + rev(t)
#pragma unroll
        for (int bq = 0; bq < NT / 2; ++bq) {
            double z[4];
            const double a0 = L.amp[k_base + 2 * bq], a1 = L.amp[k_base + 2 * bq + 1];  // folded into the pairs' radii
            fm::normal_quad_fast<RB_EAGER_SPECTRUM>(a.k0, a.k1, rng_vol, (uint32_t)((k_base >> 1) + bq), tabs, z, a0, a1);
            const int t0 = rb_rev<LT>(2 * bq), t1 = rb_rev<LT>(2 * bq + 1);
            xr[t0 * 4 + v] = z[0];
            xi[t0 * 4 + v] = z[1];
            xr[t1 * 4 + v] = z[2];
            xi[t1 * 4 + v] = z[3];
        }
    }

    // ---- stages 1, 2: index bits 0, 1 (v), twiddles 1 and i ----
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int v = 0; v < 4; v += 2) {
            const double ar = xr[t * 4 + v], ai = xi[t * 4 + v], br = xr[t * 4 + v + 1], bi = xi[t * 4 + v + 1];
            xr[t * 4 + v] = ar + br;
            xi[t * 4 + v] = ai + bi;
            xr[t * 4 + v + 1] = ar - br;
            xi[t * 4 + v + 1] = ai - bi;
        }
        {  // (0,2) with w = 1
            const double ar = xr[t * 4], ai = xi[t * 4], br = xr[t * 4 + 2], bi = xi[t * 4 + 2];
            xr[t * 4] = ar + br;
            xi[t * 4] = ai + bi;
            xr[t * 4 + 2] = ar - br;
            xi[t * 4 + 2] = ai - bi;
        }
        {  // (1,3) with w = e^{2 pi i /4} = i:  w*b = (-bi, br)
            const double ar = xr[t * 4 + 1], ai = xi[t * 4 + 1], br = xr[t * 4 + 3], bi = xi[t * 4 + 3];
            xr[t * 4 + 1] = ar - bi;
            xi[t * 4 + 1] = ai + br;
            xr[t * 4 + 3] = ar + bi;
            xi[t * 4 + 3] = ai - br;
        }
    }

    // ---- stages 3 .. 2+LG: index bits 2 .. 1+LG (the lane bits): butterflies between lanes ----
    // A butterfly (lo, up) -> (lo + w up, lo - w up) has its two ends in lanes g and g ^ 2^b.  Each lane first
    // multiplies its own value by w_eff (w in the upper lane, 1 in the lower), the lanes swap, and the result is
    // partner + sgn * own (sgn = +1 lower, -1 upper): no per-element selects.
    auto lane_stage = [&](auto b_tag) {
        constexpr int b = decltype(b_tag)::value;
        constexpr int DELTA = P << b;
        const int j_hi = (g & ((1 << b) - 1)) << 2;  // twiddle exponent j = j_hi | v, stage s = 3 + b
        if constexpr (DELTA == RB_SWAP4 || DELTA == 16 || DELTA == 32) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const double2 w = L.tw[(j_hi | v) << (LG + LT - 1 - b)];
#pragma unroll
                for (int tp = 0; tp < NT / 2; ++tp) {
                    const int i0 = (2 * tp) * 4 + v, i1 = (2 * tp + 1) * 4 + v;
                    lane_swap<DELTA>(xr[i0], xr[i1]);  // now [i0] = lower end, [i1] = upper end of ONE butterfly
                    lane_swap<DELTA>(xi[i0], xi[i1]);
                    double lo_r, lo_i, hi_r, hi_i;
                    rb_butterfly(w, xr[i0], xi[i0], xr[i1], xi[i1], lo_r, lo_i, hi_r, hi_i);
                    lane_swap<DELTA>(lo_r, hi_r);
                    lane_swap<DELTA>(lo_i, hi_i);
                    xr[i0] = lo_r;
                    xr[i1] = hi_r;
                    xi[i0] = lo_i;
                    xi[i1] = hi_i;
                }
            }
            return;
        }
        const bool upper = ((g >> b) & 1) != 0;
        const double sgn = upper ? -1.0 : 1.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double2 w = L.tw[(j_hi | v) << (LG + LT - 1 - b)];
            const double wx = upper ? w.x : 1.0, wy = upper ? w.y : 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double mr = xr[t * 4 + v], mi = xi[t * 4 + v];
                const double tr = wx * mr - wy * mi, ti = wx * mi + wy * mr;
                const double pr = lane_xor<DELTA>(tr), pi = lane_xor<DELTA>(ti);
                xr[t * 4 + v] = fma(sgn, tr, pr);
                xi[t * 4 + v] = fma(sgn, ti, pi);
            }
        }
    };
    if constexpr (LG > 0) lane_stage(std::integral_constant<int, 0>{});
    if constexpr (LG > 1) lane_stage(std::integral_constant<int, 1>{});
    if constexpr (LG > 2) lane_stage(std::integral_constant<int, 2>{});
    if constexpr (LG > 3) lane_stage(std::integral_constant<int, 3>{});
    if constexpr (LG > 4) lane_stage(std::integral_constant<int, 4>{});
    if constexpr (LG > 5) lane_stage(std::integral_constant<int, 5>{});

    // ---- stages 3+LG .. 2+LG+LT: index bits 2+LG .. (t): butterflies between registers ----
#pragma unroll
    for (int bt = 0; bt < LT; ++bt) {
#pragma unroll
        for (int tl = 0; tl < (1 << bt); ++tl) {  // t bits below bt
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int j = (((tl << LG) | g) << 2) | v;  // i0 mod 2^(s-1), s = 3 + LG + bt
                const double2 w = L.tw[j << (LT - 1 - bt)];
#pragma unroll
                for (int th = 0; th < (NT / 2 >> bt); ++th) {  // t bits above bt
                    const int t_lo = (th << (bt + 1)) | tl, t_up = t_lo | (1 << bt);
                    rb_butterfly(w, xr[t_lo * 4 + v], xi[t_lo * 4 + v], xr[t_up * 4 + v], xi[t_up * 4 + v], xr[t_lo * 4 + v],
                                 xi[t_lo * 4 + v], xr[t_up * 4 + v], xi[t_up * 4 + v]);
                }
            }
        }
    }
    // now xr/xi[t*4+v] = Re/Im x_n, n = 4(tG + g) + v

    // ---- price stepping, two paths per lane ----
    // S advances as a product.  A lane holds four consecutive steps of its two paths: the step factors e^inc come from
    // a short polynomial without range reduction (|inc| is tested for the whole wave: a price step's exponent is a few
    // per cent), their running products p_v are formed in the lane, an exclusive product scan over the G lanes of the
    // pair gives the lane's lead, and S = (S_tile * lead) * p_v.  The next tile starts from the STORED value of this
    // tile's last step (broadcast from the last lane), so the sequence of stored prices is exactly a chain of
    // multiplications like the reference's S_j = S_{j-1} exp(.) (:363), only associated differently within a tile.
    lead = g == 0;
    double* col = a.out + col_a;
    if (lead) rb_store_pair(col, a.S0, a.S0, true, true);
    double S_a = a.S0, S_b = a.S0;  // price at the start of the tile (the same bits in every lane of the pair)
    // this thread's part of every tile's write-out: rows wr_row + i G (i < 4) of the tile, pair wr_pc of the workgroup
    const int wr_row = tid / PW, wr_pc = tid % PW;
    double* wr_out = a.out + 2 * (block_index * PW + wr_pc) + (int64_t)(wr_row + 1) * a.ld;
    const double sq_xi_dt = sqrt(a.xi) * a.sqdt;  // sqrt(xi dt)
    const double neg_half_xi_dt = -0.5 * a.xi * a.dt, r_dt = a.r * a.dt;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (((t * G) << 2) < a.n_steps) {  // wave-uniform: the tile has at least one live step
            const int nl = ((t * G + g) << 2);
            double2* stage = L.stage + (NBUF == 2 ? (t & 1) * (4 * G * RS) : 0);
            // Everything below works in place on the tile's eight transform values (xr = path A, xi = path B), so that the
            // tile adds no long-lived registers to the 64 of the transform: first e^{(X + comp)/2} (sqrt(v) = sqrt(xi)
            // times it: one exponential gives both v and sqrt(v dt)), then, one path at a time, the four normals of
            // Philox block nl/4 of the path's price stream turn them into the exponents of the four price steps.
            // Steps beyond the grid (last tile only) are computed like any other -- finite, never stored, never used.
            double* const ia = xr + t * 4;
            double* const ib = xi + t * 4;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const double cmp = L.comp[nl + v];  // 128 log2(e) comp_n
                fm::exp2_pair(fma(ia[v], RB_HALF_LOG2E, cmp), fma(ib[v], RB_HALF_LOG2E, cmp), tabs, ia[v], ib[v]);
                __builtin_amdgcn_sched_barrier(0);  // one step's pair of chains at a time: more of them cost registers, not time
            }
            {
                double z[4];
                // (sqrt(xi dt) folded into the pairs' radii: z = sqrt(xi dt) N(0,1))
                fm::normal_quad_fast<RB_EAGER_PRICE>(a.k0, a.k1, rng_a, (uint32_t)(nl >> 2), tabs, z, sq_xi_dt, sq_xi_dt);
#pragma unroll
                for (int v = 0; v < 4; ++v)  // (r - v/2) dt + sqrt(v dt) N,  v = xi e^2
                    ia[v] = fma(ia[v], z[v], fma(neg_half_xi_dt, ia[v] * ia[v], r_dt));
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                double z[4];
                fm::normal_quad_fast<RB_EAGER_PRICE>(a.k0, a.k1, rng_b, (uint32_t)(nl >> 2), tabs, z, sq_xi_dt, sq_xi_dt);
#pragma unroll
                for (int v = 0; v < 4; ++v) ib[v] = fma(ib[v], z[v], fma(neg_half_xi_dt, ib[v] * ib[v], r_dt));
            }
            double big = 0.0;
#pragma unroll
            for (int v = 0; v < 4; ++v) big = fmax(big, fmax(fabs(ia[v]), fabs(ib[v])));
            // e^inc - 1 by the shortest polynomial the wave's largest exponent allows, then the running products
            // p_v = prod_{u <= v} e^{inc_u} (in place: ia/ib end up holding them)
            if (__builtin_amdgcn_ballot_w64(big > fm::SMALL6_EXP_BOUND) == 0ull) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    fm::expm1_small6_2(ia[v], ib[v], ia[v], ib[v]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (__builtin_amdgcn_ballot_w64(big > 0.34) == 0ull) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    fm::expm1_small9_2(ia[v], ib[v], ia[v], ib[v]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double ea, eb;
                    fm::exp_full2(ia[v], ib[v], ea, eb);
                    ia[v] = ea - 1.0;  // (absolute error 1e-16: what the product below needs)
                    ib[v] = eb - 1.0;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            double* const pa = ia;
            double* const pb = ib;
            pa[0] = 1.0 + ia[0];
            pb[0] = 1.0 + ib[0];
#pragma unroll
            for (int v = 1; v < 4; ++v) {
                pa[v] = fma(pa[v - 1], ia[v], pa[v - 1]);
                pb[v] = fma(pb[v - 1], ib[v], pb[v - 1]);
            }
            // exclusive product scan of the lane totals over g = 0..G-1 (lanes c, c+P, c+2P, ...)
            // (Round 6, profiles/r06_rb_limiter.json: by ablation these 1 + LG dependent lane exchanges are 6-7 % of the kernel.
            // Hoisting the scans of all NT tiles of a share into one phase -- 2 NT chains in flight together -- was tried and
            // LOST 1.5-3.3 %, bit-identical output: profiles/r06_rb_batched_scan_attempt.diff.)
#ifdef RB_ABL_NO_SCAN   // (timing study only: tools/build_variant.sh abl_scan kernels_rbergomi.hip -DRB_ABL_NO_SCAN)
            double la = 1.0, lb = 1.0;
#else
            double la = __shfl_up(pa[3], P, 64), lb = __shfl_up(pb[3], P, 64);
            if (g == 0) {
                la = 1.0;
                lb = 1.0;
            }
#pragma unroll
            for (int b = 0; b < LG; ++b) {
                // lanes g <= 2^b read lane 0 of their pair, whose lead is 1 and stays 1: no select on the product
                const int src = max(lane - (P << b), c);
                la *= __shfl(la, src, 64);
                lb *= __shfl(lb, src, 64);
            }
#endif
            const double base_a = S_a * la, base_b = S_b * lb;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                pa[v] *= base_a;
                pb[v] *= base_b;
                stage[(4 * g + v) * RS + wave * P + c] = make_double2(pa[v], pb[v]);
            }
            // the tile's last step of the grid: lane G - 1, v = 3, except in a last tile that the grid leaves unfinished
            double last_a = pa[3], last_b = pb[3];
            int src_g = G - 1;
            if ((((t + 1) * G) << 2) > a.n_steps) {  // wave-uniform
                const int n_last = a.n_steps - 1 - ((t * G) << 2);
                src_g = n_last >> 2;
                switch (n_last & 3) {
                    case 0: last_a = pa[0]; last_b = pb[0]; break;
                    case 1: last_a = pa[1]; last_b = pb[1]; break;
                    case 2: last_a = pa[2]; last_b = pb[2]; break;
                    default: break;
                }
            }
#ifdef RB_ABL_NO_SCAN
            S_a = last_a + (double)src_g * 1e-300;
            S_b = last_b;
#else
            S_a = __shfl(last_a, src_g * P + c, 64);
            S_b = __shfl(last_b, src_g * P + c, 64);
#endif
#ifndef RB_ABL_NO_BARRIER   // (timing study only: the tile barrier's share of the waiting -- none, r06_rb_limiter.json)
            __syncthreads();
#endif
            // write-out: the tile is 4G rows x PW pairs = 1024 16-byte units, four per thread (rows wr_row + i G); a
            // wavefront store covers 64 / PW complete rows of PW * 16 contiguous bytes.  Columns are never masked: rows
            // are padded to 256 columns and a workgroup's 2 PW columns divide that, so the columns behind n_paths
            // exist (scratch, never read back) -- like the GBM generator's unconditional stores.
            {
                const double2* sbuf = static_cast<const double2*>(__builtin_assume_aligned(stage, 16));
                double* tile_out = wr_out + (int64_t)t * (4 * G) * a.ld;
                double2 sv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // all four LDS reads in flight before the first store waits for one
                    sv[i] = sbuf[(wr_row + i * G) * RS + wr_pc];
                    asm volatile("" : "+v"(sv[i].x), "+v"(sv[i].y));
                }
                if ((((t + 1) * G) << 2) <= a.n_steps) {  // wave-uniform: every row of the tile is a step of the grid
#pragma unroll
                    for (int i = 0; i < 4; ++i) rb_store_pair(tile_out + (int64_t)(i * G) * a.ld, sv[i].x, sv[i].y, true, true);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (((t * G) << 2) + wr_row + i * G < a.n_steps)
                            rb_store_pair(tile_out + (int64_t)(i * G) * a.ld, sv[i].x, sv[i].y, true, true);
                }
            }
            if (NBUF == 1) __syncthreads();
        }
    }
    end_a = S_a;
    end_b = S_b;
    // an odd number of live tiles leaves the last tile's reads and the next share's first writes on the same buffer
    if (NBUF == 2 && (((a.n_steps + 4 * G - 1) / (4 * G)) & 1)) __syncthreads();
}

template <int LG, int LT>
__device__ __forceinline__ void rb_generate_fft(const RbArgs& a, int64_t block_index, double* smem, fm::Tables* tabs,
                                                double& end_a, double& end_b, bool& live_a, bool& live_b, bool& lead) {
    const RbLds L = rb_stage_lds(a, smem, tabs, RB_HALF_LOG2E);
    rb_fft_block<LG, LT>(a, L, block_index, (int)threadIdx.x, tabs, end_a, end_b, live_a, live_b, lead);
}

// Mz < 32 (at most 16 steps): one pair per lane, the transform evaluated directly.
__device__ __forceinline__ void rb_generate_small(const RbArgs& a, int64_t block_index, double* smem, fm::Tables* tabs,
                                                  double& end_a, double& end_b, bool& live_a, bool& live_b, bool& lead) {
    const RbLds L = rb_stage_lds(a, smem, tabs, 1.0);
    const int M = a.M;  // 1, 2, 4, 8 or 16
    const int64_t q = block_index * (int64_t)blockDim.x + threadIdx.x;
    const int64_t col_a = 2 * q;
    live_a = col_a < a.n_paths;
    live_b = col_a + 1 < a.n_paths;
    lead = true;
    const uint64_t id_a = a.path_begin + (uint64_t)col_a, id_b = id_a + 1;
    double yr[16], yi[16];
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        double z[4] = {0.0, 0.0, 0.0, 0.0};
        if (k < M) fm::normal_quad_fast(a.k0, a.k1, id_a >> 1, (uint32_t)(k >> 1), STREAM_VOL, tabs, z);
        const double a0 = k < M ? L.amp[k] : 0.0, a1 = k + 1 < M ? L.amp[k + 1] : 0.0;
        yr[k] = a0 * z[0];
        yi[k] = a0 * z[1];
        yr[k + 1] = a1 * z[2];
        yi[k + 1] = a1 * z[3];
    }
    double* col = a.out + col_a;
    rb_store_pair(col, a.S0, a.S0, live_a, live_b);
    double ls_a = a.logS0, ls_b = a.logS0;
    double za[4], zb[4];
    for (int n = 0; n < a.n_steps; ++n) {
        double re = 0.0, im = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            // e^{2 pi i (k n mod M)/M} from the LDS twiddle table (q < M/2; the other half is its negative)
            const int qn = (k * n) & (M - 1);
            double c = 1.0, sn = 0.0;
            if (M > 1 && k < M) {
                const double2 w = L.tw[qn & (M / 2 - 1)];
                const double sg = (qn >= M / 2) ? -1.0 : 1.0;
                c = sg * w.x;
                sn = sg * w.y;
            }
            re += yr[k] * c - yi[k] * sn;
            im += yr[k] * sn + yi[k] * c;
        }
        if ((n & 3) == 0) {
            fm::normal_quad_fast(a.k0, a.k1, id_a, (uint32_t)(n >> 2), STREAM_PRICE, tabs, za);
            fm::normal_quad_fast(a.k0, a.k1, id_b, (uint32_t)(n >> 2), STREAM_PRICE, tabs, zb);
        }
        const double var_a = fm::scaled_exp(a.xi, re + L.comp[n]), var_b = fm::scaled_exp(a.xi, im + L.comp[n]);
        ls_a += fma(fm::sqrt_pos(fmax(var_a, 1e-300)) * a.sqdt, za[n & 3], (a.r - 0.5 * var_a) * a.dt);
        ls_b += fma(fm::sqrt_pos(fmax(var_b, 1e-300)) * a.sqdt, zb[n & 3], (a.r - 0.5 * var_b) * a.dt);
        end_a = fm::scaled_exp(1.0, ls_a);
        end_b = fm::scaled_exp(1.0, ls_b);
        rb_store_pair(col + (int64_t)(n + 1) * a.ld, end_a, end_b, live_a, live_b);
    }
}

}  // namespace mcg
