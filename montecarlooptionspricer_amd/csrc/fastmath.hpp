// Hand-scheduled binary64 math for the path kernels (gfx950).  The GBM step is issue-bound on the
// fp64 VALU (a wave64 fp64 op takes 4 SIMD cycles; v_rcp/rsq/sqrt_f64 take ~16.5; 32-bit integer
// and bit ops ~2.8 -- tools/ubench_issue.hip), so every function here is written to minimise the
// fp64 instruction count for exactly the argument ranges the kernels produce, with no special-case
// branches (a divergent branch costs both sides on a 64-lane wave).  Accuracy target: <= ~2 ulp,
// checked against mpmath in tests (mcg_debug_eval).
//
//   scaled_exp(S, a)        S * e^a           any finite a (overflow -> inf, underflow -> 0)
//   neg2log(u, tab)         -2 ln u           u in (0, 1); 1024-entry {1/c, -2 ln c} table in LDS
//   sqrt_pos(x)             sqrt(x)           x in [1e-300, 1e300], no denormal/negative handling
//   sincos_table(wb, tab)   cos/sin(2 pi f)   f = ((wb >> 8) + 1/2) 2^-24, from the raw Philox word; 1024-entry table
//   normal_quad_fast(...)   the four normals of one Philox block (philox.hpp's contract)
//
// Polynomials: interpolation at Chebyshev nodes in 60-digit arithmetic, rounded to binary64
// (tools/gen_coeffs.py prints them with their achieved max error).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "philox.hpp"

namespace mcg {
namespace fm {

constexpr int LOG_TAB_ENTRIES = 1024;    // x 16 B = 16 KiB of LDS per workgroup
constexpr int SINCOS_BITS = 10;
constexpr int SINCOS_TAB_ENTRIES = 1 << SINCOS_BITS;  // x 16 B = 16 KiB
constexpr int EXP2_TAB_ENTRIES = 256;     // x 8 B = 2 KiB
constexpr int TABLE_UNITS = LOG_TAB_ENTRIES + SINCOS_TAB_ENTRIES + EXP2_TAB_ENTRIES / 2;  // 16-byte units of all three

// The lookup tables as they sit in LDS (and, back to back, in the device buffer they are staged from:
// fm::LOG_TAB_HOST, fm::SINCOS_TAB_HOST, fm::EXP2_TAB_HOST).
struct Tables {
    static constexpr int SC_BITS = SINCOS_BITS;
    double2 log[LOG_TAB_ENTRIES];        // {1/c_i, -2 ln c_i}
    double2 sincos[SINCOS_TAB_ENTRIES];  // {cos, sin}(2 pi i / 1024)
    double exp2[EXP2_TAB_ENTRIES];       // 2^(j/256)
};
// What a kernel that draws normals only needs of them (the GBM generator: no exp2 table, and -- BITS < 10 -- every
// 2^(10-BITS)-th sin/cos entry): LDS is what decides how many of its workgroups a CU holds.
template <int BITS>
struct NormalTables {
    static_assert(BITS >= 6 && BITS <= SINCOS_BITS, "a sub-sampling of the 1024-entry table");
    static constexpr int SC_BITS = BITS;
    double2 log[LOG_TAB_ENTRIES];
    double2 sincos[1 << BITS];
};

__device__ __forceinline__ double from_words(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }

// q*r + C with the constant C read from a scalar register pair.  hipcc otherwise keeps hoisted
// constants in VGPRs and lowers each Horner step to v_mov_b64 + v_fmac_f64 (the two-address form
// clobbers its addend); one VOP3 v_fma_f64 with an SGPR addend needs no copy.  (Tried: pinning only the constant with
// an empty asm and leaving the FMA to the compiler -- that removes the `s_nop 0` hipcc puts behind an asm FMA whose
// result the next instruction reads, 20 per Philox block, but costs two copies and seven scalar spills: no gain.)
__device__ __forceinline__ double fma_sc(double q, double r, double C) {
    double out;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(q), "v"(r), "s"(C));
    return out;
}

// S * e^a.  k = rint(a/ln2), r = a - k ln2 (hi/lo split; k*hi exact for |k| < 2^21),
// e^r = 1 + r + r^2 q(r) with q of degree 9 on |r| <= ln2/2 (max rel err 2^-55.8),
// result = ldexp(S + S*(e^r - 1), k).  18 fp64-class instructions.
__device__ __forceinline__ double scaled_exp(double S, double a) {
    // Wave-uniform shortcut: when every lane has |a| <= 0.34 (< ln2/2) the range reduction is the
    // identity (k = 0, r = a) and is skipped for the whole wave -- bit-identical to the general path,
    // and the common case for a price step (|drift + vol z| ~ 1e-2).
    const bool reduce = __builtin_amdgcn_ballot_w64(!(__builtin_fabs(a) <= 0.34)) != 0ull;
    double kd = 0.0, r = a;
    if (reduce) {
        asm volatile("" ::);  // keep this a real (scalar) branch: hipcc would otherwise if-convert it
        kd = __builtin_rint(a * 0x1.71547652b82fep+0);
        r = __builtin_fma(kd, -0x1.62e42fee00000p-1, a);
        r = __builtin_fma(kd, -0x1.a39ef35793c76p-33, r);
    }
    double q = 0x1.af38a9b0ec855p-26;
    q = fma_sc(q, r, 0x1.289185613a3d6p-22);
    q = fma_sc(q, r, 0x1.71de0dae63bb3p-19);
    q = fma_sc(q, r, 0x1.a019b90d2ae7ap-16);
    q = fma_sc(q, r, 0x1.a01a01a7c41d5p-13);
    q = fma_sc(q, r, 0x1.6c16c1788bd90p-10);
    q = fma_sc(q, r, 0x1.11111111109b3p-7);
    q = fma_sc(q, r, 0x1.5555555553d63p-5);
    q = fma_sc(q, r, 0x1.5555555555556p-3);
    q = fma_sc(q, r, 0x1.0000000000001p-1);
    const double em1 = __builtin_fma(r * r, q, r);  // e^r - 1
    double v = __builtin_fma(S, em1, S);
    if (reduce) {
        asm volatile("" ::);
        v = __builtin_ldexp(v, (int)kd);  // v_cvt_i32_f64 saturates; v_ldexp_f64 clamps
    }
    return v;
}

// e^a with the range reduction always on (no wave-uniform test): for arguments that are rarely all small,
// e.g. the log-variance and log-price of the rBergomi kernels.
__device__ __forceinline__ double exp_full(double a) {
    const double kd = __builtin_rint(a * 0x1.71547652b82fep+0);
    double r = __builtin_fma(kd, -0x1.62e42fee00000p-1, a);
    r = __builtin_fma(kd, -0x1.a39ef35793c76p-33, r);
    double q = 0x1.af38a9b0ec855p-26;
    q = fma_sc(q, r, 0x1.289185613a3d6p-22);
    q = fma_sc(q, r, 0x1.71de0dae63bb3p-19);
    q = fma_sc(q, r, 0x1.a019b90d2ae7ap-16);
    q = fma_sc(q, r, 0x1.a01a01a7c41d5p-13);
    q = fma_sc(q, r, 0x1.6c16c1788bd90p-10);
    q = fma_sc(q, r, 0x1.11111111109b3p-7);
    q = fma_sc(q, r, 0x1.5555555553d63p-5);
    q = fma_sc(q, r, 0x1.5555555555556p-3);
    q = fma_sc(q, r, 0x1.0000000000001p-1);
    const double em1 = __builtin_fma(r * r, q, r);
    return __builtin_ldexp(1.0 + em1, (int)kd);
}

// ---- two arguments at a time ---------------------------------------------------------------------------------
// The rBergomi kernels run at two to three waves per SIMD, where a chain of dependent fp64 FMAs (issue 4 cycles,
// result later) leaves issue slots empty, and hipcc schedules an asm FMA as an opaque unit: it neither interleaves two
// such chains nor drops the s_nop it puts behind each.  The kernels handle two paths per lane, so every polynomial
// exists twice: one asm block advances both chains by a step.
__device__ __forceinline__ void horner2(double& qa, double& qb, double ra, double rb, double C) {
    asm("v_fma_f64 %0, %0, %2, %4\n\tv_fma_f64 %1, %1, %3, %4" : "+v"(qa), "+v"(qb) : "v"(ra), "v"(rb), "s"(C));
}
// Several steps of both chains in ONE asm statement: behind every asm statement whose result the next instruction
// reads hipcc puts an `s_nop 0` (it cannot see that the statement is a plain FMA); inside a statement there is none,
// and dependent VALU instructions need none.
#define MCG_H2(n) "v_fma_f64 %0, %0, %2, %" #n "\n\tv_fma_f64 %1, %1, %3, %" #n "\n\t"
__device__ __forceinline__ void horner2x3(double& qa, double& qb, double ra, double rb, double c1, double c2, double c3) {
    asm(MCG_H2(4) MCG_H2(5) MCG_H2(6) : "+v"(qa), "+v"(qb) : "v"(ra), "v"(rb), "s"(c1), "s"(c2), "s"(c3));
}
__device__ __forceinline__ void horner2x6(double& qa, double& qb, double ra, double rb, double c1, double c2, double c3,
                                          double c4, double c5, double c6) {
    asm(MCG_H2(4) MCG_H2(5) MCG_H2(6) MCG_H2(7) MCG_H2(8) MCG_H2(9)
        : "+v"(qa), "+v"(qb)
        : "v"(ra), "v"(rb), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6));
}
__device__ __forceinline__ void horner2x9(double& qa, double& qb, double ra, double rb, double c1, double c2, double c3,
                                          double c4, double c5, double c6, double c7, double c8, double c9) {
    asm(MCG_H2(4) MCG_H2(5) MCG_H2(6) MCG_H2(7) MCG_H2(8) MCG_H2(9) MCG_H2(10) MCG_H2(11) MCG_H2(12)
        : "+v"(qa), "+v"(qb)
        : "v"(ra), "v"(rb), "s"(c1), "s"(c2), "s"(c3), "s"(c4), "s"(c5), "s"(c6), "s"(c7), "s"(c8), "s"(c9));
}
__device__ __forceinline__ void horner2x4(double& qa, double& qb, double ra, double rb, double c1, double c2, double c3,
                                          double c4) {
    asm(MCG_H2(4) MCG_H2(5) MCG_H2(6) MCG_H2(7) : "+v"(qa), "+v"(qb) : "v"(ra), "v"(rb), "s"(c1), "s"(c2), "s"(c3), "s"(c4));
}
#undef MCG_H2

// e^a, e^b (exp_full twice, interleaved)
__device__ __forceinline__ void exp_full2(double a, double b, double& ea, double& eb) {
    const double ka = __builtin_rint(a * 0x1.71547652b82fep+0), kb = __builtin_rint(b * 0x1.71547652b82fep+0);
    double ra = __builtin_fma(ka, -0x1.62e42fee00000p-1, a), rb = __builtin_fma(kb, -0x1.62e42fee00000p-1, b);
    ra = __builtin_fma(ka, -0x1.a39ef35793c76p-33, ra);
    rb = __builtin_fma(kb, -0x1.a39ef35793c76p-33, rb);
    double qa = 0x1.af38a9b0ec855p-26, qb = 0x1.af38a9b0ec855p-26;
    horner2x9(qa, qb, ra, rb, 0x1.289185613a3d6p-22, 0x1.71de0dae63bb3p-19, 0x1.a019b90d2ae7ap-16, 0x1.a01a01a7c41d5p-13,
              0x1.6c16c1788bd90p-10, 0x1.11111111109b3p-7, 0x1.5555555553d63p-5, 0x1.5555555555556p-3,
              0x1.0000000000001p-1);
    ea = __builtin_ldexp(1.0 + __builtin_fma(ra * ra, qa, ra), (int)ka);
    eb = __builtin_ldexp(1.0 + __builtin_fma(rb * rb, qb, rb), (int)kb);
}

// 2^(ta/256), 2^(tb/256) for arguments that are ALREADY in units of (ln 2)/256 (the caller folds 256 log2(e) into whatever
// produces them: the rBergomi variance factor is 2^((c X + table)/256)): n = rint(t) = 256 k + j, g = t - n exactly,
// 2^(t/256) = 2^k * 2^(j/256) * 2^(g/256), the middle factor from a 256-entry LDS table, the last as 1 + g h(g) with h of
// degree 3 on |g| <= 1/2 (max rel err 2^-57.5, tools/gen_coeffs.py).  9 fp64 instructions per value (+3 integer, one
// LDS read) against the 10 of the 64-entry table this replaced in round 3, the 15 of a degree-10 polynomial on the whole
// octave and the 20 of exp_full2 -- and fp64 instructions are what the generator's clock pays for.
__device__ __forceinline__ void exp2_pair(double ta, double tb, const Tables* tab, double& ea, double& eb) {
    const double na = __builtin_rint(ta), nb = __builtin_rint(tb);
    const int ia = (int)na, ib = (int)nb;  // v_cvt_i32_f64 saturates: |t| beyond 2^31 ends in ldexp's clamp either way
    const double Ta = tab->exp2[ia & 255], Tb = tab->exp2[ib & 255];
    const double ga = ta - na, gb = tb - nb;
    double qa = 0x1.3b2ab83ecf101p-39, qb = 0x1.3b2ab83ecf101p-39;
    horner2x3(qa, qb, ga, gb, 0x1.c6b0902ba1a20p-29, 0x1.ebfbdff82c585p-19, 0x1.62e42fefa39d9p-9);
    ea = __builtin_ldexp(__builtin_fma(Ta * ga, qa, Ta), ia >> 8);
    eb = __builtin_ldexp(__builtin_fma(Tb * gb, qb, Tb), ib >> 8);
}

// e^a - 1, e^b - 1 for |a|, |b| <= 0.1 (the polynomial of scaled_exp_small6) and for <= 0.34 (that of scaled_exp
// without its range reduction): a price step's exponent, whose size the caller has tested for the whole wave.
__device__ __forceinline__ void expm1_small6_2(double a, double b, double& ea, double& eb) {
    double qa = 0x1.a02eb88e6a6ffp-16, qb = 0x1.a02eb88e6a6ffp-16;
    horner2x6(qa, qb, a, b, 0x1.a033e66a22569p-13, 0x1.6c16c10206fa6p-10, 0x1.1111108c7c825p-7, 0x1.555555555664ep-5,
              0x1.5555555557fc2p-3, 0x1.0000000000000p-1);
    ea = __builtin_fma(a * a, qa, a);
    eb = __builtin_fma(b * b, qb, b);
}
__device__ __forceinline__ void expm1_small9_2(double a, double b, double& ea, double& eb) {
    double qa = 0x1.af38a9b0ec855p-26, qb = 0x1.af38a9b0ec855p-26;
    horner2x9(qa, qb, a, b, 0x1.289185613a3d6p-22, 0x1.71de0dae63bb3p-19, 0x1.a019b90d2ae7ap-16, 0x1.a01a01a7c41d5p-13,
              0x1.6c16c1788bd90p-10, 0x1.11111111109b3p-7, 0x1.5555555553d63p-5, 0x1.5555555555556p-3,
              0x1.0000000000001p-1);
    ea = __builtin_fma(a * a, qa, a);
    eb = __builtin_fma(b * b, qb, b);
}

// S * e^a for |a| <= 0.125, guaranteed by the caller from the step's parameters (a GBM step has
// |a| <= |drift| + vol * 7.55: the 40-bit radius uniform caps |z| at sqrt(2*41*ln2) = 7.54).  No range
// reduction, no branch; e^a = 1 + a + a^2 q(a) with q of degree 7 (max rel err 2^-58.7 on the interval).
constexpr double SMALL_EXP_BOUND = 0.125;
constexpr double MAX_ABS_NORMAL = 7.55;
__device__ __forceinline__ double scaled_exp_small(double S, double a) {
    double q = 0x1.71f9218da2e29p-19;
    q = fma_sc(q, a, 0x1.a03effca19d70p-16);
    q = fma_sc(q, a, 0x1.a01a00930ee10p-13);
    q = fma_sc(q, a, 0x1.6c16bffa2459dp-10);
    q = fma_sc(q, a, 0x1.11111111146e0p-7);
    q = fma_sc(q, a, 0x1.555555555e951p-5);
    q = fma_sc(q, a, 0x1.5555555555555p-3);
    q = fma_sc(q, a, 0x1.ffffffffffffep-2);
    const double em1 = __builtin_fma(a * a, q, a);
    return __builtin_fma(S, em1, S);
}

// S * e^a for |a| <= 0.1: one term less (q of degree 6, max rel err 2^-54.2 = 0.4 ulp before rounding).
constexpr double SMALL6_EXP_BOUND = 0.1;
__device__ __forceinline__ double scaled_exp_small6(double S, double a) {
    double q = 0x1.a02eb88e6a6ffp-16;
    q = fma_sc(q, a, 0x1.a033e66a22569p-13);
    q = fma_sc(q, a, 0x1.6c16c10206fa6p-10);
    q = fma_sc(q, a, 0x1.1111108c7c825p-7);
    q = fma_sc(q, a, 0x1.555555555664ep-5);
    q = fma_sc(q, a, 0x1.5555555557fc2p-3);
    q = fma_sc(q, a, 0x1.0000000000000p-1);
    const double em1 = __builtin_fma(a * a, q, a);
    return __builtin_fma(S, em1, S);
}

// -2 ln u for u in (0,1).  u = z * 2^k with z = frexp mantissa in [0.5, 1) (v_frexp_exp_i32_f64, v_frexp_mant_f64); i = top ten mantissa bits = interval of z, r = z/c_i - 1 via one FMA with the tabulated 1/c_i;
// ln z = ln c_i + log1p(r), log1p(r) = r + r^2 q(r), q of degree 3 on |r| <= 2^-11 (max rel err 2^-60.4; 1024 intervals -- with 128 it took degree 5, two FMAs more per pair).
// The last interval [1 - 2^-11, 1) has c = 1 exactly, so u -> 1 keeps full relative accuracy (no cancellation).
// tab: LDS, entry i = {1/c_i, -2 ln c_i}.
struct LogSplit {
    double z;      // mantissa in [0.5, 1)
    int k;         // exponent: u = z 2^k
    uint32_t idx;  // table interval
};
__device__ __forceinline__ LogSplit log_split(double u) {
    const uint32_t hi = (uint32_t)__double2hiint(u);
    LogSplit s;
    s.k = __builtin_amdgcn_frexp_exp(u);
    s.idx = (hi >> 10) & 1023u;
    s.z = __builtin_amdgcn_frexp_mant(u);  // (a v_and_or on the high word costs two register copies on top)
    return s;
}

// -2 log1p(r) = r P(r), P(r) = -2 - 2 r q(r) = fma(r, Q(r), -2) with Q = -2 q: one instruction less than
// -2 (r + r^2 q(r)) -- no r^2 -- and exact doubling of q's coefficients (tools/gen_coeffs.py: LOG_Q2).
constexpr double LOG_Q2_3 = -0x1.99999e5b2a5bcp-2, LOG_Q2_2 = 0x1.000002c63f1bap-1, LOG_Q2_1 = -0x1.5555555555542p-1,
                 LOG_Q2_0 = 0x1.fffffffffffe9p-1;

// (split, table entry) -> -2 ln u = -2 k ln2 + (-2 ln c) + r P(r)
__device__ __forceinline__ double neg2log_entry(const LogSplit& sp, const double2 e) {
    const double r = __builtin_fma(sp.z, e.x, -1.0);
    double q = LOG_Q2_3;
    q = fma_sc(q, r, LOG_Q2_2);
    q = fma_sc(q, r, LOG_Q2_1);
    q = fma_sc(q, r, LOG_Q2_0);
    const double P = __builtin_fma(r, q, -2.0);
    const double base = __builtin_fma((double)sp.k, -0x1.62e42fefa39efp+0, e.y);
    return __builtin_fma(r, P, base);
}
// two of them, the polynomial chains interleaved (bit-identical to neg2log_entry twice)
__device__ __forceinline__ void neg2log_entry2(const LogSplit& s0, const double2 e0, const LogSplit& s1, const double2 e1,
                                               double& out0, double& out1) {
    const double r0 = __builtin_fma(s0.z, e0.x, -1.0), r1 = __builtin_fma(s1.z, e1.x, -1.0);
    double q0 = LOG_Q2_3, q1 = LOG_Q2_3;
    horner2x3(q0, q1, r0, r1, LOG_Q2_2, LOG_Q2_1, LOG_Q2_0);
    const double P0 = __builtin_fma(r0, q0, -2.0), P1 = __builtin_fma(r1, q1, -2.0);
    out0 = __builtin_fma(r0, P0, __builtin_fma((double)s0.k, -0x1.62e42fefa39efp+0, e0.y));
    out1 = __builtin_fma(r1, P1, __builtin_fma((double)s1.k, -0x1.62e42fefa39efp+0, e1.y));
}
__device__ __forceinline__ double neg2log(double u, const double2* tab) {
    const LogSplit sp = log_split(u);
    return neg2log_entry(sp, tab[sp.idx]);
}

// vol^2 * (-2 ln u): the same evaluation with the scale folded into its constants -- the table's second column is
// pre-multiplied when it is staged (load_tables_scaled), and the host hands over c_k = -2 ln2 vol^2, c_l = -2 vol^2 and
// the polynomial's coefficients times vol^2 -- so that the square root yields vol * sqrt(-2 ln u) directly and the GBM
// step saves a multiply:  vol^2 (-2 log1p(r)) = r (c_l + r (vol^2 Q(r))).
struct LogScale {
    double c_k, c_l, q3, q2, q1, q0;
};
__host__ __device__ inline LogScale make_log_scale(double vol2) {
    return LogScale{-0x1.62e42fefa39efp+0 * vol2, -2.0 * vol2, LOG_Q2_3 * vol2, LOG_Q2_2 * vol2, LOG_Q2_1 * vol2, LOG_Q2_0 * vol2};
}
__device__ __forceinline__ double neg2log_scaled(double u, const double2* tab, const LogScale& L) {
    const LogSplit sp = log_split(u);
    const double2 e = tab[sp.idx];
    const double r = __builtin_fma(sp.z, e.x, -1.0);
    double q = L.q3;
    q = fma_sc(q, r, L.q2);
    q = fma_sc(q, r, L.q1);
    q = fma_sc(q, r, L.q0);
    const double P = __builtin_fma(r, q, L.c_l);
    const double base = __builtin_fma((double)sp.k, L.c_k, e.y);
    return __builtin_fma(r, P, base);
}

// sqrt for positive normal x in five instructions (measured <= 0.75 ulp): with y = rsq(x) accurate to
// ~2^-23 and e = 1 - x y^2, sqrt(x) = x y (1 - e)^(-1/2) = g (1 + e/2 + 3e^2/8 + O(e^3)), g = x y;
// the cubic term is < 2^-69.
__device__ __forceinline__ double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    const double e = __builtin_fma(-g, y, 1.0);
    const double p = __builtin_fma(e, 0.375, 0.5);
    return __builtin_fma(g * e, p, g);
}

// cos(2 pi f), sin(2 pi f) with f = ((wb >> 8) + 1/2) * 2^-24, straight from the Philox word:
// 2 pi f = 2 pi i/1024 + delta with i the top 10 angle bits and delta = 2 pi (low14 + 1/2) 2^-24 in
// (0, 0.006136).  (cos, sin)(2 pi i/1024) come from a 1024-entry LDS table (correctly rounded),
// sin(delta) from a 3-term and cos(delta) from a 3-term series (1 - d^2/2 + d^4/24: truncation 7.4e-17; 6.5e-20 for the
// sine), combined by the angle-addition formulas: 12 fp64 instructions and no octant logic.  (512 entries and one more
// cosine term until round 3.)
// the table-independent half: (cos, sin)(delta)
// BITS < 10 (a sub-sampled table, NormalTables): delta < 2 pi / 2^BITS, the cosine keeps its d^6 term (512 entries:
// truncation 1.3e-20; the sine's 8.3e-18).
template <int BITS = SINCOS_BITS>
__device__ __forceinline__ void sincos_small(uint32_t wb, double& cd_out, double& sd_out) {
    const double delta = __builtin_fma((double)((wb >> 8) & ((1u << (24 - BITS)) - 1u)), 0x1.921fb54442d18p-22, 0x1.921fb54442d18p-23);
    const double d2 = delta * delta;
    const double ts = __builtin_fma(d2, 0x1.1111111111111p-7, -0x1.5555555555555p-3);  // 1/120, -1/6
    sd_out = __builtin_fma(delta * d2, ts, delta);
    double tc;
    if constexpr (BITS >= SINCOS_BITS) {
        tc = __builtin_fma(d2, 0x1.5555555555555p-5, -0.5);                            // 1/24, -1/2
    } else {
        tc = __builtin_fma(d2, -0x1.6c16c16c16c17p-10, 0x1.5555555555555p-5);           // -1/720, 1/24
        tc = __builtin_fma(tc, d2, -0.5);
    }
    cd_out = __builtin_fma(tc, d2, 1.0);
}
template <int BITS = SINCOS_BITS>
__device__ __forceinline__ void sincos_entry(uint32_t wb, const double2 e, double& c_out, double& s_out) {
    double cd, sd;
    sincos_small<BITS>(wb, cd, sd);
    c_out = __builtin_fma(e.x, cd, -(e.y * sd));
    s_out = __builtin_fma(e.y, cd, e.x * sd);
}
template <int BITS = SINCOS_BITS>
__device__ __forceinline__ void sincos_table(uint32_t wb, const double2* sc_tab, double& c_out, double& s_out) {
    sincos_entry<BITS>(wb, sc_tab[wb >> (32 - BITS)], c_out, s_out);
}

// One Box-Muller pair from 64 Philox bits (philox.hpp contract).
__device__ __forceinline__ void box_muller_pair(uint32_t wa, uint32_t wb, const Tables* tab, double& z0,
                                                double& z1) {
    const double rad = sqrt_pos(neg2log(radius_u01(wa, wb), tab->log));
    double c, s;
    sincos_table(wb, tab->sincos, c, s);
    z0 = rad * c;
    z1 = rad * s;
}

// The same pair, already scaled and shifted: a0 = shift + scale*z0, a1 = shift + scale*z1
// (the exponent of a price step).  Folding scale into the radius saves one multiply per pair.
template <class T>
__device__ __forceinline__ void box_muller_pair_affine(uint32_t wa, uint32_t wb, const T* tab, double scale,
                                                       double shift, double& a0, double& a1) {
    const double rad = scale * sqrt_pos(neg2log(radius_u01(wa, wb), tab->log));
    double c, s;
    sincos_table<T::SC_BITS>(wb, tab->sincos, c, s);
    a0 = __builtin_fma(rad, c, shift);
    a1 = __builtin_fma(rad, s, shift);
}

// The affine pair with vol folded into the logarithm (tables staged by load_tables_scaled(vol^2)); vol > 0.
template <class T>
__device__ __forceinline__ void box_muller_pair_affine_scaled(uint32_t wa, uint32_t wb, const T* tab, const LogScale& L,
                                                              double shift, double& a0, double& a1) {
    const double rad = sqrt_pos(neg2log_scaled(radius_u01(wa, wb), tab->log, L));  // = vol * sqrt(-2 ln u)
    double c, s;
    sincos_table<T::SC_BITS>(wb, tab->sincos, c, s);
    a0 = __builtin_fma(rad, c, shift);
    a1 = __builtin_fma(rad, s, shift);
}

// One Philox block -> four N(0,1) deviates.
// EAGER: request all four table entries before any is used (costs ~12 registers for the time of the lookups).
// amp0, amp1: factors of the first and of the second pair (the rBergomi spectrum's amplitudes a_k: folded into the radius,
// one multiply per pair instead of one per deviate); 1 for plain deviates.
template <bool EAGER = false>
__device__ __forceinline__ void normal_quad_words(const Philox4 w, const Tables* tab, double (&z)[4], double amp0, double amp1);

template <bool EAGER = false>
__device__ __forceinline__ void normal_quad_fast(uint32_t k0, uint32_t k1, uint64_t path, uint32_t block,
                                                 uint32_t stream, const Tables* tab, double (&z)[4], double amp0 = 1.0,
                                                 double amp1 = 1.0) {
    normal_quad_words<EAGER>(philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), block, stream, k0, k1), tab, z, amp0, amp1);
}
// the same from the hoisted per-path part of the block function (philox_lane_setup(path, stream, k1))
template <bool EAGER = false>
__device__ __forceinline__ void normal_quad_fast(uint32_t k0, uint32_t k1, const PhiloxLane& L, uint32_t block, const Tables* tab,
                                                 double (&z)[4], double amp0 = 1.0, double amp1 = 1.0) {
    normal_quad_words<EAGER>(philox4x32_10_path(L, block, k0, k1), tab, z, amp0, amp1);
}

template <bool EAGER>
__device__ __forceinline__ void normal_quad_words(const Philox4 w, const Tables* tab, double (&z)[4], double amp0, double amp1) {
    if constexpr (!EAGER) {
        box_muller_pair_affine(w.w0, w.w1, tab, amp0, 0.0, z[0], z[1]);
        box_muller_pair_affine(w.w2, w.w3, tab, amp1, 0.0, z[2], z[3]);
        return;
    }
    // All four table entries are requested before any of them is used: a lookup is ~100 cycles of LDS latency, and a
    // kernel at two waves per SIMD has little else to issue meanwhile (requested one by one, each was waited for).
    const LogSplit s0 = log_split(radius_u01(w.w0, w.w1)), s1 = log_split(radius_u01(w.w2, w.w3));
    double2 a0 = tab->sincos[w.w1 >> (32 - SINCOS_BITS)], a1 = tab->sincos[w.w3 >> (32 - SINCOS_BITS)], l0 = tab->log[s0.idx], l1 = tab->log[s1.idx];
    double cd0, sd0, cd1, sd1;  // meanwhile: the halves that need no table
    sincos_small(w.w1, cd0, sd0);
    sincos_small(w.w3, cd1, sd1);
    __builtin_amdgcn_sched_barrier(0);
    double n0, n1;
    neg2log_entry2(s0, l0, s1, l1, n0, n1);
    const double r0 = amp0 * sqrt_pos(n0), r1 = amp1 * sqrt_pos(n1);
    z[0] = r0 * __builtin_fma(a0.x, cd0, -(a0.y * sd0));
    z[1] = r0 * __builtin_fma(a0.y, cd0, a0.x * sd0);
    z[2] = r1 * __builtin_fma(a1.x, cd1, -(a1.y * sd1));
    z[3] = r1 * __builtin_fma(a1.y, cd1, a1.x * sd1);
}

// Cooperative copy of the tables (global, 34 KiB) into LDS; call before the first normal and
// follow with __syncthreads().
__device__ __forceinline__ void load_tables(Tables* lds, const double2* __restrict__ gtab) {
    double2* dst = reinterpret_cast<double2*>(lds);
    for (int i = threadIdx.x; i < TABLE_UNITS; i += blockDim.x) dst[i] = gtab[i];
}

// The same copy with the logarithm table's second column multiplied by `scale` (neg2log_scaled).
__device__ __forceinline__ void load_tables_scaled(Tables* lds, const double2* __restrict__ gtab, double scale) {
    double2* dst = reinterpret_cast<double2*>(lds);
    for (int i = threadIdx.x; i < TABLE_UNITS; i += blockDim.x) {
        double2 e = gtab[i];
        if (i < LOG_TAB_ENTRIES) e.y *= scale;
        dst[i] = e;
    }
}

// NormalTables<BITS> from the same device buffer (logarithm table, then every 2^(10-BITS)-th sin/cos entry); the
// logarithm table's second column times `scale` (1: as it is).
template <int BITS>
__device__ __forceinline__ void load_normal_tables(NormalTables<BITS>* lds, const double2* __restrict__ gtab, double scale) {
    for (int i = threadIdx.x; i < LOG_TAB_ENTRIES; i += blockDim.x) {
        double2 e = gtab[i];
        e.y *= scale;
        lds->log[i] = e;
    }
    for (int i = threadIdx.x; i < (1 << BITS); i += blockDim.x) lds->sincos[i] = gtab[LOG_TAB_ENTRIES + (i << (SINCOS_BITS - BITS))];
}

}  // namespace fm
}  // namespace mcg
