// Column-accumulator arithmetic for bn256::Fr on gfx950 (device only) - the 256-bit counterpart of gl_wide.hpp.
//
// hipcc compiles the 4x64 CIOS product (bn254_field.hpp: fr_mul) to ~730 instructions: 132 v_mad_u64_u32, and around them carry
// chains (v_addc_co), register moves and the s_nop slots of the VALU -> SGPR -> VALU carry hazard. Here a product is kept in
// "columns" instead: with 32-bit limbs a_i, b_j the partial product a_i b_j is added by ONE v_mad_u64_u32 into the 64-bit
// accumulator C[i+j] (weight 2^(32(i+j))), and the carry-out of that addition is banked by ONE v_addc_co_u32 in the counter
// T[i+j] (weight 2^(32(i+j)+64)). No carry is propagated until the value is reduced:
//
//   value = sum_k C[k] 2^(32k) + sum_k T[k] 2^(32k+64)
//
// A multiply-accumulate is 64 mad + 64 addc; sums of products (the dot products of a sum-check round) share one Montgomery
// reduction, which works on the columns directly: limb i is made exact, m = limb * (-r^-1) mod 2^32, and m r is added with the
// same mad/addc pairs. The asm blocks keep the gfx950 hazard distance themselves (four mads, then their four addc).
#pragma once
#include "bn254_field.hpp"

namespace hg {
namespace bn {

typedef uint32_t u32;

struct WCol {
    u64 C[16];   // C[15] only receives reduction carries (the products stop at column 14)
    u32 T[16];
};
__device__ __forceinline__ WCol wcol_zero() {
    WCol w;
#pragma unroll
    for (int k = 0; k < 16; k++) { w.C[k] = 0; w.T[k] = 0; }
    return w;
}

// C[k..k+3] += x * y[0..3], carries banked in T[k..k+3]
#define BN_WIDE_ROW4(c0, c1, c2, c3, t0, t1, t2, t3, x, y0, y1, y2, y3)                                                      \
    do {                                                                                                                     \
        u64 s0_, s1_, s2_, s3_;                                                                                              \
        asm("v_mad_u64_u32 %0, %8, %12, %13, %0\n\t"                                                                         \
            "v_mad_u64_u32 %1, %9, %12, %14, %1\n\t"                                                                         \
            "v_mad_u64_u32 %2, %10, %12, %15, %2\n\t"                                                                        \
            "v_mad_u64_u32 %3, %11, %12, %16, %3\n\t"                                                                        \
            "v_addc_co_u32_e64 %4, %8, 0, %4, %8\n\t"                                                                        \
            "v_addc_co_u32_e64 %5, %9, 0, %5, %9\n\t"                                                                        \
            "v_addc_co_u32_e64 %6, %10, 0, %6, %10\n\t"                                                                      \
            "v_addc_co_u32_e64 %7, %11, 0, %7, %11"                                                                          \
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "=&s"(s0_), "=&s"(s1_), "=&s"(s2_), \
              "=&s"(s3_)                                                                                                     \
            : "v"(x), "v"(y0), "v"(y1), "v"(y2), "v"(y3));                                                                   \
    } while (0)

// w += a * b (plain integer product of two residues)
__device__ __forceinline__ void wcol_mac(WCol& w, const Fr& a, const Fr& b) {
    u32 al[8], bl[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        al[2 * i] = (u32)a.l[i]; al[2 * i + 1] = (u32)(a.l[i] >> 32);
        bl[2 * i] = (u32)b.l[i]; bl[2 * i + 1] = (u32)(b.l[i] >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        BN_WIDE_ROW4(w.C[i], w.C[i + 1], w.C[i + 2], w.C[i + 3], w.T[i], w.T[i + 1], w.T[i + 2], w.T[i + 3], al[i], bl[0], bl[1], bl[2], bl[3]);
        BN_WIDE_ROW4(w.C[i + 4], w.C[i + 5], w.C[i + 6], w.C[i + 7], w.T[i + 4], w.T[i + 5], w.T[i + 6], w.T[i + 7], al[i], bl[4], bl[5], bl[6],
                     bl[7]);
    }
}

// w += x * b for a plain 64-bit integer x (table entries of the integer kernels: addresses, counters, limb values): two limb rows
__device__ __forceinline__ void wcol_mac_u64(WCol& w, u64 x, const Fr& b) {
    u32 bl[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { bl[2 * i] = (u32)b.l[i]; bl[2 * i + 1] = (u32)(b.l[i] >> 32); }
    const u32 xl[2] = {(u32)x, (u32)(x >> 32)};
#pragma unroll
    for (int i = 0; i < 2; i++) {
        BN_WIDE_ROW4(w.C[i], w.C[i + 1], w.C[i + 2], w.C[i + 3], w.T[i], w.T[i + 1], w.T[i + 2], w.T[i + 3], xl[i], bl[0], bl[1], bl[2], bl[3]);
        BN_WIDE_ROW4(w.C[i + 4], w.C[i + 5], w.C[i + 6], w.C[i + 7], w.T[i + 4], w.T[i + 5], w.T[i + 6], w.T[i + 7], xl[i], bl[4], bl[5], bl[6],
                     bl[7]);
    }
}

// running normalisation: 96-bit accumulator (lo, hi) at the weight of the current column
struct WRun { u64 lo; u32 hi; };
__device__ __forceinline__ void wrun_add(WRun& r, u64 v) {
    const u64 s = r.lo + v;
    r.hi += s < v ? 1u : 0u;
    r.lo = s;
}
__device__ __forceinline__ void wrun_shift(WRun& r) {  // next column
    r.lo = (r.lo >> 32) | ((u64)r.hi << 32);
    r.hi = 0;
}

constexpr u32 FR_INV32 = 0xefffffffu;  // -r^-1 mod 2^32 (low half of FR_INV)

// value(w) R^-1 mod r in [0, r) for value(w) < 2^10 r^2 (R = 2^256): Montgomery reduction on the columns
__device__ __forceinline__ Fr wcol_reduce(WCol& w) {
    const u32 pl[8] = {(u32)FR_P0, (u32)(FR_P0 >> 32), (u32)FR_P1, (u32)(FR_P1 >> 32), (u32)FR_P2, (u32)(FR_P2 >> 32), (u32)FR_P3, (u32)(FR_P3 >> 32)};
    WRun run;
    run.lo = 0; run.hi = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        wrun_add(run, w.C[i]);
        if (i >= 2) wrun_add(run, (u64)w.T[i - 2]);
        const u32 m = (u32)run.lo * FR_INV32;
        wrun_add(run, (u64)m * pl[0]);  // clears the low 32 bits of the running value
        // m * r[1..7] into columns i+1 .. i+7
        BN_WIDE_ROW4(w.C[i + 1], w.C[i + 2], w.C[i + 3], w.C[i + 4], w.T[i + 1], w.T[i + 2], w.T[i + 3], w.T[i + 4], m, pl[1], pl[2], pl[3], pl[4]);
        {
            u64 s0_, s1_, s2_;
            asm("v_mad_u64_u32 %0, %6, %9, %10, %0\n\t"
                "v_mad_u64_u32 %1, %7, %9, %11, %1\n\t"
                "v_mad_u64_u32 %2, %8, %9, %12, %2\n\t"
                "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
                "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
                "v_addc_co_u32_e64 %5, %8, 0, %5, %8"
                : "+v"(w.C[i + 5]), "+v"(w.C[i + 6]), "+v"(w.C[i + 7]), "+v"(w.T[i + 5]), "+v"(w.T[i + 6]), "+v"(w.T[i + 7]), "=&s"(s0_), "=&s"(s1_),
                  "=&s"(s2_)
                : "v"(m), "v"(pl[5]), "v"(pl[6]), "v"(pl[7]));
        }
        wrun_shift(run);
    }
    // columns 8 .. 17 -> limbs of the result (before the final subtractions): 8 limbs + overflow
    u32 limb[8];
    u64 over = 0;
#pragma unroll
    for (int k = 8; k < 18; k++) {
        if (k < 16) wrun_add(run, w.C[k]);
        if (k - 2 < 16) wrun_add(run, (u64)w.T[k - 2]);
        if (k < 16) limb[k - 8] = (u32)run.lo;
        else over |= (u64)(u32)run.lo << (32 * (k - 16));
        wrun_shift(run);
    }
    Fr r = fr_make((u64)limb[0] | ((u64)limb[1] << 32), (u64)limb[2] | ((u64)limb[3] << 32), (u64)limb[4] | ((u64)limb[5] << 32),
                   (u64)limb[6] | ((u64)limb[7] << 32));
    // value = over 2^256 + r < 2^10 r: subtract r while it does not fit or is >= r (a handful of iterations at most for sums of
    // a few dozen products: every product contributes less than r / 5 after the division by R)
    for (int it = 0; it < 1100 && (over || fr_geq_p(r)); it++) {
        const bool borrow = !fr_geq_p(r);
        r = fr_sub_p(r);
        if (borrow) over--;
    }
    return r;
}

// ---- x + r d for a multiplier r that is FIXED for a whole launch (the fold of a sum-check round) -----------------------------------
// A general Montgomery product is 64 + 64 multiply-adds (product, reduction). With r fixed, the reduction moves into eight
// precomputed constants K_i = r 2^(32 i) R^-1 mod p (plain integers below p, FoldK::k[8 i ..]): for residues in Montgomery form
//   r d R^-1 = sum_i d_i K_i   (d_i = the 32-bit limbs of d),
// 64 multiply-adds into EIGHT columns, and what is left to reduce is x + sum < 2^36 p: one quotient estimate (a double
// multiplication by 2^224 / p, exact to well below one unit for quotients under 2^36, rounded down from slightly below), q p
// subtracted with 16 multiply-adds, one conditional subtraction of p. The constants are read through a uniform address (scalar loads: they sit in
// SGPRs, one per v_mad_u64_u32 as its single constant-bus operand). About half the instructions of fr_add(x, fr_mul_wide(r, d)).
struct FoldK { u32 k[64]; };
// K_i = REDC(r * 2^(32 i)), host or device (r in Montgomery form)
BN_HD void fold_consts(const Fr& r, FoldK* out) {
    for (int i = 0; i < 8; i++) {
        Fr e = fr_make(0, 0, 0, 0);
        e.l[i >> 1] = 1ULL << (32 * (i & 1));
        const Fr ki = fr_mul(r, e);
        for (int q = 0; q < 4; q++) { out->k[8 * i + 2 * q] = (u32)ki.l[q]; out->k[8 * i + 2 * q + 1] = (u32)(ki.l[q] >> 32); }
    }
}
// C[k..k+3] += x * y[0..3] with the y in SGPRs, carries banked in T[k..k+3]
#define BN_WIDE_ROW4S(c0, c1, c2, c3, t0, t1, t2, t3, x, y0, y1, y2, y3)                                                     \
    do {                                                                                                                     \
        u64 s0_, s1_, s2_, s3_;                                                                                              \
        asm("v_mad_u64_u32 %0, %8, %12, %13, %0\n\t"                                                                         \
            "v_mad_u64_u32 %1, %9, %12, %14, %1\n\t"                                                                         \
            "v_mad_u64_u32 %2, %10, %12, %15, %2\n\t"                                                                        \
            "v_mad_u64_u32 %3, %11, %12, %16, %3\n\t"                                                                        \
            "v_addc_co_u32_e64 %4, %8, 0, %4, %8\n\t"                                                                        \
            "v_addc_co_u32_e64 %5, %9, 0, %5, %9\n\t"                                                                        \
            "v_addc_co_u32_e64 %6, %10, 0, %6, %10\n\t"                                                                      \
            "v_addc_co_u32_e64 %7, %11, 0, %7, %11"                                                                          \
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "=&s"(s0_), "=&s"(s1_), "=&s"(s2_), \
              "=&s"(s3_)                                                                                                     \
            : "v"(x), "s"(y0), "s"(y1), "s"(y2), "s"(y3));                                                                   \
    } while (0)
// value = sum_k C[k] 2^(32 k) + sum_k T[k] 2^(32 k + 64) < 2^36 p  ->  value mod p in [0, p)
__device__ __forceinline__ Fr wfold_finish(const u64* C, const u32* T) {
    // ten limbs
    u32 L[10];
    WRun run;
    run.lo = 0; run.hi = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        if (k < 8) wrun_add(run, C[k]);
        if (k >= 2) wrun_add(run, (u64)T[k - 2]);
        L[k] = (u32)run.lo;
        wrun_shift(run);
    }
    // q = floor(value / p) estimated from the bits above 2^224, one unit low (so that the remainder is in [0, 3p))
    const double top = (double)L[9] * 18446744073709551616.0 + (double)(((u64)L[8] << 32) | L[7]);
    const double qd = top * 1.2317090423844144e-09;   // 2^224 / p (p = 0x30644e72e131a029... ~ 2^253.597)
    // |qd - value / p| < 2^-15 (53-bit mantissa on a quotient below 2^36, 2^224 / p ~ 1e-9 for the dropped low bits): rounding
    // down from qd - 2^-10 gives floor(value / p) or one less, never more
    const u64 q = qd > 0.0009765625 ? (u64)(qd - 0.0009765625) : 0;
    // value - q p: q = q_hi 2^32 + q_lo, q_hi < 16
    const u32 pl[8] = {(u32)FR_P0, (u32)(FR_P0 >> 32), (u32)FR_P1, (u32)(FR_P1 >> 32), (u32)FR_P2, (u32)(FR_P2 >> 32), (u32)FR_P3, (u32)(FR_P3 >> 32)};
    const u32 qlo = (u32)q, qhi = (u32)(q >> 32);
    u32 Q[10];
    {
        u64 carry = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { const u64 t = (u64)qlo * pl[k] + carry; Q[k] = (u32)t; carry = t >> 32; }
        Q[8] = (u32)carry; Q[9] = 0;
        carry = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { const u64 t = (u64)qhi * pl[k] + Q[k + 1] + carry; Q[k + 1] = (u32)t; carry = t >> 32; }
        Q[9] = (u32)carry;
    }
    u32 Rl[8];
    {
        long long borrow = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { const long long t = (long long)L[k] - (long long)Q[k] + borrow; Rl[k] = (u32)t; borrow = t >> 32; }
        // (limbs 8, 9 of the difference are zero: the remainder is below 3p < 2^256)
    }
    Fr r = fr_make((u64)Rl[0] | ((u64)Rl[1] << 32), (u64)Rl[2] | ((u64)Rl[3] << 32), (u64)Rl[4] | ((u64)Rl[5] << 32), (u64)Rl[6] | ((u64)Rl[7] << 32));
    {   // remainder in [0, 2p): one branch-free subtraction of p
        const Fr m = fr_sub_p(r);
        const bool ge = fr_geq_p(r);
        r = fr_make(ge ? m.l[0] : r.l[0], ge ? m.l[1] : r.l[1], ge ? m.l[2] : r.l[2], ge ? m.l[3] : r.l[3]);
    }
    return r;
}

// a KA + v KV + t KT + add mod p for 32-bit integers a, v, t and residues KA, KV, KT, add in [0, p) given as eight 32-bit limbs each
// behind wave-uniform pointers (K[0..7], K[8..15], K[16..23], K[24..31]): the multiset hash of the Lasso memory checking,
// a + v gamma + t gamma^2 - tau, lands in Montgomery form directly when the constants are R, gamma R, gamma^2 R, p - tau R -
// 24 multiply-adds and the short reduction instead of three short products and a Montgomery reduction.
__device__ __forceinline__ Fr fr_lin3_const(u32 a, u32 v, u32 t, const u32* __restrict__ K) {
    u64 C[8];
    u32 T[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { C[i] = K[24 + i]; T[i] = 0; }
    BN_WIDE_ROW4S(C[0], C[1], C[2], C[3], T[0], T[1], T[2], T[3], a, K[0], K[1], K[2], K[3]);
    BN_WIDE_ROW4S(C[4], C[5], C[6], C[7], T[4], T[5], T[6], T[7], a, K[4], K[5], K[6], K[7]);
    BN_WIDE_ROW4S(C[0], C[1], C[2], C[3], T[0], T[1], T[2], T[3], v, K[8], K[9], K[10], K[11]);
    BN_WIDE_ROW4S(C[4], C[5], C[6], C[7], T[4], T[5], T[6], T[7], v, K[12], K[13], K[14], K[15]);
    BN_WIDE_ROW4S(C[0], C[1], C[2], C[3], T[0], T[1], T[2], T[3], t, K[16], K[17], K[18], K[19]);
    BN_WIDE_ROW4S(C[4], C[5], C[6], C[7], T[4], T[5], T[6], T[7], t, K[20], K[21], K[22], K[23]);
    return wfold_finish(C, T);
}
// x + r d mod p in [0, p), for x, d in [0, p) and K = fold_consts(r) behind a wave-uniform pointer
__device__ __forceinline__ Fr fr_fold_const(const Fr& x, const Fr& d, const u32* __restrict__ K) {
    u64 C[8];
    u32 T[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { C[2 * i] = (u32)x.l[i]; C[2 * i + 1] = x.l[i] >> 32; T[2 * i] = 0; T[2 * i + 1] = 0; }
    u32 dl[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { dl[2 * i] = (u32)d.l[i]; dl[2 * i + 1] = (u32)(d.l[i] >> 32); }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        BN_WIDE_ROW4S(C[0], C[1], C[2], C[3], T[0], T[1], T[2], T[3], dl[i], K[8 * i + 0], K[8 * i + 1], K[8 * i + 2], K[8 * i + 3]);
        BN_WIDE_ROW4S(C[4], C[5], C[6], C[7], T[4], T[5], T[6], T[7], dl[i], K[8 * i + 4], K[8 * i + 5], K[8 * i + 6], K[8 * i + 7]);
    }
    return wfold_finish(C, T);
}

__device__ __forceinline__ Fr fr_mul_wide(const Fr& a, const Fr& b) {
    WCol w = wcol_zero();
    wcol_mac(w, a, b);
    return wcol_reduce(w);
}

}  // namespace bn
}  // namespace hg
