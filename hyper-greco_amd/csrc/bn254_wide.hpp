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

__device__ __forceinline__ Fr fr_mul_wide(const Fr& a, const Fr& b) {
    WCol w = wcol_zero();
    wcol_mac(w, a, b);
    return wcol_reduce(w);
}

}  // namespace bn
}  // namespace hg
