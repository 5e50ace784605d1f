// Deferred-reduction Goldilocks arithmetic for the gfx950 sum-check kernels (device only).
//
// Measured on MI355X (scripts/ub/ratebench.hip): v_mad_u64_u32 issues at the same rate as any other VOP3
// instruction, so a 64x64 product is cheap; what the reduce-every-product kernels spent their time on was the
// carry / compare / select logic of gl_reduce128 and of the canonical adds around it (about 19 + 7 instructions
// per product). The sums of a sum-check round are long dot products (one term per table pair), so the kernels
// keep them UNREDUCED in "column" accumulators and reduce once per hypercube point:
//
//   value(WAcc) = L + M 2^32 + H 2^64 + tL 2^64 + tM 2^96 + tH 2^128
//
// L, M, H are 64-bit accumulators of the 32x32 partial products of weight 2^0, 2^32, 2^64; tL, tM, tH count their
// carry-outs (v_mad_u64_u32 delivers the carry in an SGPR pair, one v_addc_co_u32 banks it). One 64x64
// multiply-accumulate is therefore 4 v_mad_u64_u32 + 4 v_addc_co_u32 with no moves, shifts or compares.
// Results are exact integers, reduced to the canonical representative at the end, so the proof bytes are the
// same as with gl_mul [REF field semantics: goldilocks crate, Cargo.toml:28,67-68].
//
// gfx950 hazard: a VALU instruction that reads an SGPR written by a VALU instruction needs two other
// instructions (or wait states) in between; the compiler cannot see inside an asm block, so every block below
// keeps that distance itself.
#pragma once
#include "gl_field.hpp"

namespace hg {

struct WAcc {
    u64 L, M, H;
    u32 tL, tM, tH;
};
__device__ __forceinline__ WAcc wacc_zero() { WAcc w; w.L = w.M = w.H = 0; w.tL = w.tM = w.tH = 0; return w; }

// w += a * b   (a, b any 64-bit residues)
__device__ __forceinline__ void wmac(WAcc& w, u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 c0, c1, c2, c3;
    asm("v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
        "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"
        "v_mad_u64_u32 %2, %8, %11, %13, %2\n\t"
        "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
        "v_mad_u64_u32 %1, %9, %11, %12, %1\n\t"
        "v_addc_co_u32_e64 %5, %8, 0, %5, %8\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %4, %9, 0, %4, %9"
        : "+v"(w.L), "+v"(w.M), "+v"(w.H), "+v"(w.tL), "+v"(w.tM), "+v"(w.tH), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}

// two independent multiply-accumulates in one block (no wait state needed): w += a * b, z += c * d
__device__ __forceinline__ void wmac2(WAcc& w, u64 a, u64 b, WAcc& z, u64 c, u64 d) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u32 e0 = (u32)c, e1 = (u32)(c >> 32), f0 = (u32)d, f1 = (u32)(d >> 32);
    u64 c0, c1, c2, c3;
    asm("v_mad_u64_u32 %0, %12, %16, %18, %0\n\t"   // w.L += a0 b0
        "v_mad_u64_u32 %1, %13, %16, %19, %1\n\t"   // w.M += a0 b1
        "v_mad_u64_u32 %2, %14, %17, %19, %2\n\t"   // w.H += a1 b1
        "v_mad_u64_u32 %6, %15, %20, %22, %6\n\t"   // z.L += e0 f0
        "v_addc_co_u32_e64 %3, %12, 0, %3, %12\n\t"
        "v_addc_co_u32_e64 %4, %13, 0, %4, %13\n\t"
        "v_addc_co_u32_e64 %5, %14, 0, %5, %14\n\t"
        "v_addc_co_u32_e64 %9, %15, 0, %9, %15\n\t"
        "v_mad_u64_u32 %1, %12, %17, %18, %1\n\t"   // w.M += a1 b0
        "v_mad_u64_u32 %7, %13, %20, %23, %7\n\t"   // z.M += e0 f1
        "v_mad_u64_u32 %8, %14, %21, %23, %8\n\t"   // z.H += e1 f1
        "v_addc_co_u32_e64 %4, %12, 0, %4, %12\n\t"
        "v_addc_co_u32_e64 %10, %13, 0, %10, %13\n\t"
        "v_mad_u64_u32 %7, %15, %21, %22, %7\n\t"   // z.M += e1 f0
        "v_addc_co_u32_e64 %11, %14, 0, %11, %14\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %10, %15, 0, %10, %15"
        : "+v"(w.L), "+v"(w.M), "+v"(w.H), "+v"(w.tL), "+v"(w.tM), "+v"(w.tH),
          "+v"(z.L), "+v"(z.M), "+v"(z.H), "+v"(z.tL), "+v"(z.tM), "+v"(z.tH),
          "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(e0), "v"(e1), "v"(f0), "v"(f1));
}

// w += a * b + c * d  (both products into the same accumulator)
__device__ __forceinline__ void wmac_pair(WAcc& w, u64 a, u64 b, u64 c, u64 d) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u32 e0 = (u32)c, e1 = (u32)(c >> 32), f0 = (u32)d, f1 = (u32)(d >> 32);
    u64 c0, c1, c2, c3;
    asm("v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"    // L += a0 b0
        "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"    // M += a0 b1
        "v_mad_u64_u32 %2, %8, %11, %13, %2\n\t"    // H += a1 b1
        "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
        "v_mad_u64_u32 %0, %6, %14, %16, %0\n\t"    // L += e0 f0
        "v_addc_co_u32_e64 %5, %8, 0, %5, %8\n\t"
        "v_mad_u64_u32 %1, %7, %11, %12, %1\n\t"    // M += a1 b0
        "v_mad_u64_u32 %2, %8, %15, %17, %2\n\t"    // H += e1 f1
        "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
        "v_mad_u64_u32 %1, %7, %14, %17, %1\n\t"    // M += e0 f1
        "v_addc_co_u32_e64 %5, %8, 0, %5, %8\n\t"
        "v_mad_u64_u32 %1, %9, %15, %16, %1\n\t"    // M += e1 f0
        "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %4, %9, 0, %4, %9"
        : "+v"(w.L), "+v"(w.M), "+v"(w.H), "+v"(w.tL), "+v"(w.tM), "+v"(w.tH), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(e0), "v"(e1), "v"(f0), "v"(f1));
}

// The same accumulations STARTING an accumulator (the per-item folds x + r d: one or two products on top of x, reduced at once): the
// first product of a column has nothing to add to and cannot carry, so the nine zeroing moves of wacc_zero() and two carry banks go.
// w = x + a * b + c * d
__device__ __forceinline__ WAcc wacc_pair_init(u64 x, u64 a, u64 b, u64 c, u64 d) {
    WAcc w;
    w.L = x;
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u32 e0 = (u32)c, e1 = (u32)(c >> 32), f0 = (u32)d, f1 = (u32)(d >> 32);
    u64 c0, c1, c2, c3;
    asm("v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"    // L += a0 b0
        "v_mad_u64_u32 %1, %7, %10, %13, 0\n\t"     // M  = a0 b1
        "v_mad_u64_u32 %2, %8, %11, %13, 0\n\t"     // H  = a1 b1
        "v_addc_co_u32_e64 %3, %6, 0, 0, %6\n\t"    // tL = carry
        "v_mad_u64_u32 %0, %6, %14, %16, %0\n\t"    // L += e0 f0
        "v_mad_u64_u32 %1, %7, %11, %12, %1\n\t"    // M += a1 b0
        "v_mad_u64_u32 %2, %8, %15, %17, %2\n\t"    // H += e1 f1
        "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, %7, 0, 0, %7\n\t"    // tM = carry
        "v_mad_u64_u32 %1, %7, %14, %17, %1\n\t"    // M += e0 f1
        "v_addc_co_u32_e64 %5, %8, 0, 0, %8\n\t"    // tH = carry
        "v_mad_u64_u32 %1, %9, %15, %16, %1\n\t"    // M += e1 f0
        "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %4, %9, 0, %4, %9"
        : "+v"(w.L), "=&v"(w.M), "=&v"(w.H), "=&v"(w.tL), "=&v"(w.tM), "=&v"(w.tH), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(e0), "v"(e1), "v"(f0), "v"(f1));
    return w;
}
// w = x + a * b
__device__ __forceinline__ WAcc wacc_mul_init(u64 x, u64 a, u64 b) {
    WAcc w;
    w.L = x;
    w.tH = 0;
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 c0, c1, c2;
    asm("v_mad_u64_u32 %0, %5, %8, %10, %0\n\t"     // L += a0 b0
        "v_mad_u64_u32 %1, %6, %8, %11, 0\n\t"      // M  = a0 b1
        "v_mad_u64_u32 %2, %7, %9, %11, 0\n\t"      // H  = a1 b1
        "v_addc_co_u32_e64 %3, %5, 0, 0, %5\n\t"    // tL = carry
        "v_mad_u64_u32 %1, %6, %9, %10, %1\n\t"     // M += a1 b0
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %4, %6, 0, 0, %6"          // tM = carry
        : "+v"(w.L), "=&v"(w.M), "=&v"(w.H), "=&v"(w.tL), "=&v"(w.tM), "=&s"(c0), "=&s"(c1), "=&s"(c2)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
    return w;
}

// lo + hi * 2^64 with hi < 2^32  ->  canonical
__device__ __forceinline__ u64 gl_reduce96(u64 lo, u32 hi) {
    u64 t1 = ((u64)hi << 32) - hi;  // hi * (2^32 - 1) < 2^64
    u64 r = lo + t1;
    r += (r < t1) ? GL_EPS : 0;
    r -= (r >= GL_P) ? GL_P : 0;
    return r;
}

// canonical value of a column accumulator
__device__ __forceinline__ u64 wreduce(const WAcc& w) {
    // 2^64 = eps, 2^96 = -1, 2^128 = -2^32 (mod p):
    //   value = (L + M0 2^32) + 2^64 (M1 + H0 + tL) - (H1 + tM) - 2^32 tH
    //         = A + eps S0 - K,   A = (L + M0 2^32) mod 2^64, S = M1 + H0 + tL + carry(A), S0 = S mod 2^32,
    //                             K = (S >> 32) + H1 + tM + 2^32 tH   (S 2^64 = S0 eps - (S >> 32))
    u32 L1 = (u32)(w.L >> 32), M0 = (u32)w.M, M1 = (u32)(w.M >> 32), H0 = (u32)w.H, H1 = (u32)(w.H >> 32);
    u32 A1 = L1 + M0;
    u64 S = (u64)M1 + H0 + w.tL + (A1 < M0 ? 1u : 0u);  // < 2^34
    u64 A = ((u64)A1 << 32) | (u32)w.L;
    u64 K = (S >> 32) + H1 + w.tM + ((u64)w.tH << 32);  // < 2^41 while the counters stay below 2^8
    u64 t = A + (u64)(u32)S * 0xFFFFFFFFu;
    t += (t < A) ? GL_EPS : 0;      // wrapped past 2^64 = eps; cannot wrap again
    u64 u = t - K;
    u -= (t < K) ? GL_EPS : 0;      // borrowed 2^64 = eps; the wrapped value is within 2^41 of 2^64
    u -= (u >= GL_P) ? GL_P : 0;
    return u;
}

// three column accumulators hold an unreduced Ext2 dot product: A = sum a0 b0, B = sum a1 b1, C = sum a0 b1 + a1 b0
struct WE2 {
    WAcc A, B, C;
};
__device__ __forceinline__ WE2 we2_zero() { WE2 s; s.A = wacc_zero(); s.B = wacc_zero(); s.C = wacc_zero(); return s; }
__device__ __forceinline__ void we2_mac(WE2& s, E2 a, E2 b) {
    wmac2(s.A, a.c0, b.c0, s.B, a.c1, b.c1);
    wmac_pair(s.C, a.c0, b.c1, a.c1, b.c0);
}
__device__ __forceinline__ E2 we2_reduce(const WE2& s) {
    return e2(gl_add(wreduce(s.A), gl_mul_small(wreduce(s.B), 7)), wreduce(s.C));
}

// x + r * d over Ext2 for a loop-invariant r: c0 = x0 + r0 d0 + (7 r1) d1, c1 = x1 + r0 d1 + r1 d0, each one
// column accumulator (two products on top of x) and one reduction.
struct FoldR {
    u64 r0, r1, r17;
};
__device__ __forceinline__ FoldR fold_r(E2 r) { FoldR f; f.r0 = r.c0; f.r1 = r.c1; f.r17 = gl_mul_small(r.c1, 7); return f; }
__device__ __forceinline__ E2 e2_fold_wide(E2 x, E2 d, const FoldR& f) {
    const WAcc a = wacc_pair_init(x.c0, f.r0, d.c0, f.r17, d.c1);
    const WAcc b = wacc_pair_init(x.c1, f.r0, d.c1, f.r1, d.c0);
    return e2(wreduce(a), wreduce(b));
}

}  // namespace hg
