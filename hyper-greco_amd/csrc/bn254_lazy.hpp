// Branch-free "loose" arithmetic for bn256::Fr on gfx950 (device only), used by the hot BN254 round kernels.
//
// The canonical operations of bn254_field.hpp compare and conditionally subtract limb by limb: hipcc turns every one of them into
// per-lane branches (the round-3 grand-product round kernel had 311 conditional branches and 468 exec-mask saves around 947
// multiply-adds, and ran at 0.59 of the VALU issue rate). Here a residue is ANY representative in [0, 2p) ("loose"; p < 2^254, so
// 4p < 2^256 still fits the four limbs), differences are formed as y - x + 2p in (0, 4p) and every reduction ends in a
// double-precision quotient estimate that leaves [0, 2p) - no comparison against p, no data-dependent control flow anywhere:
//
//   lz_sub(y, x)        y - x + 2p                 in (0, 4p)   operand of a multiply-accumulate or of a fold only
//   lz_add / lz_subr    a + b / a - b mod p        in [0, 2p)
//   lz_fold(x, d, K)    x + r d (K = fold_consts(r), bn254_wide.hpp)    in [0, 2p)
//   lz_reduce(w)        value(w) R^-1 mod p (Montgomery, on the columns)  in [0, 2p)
//   lz_canon(a)         the canonical representative in [0, p)
//
// Multiply-accumulates (wcol_mac) take any 256-bit operands. Tables written by these kernels hold loose values; their consumers are
// the same kernels, the tail kernels (which normalise on load) and fr_from_mont (correct for any operand below 2^256).
#pragma once
#include "bn254_wide.hpp"

namespace hg {
namespace bn {

constexpr u64 LZ_2P0 = 0x87c3eb27e0000002ULL, LZ_2P1 = 0x5067d090f372e122ULL, LZ_2P2 = 0x70a08b6d0302b0baULL, LZ_2P3 = 0x60c89ce5c2634053ULL;  // 2p

// y - x + 2p for x, y in [0, 2p): in (0, 4p), fits 256 bits
__device__ __forceinline__ Fr lz_sub(const Fr& y, const Fr& x) {
    Fr t, r;
    u128 d = (u128)LZ_2P0 - x.l[0];
    t.l[0] = (u64)d;
    d = (u128)LZ_2P1 - x.l[1] - (u64)((d >> 64) & 1);
    t.l[1] = (u64)d;
    d = (u128)LZ_2P2 - x.l[2] - (u64)((d >> 64) & 1);
    t.l[2] = (u64)d;
    t.l[3] = LZ_2P3 - x.l[3] - (u64)((d >> 64) & 1);
    u128 c = (u128)y.l[0] + t.l[0];
    r.l[0] = (u64)c;
    c = (u128)y.l[1] + t.l[1] + (u64)(c >> 64);
    r.l[1] = (u64)c;
    c = (u128)y.l[2] + t.l[2] + (u64)(c >> 64);
    r.l[2] = (u64)c;
    r.l[3] = y.l[3] + t.l[3] + (u64)(c >> 64);
    return r;
}
// s in [0, 4p) -> s or s - 2p, whichever is in [0, 2p) (selected by the borrow of the subtraction)
__device__ __forceinline__ Fr lz_cond_sub_2p(const Fr& s) {
    Fr t;
    u128 d = (u128)s.l[0] - LZ_2P0;
    t.l[0] = (u64)d;
    d = (u128)s.l[1] - LZ_2P1 - (u64)((d >> 64) & 1);
    t.l[1] = (u64)d;
    d = (u128)s.l[2] - LZ_2P2 - (u64)((d >> 64) & 1);
    t.l[2] = (u64)d;
    d = (u128)s.l[3] - LZ_2P3 - (u64)((d >> 64) & 1);
    t.l[3] = (u64)d;
    const bool neg = (u64)((d >> 64) & 1) != 0;
    return fr_make(neg ? s.l[0] : t.l[0], neg ? s.l[1] : t.l[1], neg ? s.l[2] : t.l[2], neg ? s.l[3] : t.l[3]);
}
__device__ __forceinline__ Fr lz_add(const Fr& a, const Fr& b) {   // a, b in [0, 2p) -> [0, 2p)
    Fr s;
    u128 c = (u128)a.l[0] + b.l[0];
    s.l[0] = (u64)c;
    c = (u128)a.l[1] + b.l[1] + (u64)(c >> 64);
    s.l[1] = (u64)c;
    c = (u128)a.l[2] + b.l[2] + (u64)(c >> 64);
    s.l[2] = (u64)c;
    s.l[3] = a.l[3] + b.l[3] + (u64)(c >> 64);
    return lz_cond_sub_2p(s);
}
__device__ __forceinline__ Fr lz_subr(const Fr& a, const Fr& b) { return lz_cond_sub_2p(lz_sub(a, b)); }   // a - b in [0, 2p)
__device__ __forceinline__ Fr lz_canon(const Fr& a) {   // a in [0, 2p) -> [0, p)
    Fr t;
    u128 d = (u128)a.l[0] - FR_P0;
    t.l[0] = (u64)d;
    d = (u128)a.l[1] - FR_P1 - (u64)((d >> 64) & 1);
    t.l[1] = (u64)d;
    d = (u128)a.l[2] - FR_P2 - (u64)((d >> 64) & 1);
    t.l[2] = (u64)d;
    d = (u128)a.l[3] - FR_P3 - (u64)((d >> 64) & 1);
    t.l[3] = (u64)d;
    const bool neg = (u64)((d >> 64) & 1) != 0;
    return fr_make(neg ? a.l[0] : t.l[0], neg ? a.l[1] : t.l[1], neg ? a.l[2] : t.l[2], neg ? a.l[3] : t.l[3]);
}

// C[k..k+2] += x * y[0..2] with the y in SGPRs, carries banked in T[k..k+2]
#define BN_WIDE_ROW3S(c0, c1, c2, t0, t1, t2, x, y0, y1, y2)                                                                 \
    do {                                                                                                                     \
        u64 s0_, s1_, s2_;                                                                                                   \
        asm("v_mad_u64_u32 %0, %6, %9, %10, %0\n\t"                                                                          \
            "v_mad_u64_u32 %1, %7, %9, %11, %1\n\t"                                                                          \
            "v_mad_u64_u32 %2, %8, %9, %12, %2\n\t"                                                                          \
            "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"                                                                        \
            "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"                                                                        \
            "v_addc_co_u32_e64 %5, %8, 0, %5, %8"                                                                            \
            : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(t0), "+v"(t1), "+v"(t2), "=&s"(s0_), "=&s"(s1_), "=&s"(s2_)                 \
            : "v"(x), "s"(y0), "s"(y1), "s"(y2));                                                                            \
    } while (0)

constexpr u32 LZ_NP0 = 0x0fffffffu, LZ_NP1 = 0xbc1e0a6cu, LZ_NP2 = 0x86468f6eu, LZ_NP3 = 0xd7cc17b7u, LZ_NP4 = 0x7e7ea7a2u, LZ_NP5 = 0x47afba49u,
              LZ_NP6 = 0x1ece5fd6u, LZ_NP7 = 0xcf9bb18du;   // 2^256 - p

// V = e0 + e1 2^32 + sum_k C[k] 2^(32 k) + sum_k T[k] 2^(32 k + 64) (k < 8), V < 2^36 p  ->  V mod p as a loose residue in [0, 2p).
// (The callers stay below ~2^36 p: lz_fold 2p + 8 2^32 p, lz_reduce ~2^10 p, lz_lin3 ~2^34 p. At q ~ 2^40 the five conversions, the
// product with 2^224 / p and that constant's own rounding add up to ~0.85 2^-10 - too close to the 2^-10 margin to promise; at
// 2^36 they are below 2^-13.)
// q = floor(V / p) or one less from a double-precision estimate of V / 2^224 (the neglected low parts are below 2^-27 of a unit of q,
// the conversions below 2^-14 at that bound; 2^-10 is subtracted before rounding down), then V + q (2^256 - p) is formed modulo 2^256 with 15
// multiply-adds into the same columns and normalised into eight 32-bit limbs: what is left is exactly V - q p, which is below 2p.
__device__ __forceinline__ Fr lz_finish(u64* C, u32* T, u64 e0, u32 e1) {
    const double top = (double)C[7] + (double)(u32)(C[6] >> 32) + (double)T[5] + 4294967296.0 * (double)T[6] + 18446744073709551616.0 * (double)T[7];
    double qd = top * 1.2317090423844144e-09 - 0.0009765625;   // 2^224 / p
    qd = qd > 0.0 ? qd : 0.0;
    const u32 qhi = (u32)(qd * 2.3283064365386963e-10);        // 2^-32 (truncation = floor: qd >= 0)
    const u32 qlo = (u32)(qd - (double)qhi * 4294967296.0);
    u64 c8 = 0;   // column 8 and its carry counter: weights 2^256 and above, dropped
    u32 t8 = 0;
    BN_WIDE_ROW4S(C[0], C[1], C[2], C[3], T[0], T[1], T[2], T[3], qlo, LZ_NP0, LZ_NP1, LZ_NP2, LZ_NP3);
    BN_WIDE_ROW4S(C[4], C[5], C[6], C[7], T[4], T[5], T[6], T[7], qlo, LZ_NP4, LZ_NP5, LZ_NP6, LZ_NP7);
    BN_WIDE_ROW4S(C[1], C[2], C[3], C[4], T[1], T[2], T[3], T[4], qhi, LZ_NP0, LZ_NP1, LZ_NP2, LZ_NP3);
    BN_WIDE_ROW4S(C[5], C[6], C[7], c8, T[5], T[6], T[7], t8, qhi, LZ_NP4, LZ_NP5, LZ_NP6, LZ_NP7);
    u32 L[8];
    u64 s = e0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        s += (u64)(u32)C[k];
        if (k == 1) s += (u64)e1;
        if (k >= 1) s += C[k - 1] >> 32;
        if (k >= 2) s += (u64)T[k - 2];
        L[k] = (u32)s;
        s >>= 32;
    }
    return fr_make((u64)L[0] | ((u64)L[1] << 32), (u64)L[2] | ((u64)L[3] << 32), (u64)L[4] | ((u64)L[5] << 32), (u64)L[6] | ((u64)L[7] << 32));
}

// 32-byte table entries through GLOBAL loads / stores: pointers taken from a job descriptor in memory are generic to the compiler,
// which then emits flat_load (counted by lgkmcnt as well, so every scalar-load wait would also wait for the table loads)
typedef u32 lz_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Fr lz_gload(const Fr* p) {
    const __attribute__((address_space(1))) lz_u32x4* g = (const __attribute__((address_space(1))) lz_u32x4*)p;
    const lz_u32x4 a = g[0], b = g[1];
    return fr_make((u64)a.x | ((u64)a.y << 32), (u64)a.z | ((u64)a.w << 32), (u64)b.x | ((u64)b.y << 32), (u64)b.z | ((u64)b.w << 32));
}
__device__ __forceinline__ void lz_gstore(Fr* p, const Fr& v) {
    __attribute__((address_space(1))) lz_u32x4* g = (__attribute__((address_space(1))) lz_u32x4*)p;
    lz_u32x4 a, b;
    a.x = (u32)v.l[0]; a.y = (u32)(v.l[0] >> 32); a.z = (u32)v.l[1]; a.w = (u32)(v.l[1] >> 32);
    b.x = (u32)v.l[2]; b.y = (u32)(v.l[2] >> 32); b.z = (u32)v.l[3]; b.w = (u32)(v.l[3] >> 32);
    g[0] = a; g[1] = b;
}
// the same as a streaming store: tables far larger than the caches that nothing re-reads before the next launch (the folded tables of the
// big rounds, the tree levels, the hash rows)
__device__ __forceinline__ void lz_gstore_nt(Fr* p, const Fr& v) {
    __attribute__((address_space(1))) lz_u32x4* g = (__attribute__((address_space(1))) lz_u32x4*)p;
    lz_u32x4 a, b;
    a.x = (u32)v.l[0]; a.y = (u32)(v.l[0] >> 32); a.z = (u32)v.l[1]; a.w = (u32)(v.l[1] >> 32);
    b.x = (u32)v.l[2]; b.y = (u32)(v.l[2] >> 32); b.z = (u32)v.l[3]; b.w = (u32)(v.l[3] >> 32);
    __builtin_nontemporal_store(a, g);
    __builtin_nontemporal_store(b, g + 1);
}
// LDS-DMA prefetch (gfx950 global_load_lds_dwordx4): 16 bytes per lane from a per-lane global address straight into LDS at
// (wave-uniform byte address) + lane * 16, no VGPR destination - the way to have the NEXT work item's table entries in flight
// while a kernel that already fills its 256 VGPRs with column accumulators computes on the current one. The instruction is inline
// assembly: hipcc does not count it, the caller waits with lz_wait_vm0() before reading the staged bytes (a wait for fewer
// outstanding operations than there really are only waits longer, never too little: the counter retires in order).
__device__ __forceinline__ void lz_glds16(const void* gsrc, u32 lds_dst) {
    u32 keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);   // (inside a divergent region hipcc may hold a uniform value in a VGPR)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void lz_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lz_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ Fr lz_from_x4(const lz_u32x4& a, const lz_u32x4& b) {
    return fr_make((u64)a.x | ((u64)a.y << 32), (u64)a.z | ((u64)a.w << 32), (u64)b.x | ((u64)b.y << 32), (u64)b.z | ((u64)b.w << 32));
}

// the 64 limbs of fold_consts(r) held in SGPRs for a whole kernel (wave-uniform values: scalar loads, hoisted out of every loop)
struct LzK { u32 k[64]; };
__device__ __forceinline__ LzK lz_load_k(const u32* __restrict__ K) {
    LzK r;
#pragma unroll
    for (int i = 0; i < 64; i++) r.k[i] = __builtin_amdgcn_readfirstlane(K[i]);
    return r;
}

// x + r d in [0, 2p) for x in [0, 2p), d any 256-bit value, K = fold_consts(r) behind a wave-uniform pointer
__device__ __forceinline__ Fr lz_fold(const Fr& x, const Fr& d, const u32* __restrict__ K) {
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
    return lz_finish(C, T, 0, 0);
}

// value(w) R^-1 mod p in [0, 2p) for value(w) < 2^12 p^2: Montgomery reduction on the columns (limb i made exact, m = limb * (-p^-1)
// mod 2^32, m p added with multiply-add / carry pairs), then the quotient-estimate finish on columns 8 .. 15
__device__ __forceinline__ Fr lz_reduce(WCol& w) {
    constexpr u32 P0 = (u32)FR_P0, P1 = (u32)(FR_P0 >> 32), P2 = (u32)FR_P1, P3 = (u32)(FR_P1 >> 32), P4 = (u32)FR_P2, P5 = (u32)(FR_P2 >> 32),
                  P6 = (u32)FR_P3, P7 = (u32)(FR_P3 >> 32);
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 s = carry + (u64)(u32)w.C[i];
        if (i >= 1) s += w.C[i - 1] >> 32;
        if (i >= 2) s += (u64)w.T[i - 2];
        const u32 limb = (u32)s;
        const u32 m = limb * FR_INV32;
        carry = (s >> 32) + (((u64)m * P0 + limb) >> 32);   // limb + m p_0 = 0 mod 2^32
        BN_WIDE_ROW4S(w.C[i + 1], w.C[i + 2], w.C[i + 3], w.C[i + 4], w.T[i + 1], w.T[i + 2], w.T[i + 3], w.T[i + 4], m, P1, P2, P3, P4);
        BN_WIDE_ROW3S(w.C[i + 5], w.C[i + 6], w.C[i + 7], w.T[i + 5], w.T[i + 6], w.T[i + 7], m, P5, P6, P7);
    }
    // what is left, divided by 2^256: columns 8 .. 15 with their carry counters, plus the tail of columns 6, 7 and the running carry
    return lz_finish(&w.C[8], &w.T[8], carry + (w.C[7] >> 32) + (u64)w.T[6], w.T[7]);
}
// a KA + v KV + t KT + add for 32-bit integers a, v, t (constants as fr_lin3_const, bn254_wide.hpp), loose result
__device__ __forceinline__ Fr lz_lin3(u32 a, u32 v, u32 t, const u32* __restrict__ K) {
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
    return lz_finish(C, T, 0, 0);
}
__device__ __forceinline__ Fr lz_mul(const Fr& a, const Fr& b) {
    WCol w = wcol_zero();
    wcol_mac(w, a, b);
    return lz_reduce(w);
}

}  // namespace bn
}  // namespace hg
