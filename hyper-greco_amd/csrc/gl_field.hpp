// Goldilocks F = F_p (p = 2^64 - 2^32 + 1) and GoldilocksExt2 E = F[X]/(X^2 - 7) in 64-bit lanes,
// shared by the host driver and the gfx950 kernels. Values are always canonical (< p): the proof
// stream is the canonical big-endian repr [REF bfv-gkr/src/transcript.rs:183-195], and exact
// canonical arithmetic makes every reduction order give bit-identical results.
// Replaces the reference's external field crates (goldilocks::{Goldilocks, GoldilocksExt2},
// [REF Cargo.toml:28,67-68; uses at bfv-gkr/src/sk_encryption_circuit.rs:539-540]).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define HG_HD __host__ __device__ __forceinline__

namespace hg {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 GL_P = 0xFFFFFFFF00000001ULL;
constexpr u64 GL_EPS = 0xFFFFFFFFULL;

HG_HD u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    u64 c = (s < a) | (s >= GL_P);
    return s - (c ? GL_P : 0);
}
HG_HD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    return d + ((a < b) ? GL_P : 0);
}
HG_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
HG_HD u64 gl_dbl(u64 a) { return gl_add(a, a); }

HG_HD u64 gl_reduce128(u64 lo, u64 hi) {
    // hi*2^64 + lo,  2^64 = 2^32 - 1,  2^96 = -1  (mod p)
    u64 hh = hi >> 32, hl = hi & GL_EPS;
    u64 t0 = lo - hh;
    t0 -= (lo < hh) ? GL_EPS : 0;
    u64 t1 = (hl << 32) - hl;
    u64 r = t0 + t1;
    r += (r < t0) ? GL_EPS : 0;
    r -= (r >= GL_P) ? GL_P : 0;
    return r;
}
HG_HD u64 gl_mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // four 32x32+64 products (v_mad_u64_u32): measured 1.63 T mul/s on MI355X vs 1.34 for a*b + __umul64hi
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 mid = (u64)a0 * b1 + (p00 >> 32);
    u64 mid2 = (u64)a1 * b0 + (u32)mid;
    u64 hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
    u64 lo = (mid2 << 32) | (u32)p00;
    return gl_reduce128(lo, hi);
#else
    unsigned __int128 x = (unsigned __int128)a * b;
    return gl_reduce128((u64)x, (u64)(x >> 64));
#endif
}
// a * small constant (c < 2^32): the high word is < 2^32, so the reduction has no 2^96 term
HG_HD u64 gl_mul_small(u64 a, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 lo = a * (u64)c, hi = __umul64hi(a, (u64)c);
#else
    unsigned __int128 x = (unsigned __int128)a * c;
    u64 lo = (u64)x, hi = (u64)(x >> 64);
#endif
    u64 t1 = (hi << 32) - hi;
    u64 r = lo + t1;
    r += (r < lo) ? GL_EPS : 0;
    r -= (r >= GL_P) ? GL_P : 0;
    return r;
}
// 7 * a as SOME 64-bit residue (not canonical): for operands of the deferred-reduction multiply-accumulates, which accept any
// 64-bit residue. 7a = lo + hi 2^64 with hi < 7, 2^64 = eps: lo + hi eps, plus eps once more if that sum wraps.
HG_HD u64 gl_mul7_lazy(u64 a) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 lo = a * 7ULL, hi = __umul64hi(a, 7ULL);
#else
    unsigned __int128 x = (unsigned __int128)a * 7;
    u64 lo = (u64)x, hi = (u64)(x >> 64);
#endif
    u64 r = lo + ((hi << 32) - hi);
    r += (r < lo) ? GL_EPS : 0;
    return r;
}
HG_HD u64 gl_from_u64(u64 x) { return x >= GL_P ? x - GL_P : x; }

inline u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) { if (e & 1) r = gl_mul(r, b); b = gl_mul(b, b); e >>= 1; }
    return r;
}
inline u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }

struct __attribute__((aligned(16))) E2 {
    u64 c0, c1;
};

HG_HD E2 e2(u64 a, u64 b) { E2 r; r.c0 = a; r.c1 = b; return r; }
HG_HD E2 e2_zero() { return e2(0, 0); }
HG_HD E2 e2_one() { return e2(1, 0); }
HG_HD E2 e2_add(E2 a, E2 b) { return e2(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
HG_HD E2 e2_sub(E2 a, E2 b) { return e2(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
HG_HD E2 e2_dbl(E2 a) { return e2(gl_dbl(a.c0), gl_dbl(a.c1)); }
HG_HD E2 e2_add_f(E2 a, u64 b) { return e2(gl_add(a.c0, b), a.c1); }
HG_HD E2 e2_sub_f(E2 a, u64 b) { return e2(gl_sub(a.c0, b), a.c1); }
HG_HD E2 e2_mul_f(E2 a, u64 b) { return e2(gl_mul(a.c0, b), gl_mul(a.c1, b)); }
HG_HD E2 e2_mul(E2 a, E2 b) {
    // Karatsuba over X^2 = 7: 3 base multiplications + one by the constant 7
    u64 p0 = gl_mul(a.c0, b.c0), p1 = gl_mul(a.c1, b.c1);
    u64 m = gl_mul(gl_add(a.c0, a.c1), gl_add(b.c0, b.c1));
    return e2(gl_add(p0, gl_mul_small(p1, 7)), gl_sub(gl_sub(m, p0), p1));
}
HG_HD bool e2_eq(E2 a, E2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
inline E2 e2_inv(E2 a) {
    u64 n = gl_sub(gl_mul(a.c0, a.c0), gl_mul_small(gl_mul(a.c1, a.c1), 7));
    u64 ni = gl_inv(n);
    return e2(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}

}  // namespace hg
