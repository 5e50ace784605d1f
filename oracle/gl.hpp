// ORACLE (test infrastructure — never linked into the product library).
//
// Goldilocks base field F = F_p, p = 2^64 - 2^32 + 1, and its quadratic extension
// E = F[X]/(X^2 - 7) ("GoldilocksExt2").
//
// Restates the arithmetic of the third-party crate `goldilocks`
// (github.com/nulltea/goldilocks, branch `to_canonical_repr`, un-pinned in
// /root/reference/Cargo.toml:28,67-68 — source NOT under /root/reference):
//   * canonical representation in [0, p)          (used at transcript.rs:183-189, lasso.rs:655)
//   * Ext2 non-residue 7, bases order [c0, c1]    (as_bases / from_bases, transcript.rs:149-154,191-195)
// The published definition (p, X^2-7) is restated here; every op is checked against
// Python big-int arithmetic in tests/test_oracle_field.py.
#pragma once
#include <cstdint>
#include <cstddef>

namespace orc {

typedef unsigned __int128 u128;
static const uint64_t GL_P = 0xFFFFFFFF00000001ULL;
static const uint64_t GL_EPS = 0xFFFFFFFFULL;  // 2^64 mod p = 2^32 - 1

static inline uint64_t f_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    // a,b < p so a+b < 2p < 2^65; overflow or s >= p => subtract p once
    if (s < a || s >= GL_P) s -= GL_P;
    return s;
}
static inline uint64_t f_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
static inline uint64_t f_neg(uint64_t a) { return a ? GL_P - a : 0; }
static inline uint64_t f_reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hh = hi >> 32, hl = hi & GL_EPS;
    // x = lo + hl*2^64 + hh*2^96 ≡ lo + hl*(2^32-1) - hh   (2^96 ≡ -1)
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;  // borrow: add p  (== subtract 2^32-1 mod 2^64)
    uint64_t t1 = hl * GL_EPS;  // < 2^64
    uint64_t r = t0 + t1;
    if (r < t0) r += GL_EPS;    // carry: subtract p (== add 2^32-1 mod 2^64)
    if (r >= GL_P) r -= GL_P;
    return r;
}
static inline uint64_t f_mul(uint64_t a, uint64_t b) { return f_reduce128((u128)a * b); }
static inline uint64_t f_from_u64(uint64_t x) { return x >= GL_P ? x - GL_P : x; }
static inline uint64_t f_pow(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = f_mul(r, b); b = f_mul(b, b); e >>= 1; }
    return r;
}
static inline uint64_t f_inv(uint64_t a) { return f_pow(a, GL_P - 2); }

struct E {
    uint64_t c0, c1;
};
static inline E e_zero() { return E{0, 0}; }
static inline E e_one() { return E{1, 0}; }
static inline E e_from_f(uint64_t a) { return E{a, 0}; }
static inline bool e_eq(E a, E b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline E e_add(E a, E b) { return E{f_add(a.c0, b.c0), f_add(a.c1, b.c1)}; }
static inline E e_sub(E a, E b) { return E{f_sub(a.c0, b.c0), f_sub(a.c1, b.c1)}; }
static inline E e_neg(E a) { return E{f_neg(a.c0), f_neg(a.c1)}; }
static inline E e_dbl(E a) { return e_add(a, a); }
static inline E e_mul(E a, E b) {
    uint64_t a0b0 = f_mul(a.c0, b.c0), a1b1 = f_mul(a.c1, b.c1);
    uint64_t c0 = f_add(a0b0, f_mul(7, a1b1));
    uint64_t c1 = f_add(f_mul(a.c0, b.c1), f_mul(a.c1, b.c0));
    return E{c0, c1};
}
static inline E e_mul_f(E a, uint64_t b) { return E{f_mul(a.c0, b), f_mul(a.c1, b)}; }
static inline E e_add_f(E a, uint64_t b) { return E{f_add(a.c0, b), a.c1}; }
static inline E e_sub_f(E a, uint64_t b) { return E{f_sub(a.c0, b), a.c1}; }
static inline E e_sqr(E a) { return e_mul(a, a); }
static inline E e_inv(E a) {
    // (c0 + c1 X)^-1 = (c0 - c1 X) / (c0^2 - 7 c1^2)
    uint64_t n = f_sub(f_mul(a.c0, a.c0), f_mul(7, f_mul(a.c1, a.c1)));
    uint64_t ni = f_inv(n);
    return E{f_mul(a.c0, ni), f_mul(f_neg(a.c1), ni)};
}

}  // namespace orc
