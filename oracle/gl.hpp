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
//
// This header is one of the two field back ends of the oracle (the other is fr.hpp: bn256::Fr with E = F); the protocol
// headers are written against the names both define: types F / E, f_* / e_* arithmetic, wire format, challenge derivation,
// limb load / store for the C surface. field.hpp picks one per translation unit.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>

#define ORC_NS orc
#define ORC_SYM(name) orc_##name
#define ORC_F_IS_U64 1

namespace orc {

typedef unsigned __int128 u128;
typedef uint64_t F;                  // base-field element, canonical value in [0, p)
static const size_t F_LIMBS = 1;     // u64 limbs per F / E at the C surface
static const size_t E_LIMBS = 2;
static const size_t F_BYTES = 8;     // <F as PrimeField>::Repr length
static const size_t E_DEGREE = 2;    // ExtensionField::DEGREE
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

static inline F f_zero() { return 0; }
static inline F f_one() { return 1; }
static inline bool f_eq(F a, F b) { return a == b; }
static inline bool f_is_zero(F a) { return a == 0; }
static inline F f_dbl(F a) { return f_add(a, a); }
// witness coefficients arrive as Goldilocks residues (negatives as p - |z|, utils.py:4-18): already field elements here
static inline F f_from_signed_gl(uint64_t v) { return v; }
static inline uint64_t f_low_u64(F a) { return a; }  // inverse of f_from_u64 for values below 2^63 (range-shifted lookups)

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

// ExtensionField::{from_bases, as_bases} (transcript.rs:149-154, 191-195): [c0, c1]
static inline E e_from_bases(const F* b) { return E{b[0], b[1]}; }
static inline void e_as_bases(E a, F* b) { b[0] = a.c0; b[1] = a.c1; }
static inline F e_limb0(E a) { return a.c0; }  // `as_bases()[0]` (prover.rs:38-39)
static inline bool e_is_zero(E a) { return a.c0 == 0 && a.c1 == 0; }

// wire format (transcript.rs:183-189, 162-170): canonical repr byte-reversed to big-endian; non-canonical input rejected
static inline void f_write_be(F a, uint8_t* out) { for (int i = 0; i < 8; i++) out[i] = (uint8_t)(a >> (8 * (7 - i))); }
static inline bool f_read_be(const uint8_t* in, F& out) {
    uint64_t a = 0;
    for (int i = 0; i < 8; i++) a = (a << 8) | in[i];
    out = a;
    return a < GL_P;
}
// PrimeField::to_repr: canonical little-endian bytes (what an absorbing transcript hashes, transcript.rs:205-208, 224-226)
static inline void f_repr_le(F a, uint8_t* out) { memcpy(out, &a, 8); }
// plonkish fe_mod_from_le_bytes (call site transcript.rs:202): 256-bit little-endian integer reduced mod p
static inline F f_from_hash_le(const uint8_t h[32]) {
    uint64_t l[4];
    memcpy(l, h, 32);
    uint64_t r = f_from_u64(l[3]);
    for (int i = 2; i >= 0; i--) r = f_add(f_mul(r, GL_EPS), f_from_u64(l[i]));
    return r;
}
// ff::PrimeField::ROOT_OF_UNITY for Goldilocks (S = 32, multiplicative generator 7), squared down to order 2^log2n
static inline F f_root_of_unity(size_t log2n) {
    uint64_t w = f_pow(7, (GL_P - 1) >> 32);
    for (size_t i = log2n; i < 32; i++) w = f_mul(w, w);
    return w;
}
// canonical u64 limbs at the C surface
static inline F f_load(const uint64_t* p) { return p[0]; }
static inline void f_store(F a, uint64_t* p) { p[0] = a; }
static inline E e_load(const uint64_t* p) { return E{p[0], p[1]}; }
static inline void e_store(E a, uint64_t* p) { p[0] = a.c0; p[1] = a.c1; }

}  // namespace orc
