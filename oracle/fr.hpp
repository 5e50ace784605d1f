// ORACLE (test infrastructure — never linked into the product library).
//
// bn256::Fr, the scalar field of BN254 (halo2curves = "0.7.0", /root/reference/Cargo.toml:29; source NOT under
// /root/reference), as the reference's bn254 test family uses it: F = E = Fr, DEGREE = 1
// [REF bfv-gkr/src/sk_encryption_circuit.rs:539-540 `type F = ...`, :614-626 generate_sk_enc_test!(... Fr, Fr ...)].
//   r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
//   to_repr = 32 little-endian bytes of the canonical value; S = 28, multiplicative generator 7,
//   ROOT_OF_UNITY = 7^((r-1)/2^28) = 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
// The published definition is restated: Montgomery form over 4 x 64-bit limbs with unsigned __int128 products (operand
// scanning, reduction interleaved per limb). Written independently of the product's device arithmetic
// (hyper-greco_amd/csrc/bn254*.hpp: 32-bit-limb column accumulators). Every op is checked against Python integers in
// tests/test_oracle_kats.py; the Python oracle (oracle/bn254.py) cross-checks the protocol level at n = 1024.
//
// Second field back end of the oracle (see gl.hpp / field.hpp): same names, E = F.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>

#define ORC_NS orcbn
#define ORC_SYM(name) orcbn_##name
#define ORC_F_IS_U64 0

namespace orcbn {

typedef unsigned __int128 u128;

struct Fr {
    uint64_t l[4];  // Montgomery form: value * 2^256 mod r, fully reduced
};
typedef Fr F;
typedef Fr E;
static const size_t F_LIMBS = 4;
static const size_t E_LIMBS = 4;
static const size_t F_BYTES = 32;
static const size_t E_DEGREE = 1;

static const uint64_t FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t FR_R[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};   // 2^256 mod r
static const uint64_t FR_R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};  // 2^512 mod r
static const uint64_t FR_INV = 0xc2e1f593efffffffULL;  // -r^-1 mod 2^64

static inline bool fr_geq_mod(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > FR_MOD[i]) return true;
        if (a[i] < FR_MOD[i]) return false;
    }
    return true;
}
static inline void fr_sub_mod(uint64_t a[4]) {
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - FR_MOD[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}
static inline Fr f_add(Fr a, Fr b) {
    Fr r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    // a, b < r < 2^254, so no carry out of 256 bits
    if (fr_geq_mod(r.l)) fr_sub_mod(r.l);
    return r;
}
static inline Fr f_sub(Fr a, Fr b) {
    Fr r;
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + FR_MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    }
    return r;
}
static inline Fr f_zero() { return Fr{{0, 0, 0, 0}}; }
static inline bool f_is_zero(Fr a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline bool f_eq(Fr a, Fr b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
static inline Fr f_neg(Fr a) { return f_is_zero(a) ? a : f_sub(f_zero(), a); }
static inline Fr f_dbl(Fr a) { return f_add(a, a); }

// Montgomery product a * b * 2^-256 mod r
static inline Fr fr_mont_mul(const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * FR_INV;
        c = (u128)m * FR_MOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * FR_MOD[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || fr_geq_mod(r.l)) fr_sub_mod(r.l);
    return r;
}
static inline Fr f_mul(Fr a, Fr b) { return fr_mont_mul(a.l, b.l); }
static inline Fr f_one() { return Fr{{FR_R[0], FR_R[1], FR_R[2], FR_R[3]}}; }
// canonical limbs (value < r assumed) <-> Montgomery form
static inline Fr fr_from_canonical(const uint64_t v[4]) { return fr_mont_mul(v, FR_R2); }
static inline void fr_to_canonical(Fr a, uint64_t out[4]) {
    const uint64_t one[4] = {1, 0, 0, 0};
    Fr c = fr_mont_mul(a.l, one);
    memcpy(out, c.l, 32);
}
static inline Fr f_from_u64(uint64_t x) { const uint64_t v[4] = {x, 0, 0, 0}; return fr_from_canonical(v); }
static inline Fr f_pow(Fr b, uint64_t e) {
    Fr r = f_one();
    while (e) { if (e & 1) r = f_mul(r, b); b = f_mul(b, b); e >>= 1; }
    return r;
}
static inline Fr fr_pow_limbs(Fr b, const uint64_t e[4]) {
    Fr r = f_one();
    for (int i = 3; i >= 0; i--)
        for (int bit = 63; bit >= 0; bit--) {
            r = f_mul(r, r);
            if ((e[i] >> bit) & 1) r = f_mul(r, b);
        }
    return r;
}
static inline Fr f_inv(Fr a) {  // a^(r-2)
    uint64_t e[4] = {FR_MOD[0] - 2, FR_MOD[1], FR_MOD[2], FR_MOD[3]};
    return fr_pow_limbs(a, e);
}
// Witness coefficients arrive as Goldilocks residues of small signed integers (negatives as p_GL - |z|, utils.py:4-18);
// over Fr the same integer z is r - |z| (the reference's bn254 fixtures store exactly that).
static inline Fr f_from_signed_gl(uint64_t v) {
    const uint64_t GLP = 0xFFFFFFFF00000001ULL;
    return v > GLP / 2 ? f_neg(f_from_u64(GLP - v)) : f_from_u64(v);
}
static inline uint64_t f_low_u64(Fr a) { uint64_t c[4]; fr_to_canonical(a, c); return c[0]; }

// ---- E = F -------------------------------------------------------------------------------------
static inline E e_zero() { return f_zero(); }
static inline E e_one() { return f_one(); }
static inline E e_from_f(F a) { return a; }
static inline bool e_eq(E a, E b) { return f_eq(a, b); }
static inline bool e_is_zero(E a) { return f_is_zero(a); }
static inline E e_add(E a, E b) { return f_add(a, b); }
static inline E e_sub(E a, E b) { return f_sub(a, b); }
static inline E e_neg(E a) { return f_neg(a); }
static inline E e_dbl(E a) { return f_add(a, a); }
static inline E e_mul(E a, E b) { return f_mul(a, b); }
static inline E e_mul_f(E a, F b) { return f_mul(a, b); }
static inline E e_add_f(E a, F b) { return f_add(a, b); }
static inline E e_sub_f(E a, F b) { return f_sub(a, b); }
static inline E e_sqr(E a) { return f_mul(a, a); }
static inline E e_inv(E a) { return f_inv(a); }
static inline E e_from_bases(const F* b) { return b[0]; }
static inline void e_as_bases(E a, F* b) { b[0] = a; }
static inline F e_limb0(E a) { return a; }

// wire format (transcript.rs:183-189, 162-170): 32-byte canonical repr byte-reversed to big-endian
static inline void f_write_be(F a, uint8_t* out) {
    uint64_t c[4];
    fr_to_canonical(a, c);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(c[3 - i / 8] >> (8 * (7 - i % 8)));
}
static inline bool f_read_be(const uint8_t* in, F& out) {
    uint64_t c[4] = {0, 0, 0, 0};
    for (int i = 0; i < 32; i++) c[3 - i / 8] = (c[3 - i / 8] << 8) | in[i];
    if (fr_geq_mod(c)) { out = f_zero(); return false; }
    out = fr_from_canonical(c);
    return true;
}
static inline void f_repr_le(F a, uint8_t* out) { uint64_t c[4]; fr_to_canonical(a, c); memcpy(out, c, 32); }
// plonkish fe_mod_from_le_bytes (call site transcript.rs:202): 256-bit little-endian integer reduced mod r (2^256 < 6 r)
static inline F f_from_hash_le(const uint8_t h[32]) {
    uint64_t c[4];
    memcpy(c, h, 32);
    while (fr_geq_mod(c)) fr_sub_mod(c);
    return fr_from_canonical(c);
}
static inline F f_root_of_unity(size_t log2n) {
    // ROOT_OF_UNITY = 7^((r-1) >> 28), then squared down to order 2^log2n
    uint64_t e[4] = {FR_MOD[0] - 1, FR_MOD[1], FR_MOD[2], FR_MOD[3]};
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 28) | (i < 3 ? e[i + 1] << 36 : 0);
    Fr w = fr_pow_limbs(f_from_u64(7), e);
    for (size_t i = log2n; i < 28; i++) w = f_mul(w, w);
    return w;
}
static inline F f_load(const uint64_t* p) { return fr_from_canonical(p); }
static inline void f_store(F a, uint64_t* p) { fr_to_canonical(a, p); }
static inline E e_load(const uint64_t* p) { return fr_from_canonical(p); }
static inline void e_store(E a, uint64_t* p) { fr_to_canonical(a, p); }

}  // namespace orcbn
