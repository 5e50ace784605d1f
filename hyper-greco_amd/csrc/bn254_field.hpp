// BN254 scalar field Fr (r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001) in four 64-bit
// limbs, Montgomery form (R = 2^256) inside the kernels, canonical little-endian limbs at the C ABI - the field of the
// reference's `bn254` test family, where the extension is the field itself
// [REF bfv-gkr/src/sk_encryption_circuit.rs:540,614-626: generate_sk_enc_test!("bn254", Fr, Fr, ..); halo2curves 0.7.0
// bn256::Fr, Cargo.toml:29]. Field layer of the BN254 path (BASELINE config 5), shared by the kernels (bn254.hip) and
// the host verifier (verifier.cpp); the Goldilocks prover does not use this header.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace hg {
namespace bn {

typedef uint64_t u64;
typedef unsigned __int128 u128;

struct __attribute__((aligned(16))) Fr {
    u64 l[4];
};

#define BN_HD __host__ __device__ __forceinline__

constexpr u64 FR_P0 = 0x43e1f593f0000001ULL, FR_P1 = 0x2833e84879b97091ULL, FR_P2 = 0xb85045b68181585dULL, FR_P3 = 0x30644e72e131a029ULL;
constexpr u64 FR_INV = 0xc2e1f593efffffffULL;  // -r^-1 mod 2^64

BN_HD Fr fr_make(u64 a, u64 b, u64 c, u64 d) { Fr r; r.l[0] = a; r.l[1] = b; r.l[2] = c; r.l[3] = d; return r; }
BN_HD Fr fr_zero() { return fr_make(0, 0, 0, 0); }
BN_HD Fr fr_one_mont() { return fr_make(0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL); }  // R mod r
BN_HD Fr fr_r2() { return fr_make(0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL); }        // R^2 mod r

BN_HD bool fr_geq_p(const Fr& a) {
    if (a.l[3] != FR_P3) return a.l[3] > FR_P3;
    if (a.l[2] != FR_P2) return a.l[2] > FR_P2;
    if (a.l[1] != FR_P1) return a.l[1] > FR_P1;
    return a.l[0] >= FR_P0;
}
BN_HD Fr fr_sub_p(const Fr& a) {
    Fr r;
    u128 d = (u128)a.l[0] - FR_P0;
    r.l[0] = (u64)d;
    d = (u128)a.l[1] - FR_P1 - (u64)((d >> 64) & 1);
    r.l[1] = (u64)d;
    d = (u128)a.l[2] - FR_P2 - (u64)((d >> 64) & 1);
    r.l[2] = (u64)d;
    r.l[3] = a.l[3] - FR_P3 - (u64)((d >> 64) & 1);
    return r;
}
BN_HD Fr fr_add(const Fr& a, const Fr& b) {  // a, b < r < 2^254: no carry out of the top limb
    Fr s;
    u128 c = (u128)a.l[0] + b.l[0];
    s.l[0] = (u64)c;
    c = (u128)a.l[1] + b.l[1] + (u64)(c >> 64);
    s.l[1] = (u64)c;
    c = (u128)a.l[2] + b.l[2] + (u64)(c >> 64);
    s.l[2] = (u64)c;
    s.l[3] = a.l[3] + b.l[3] + (u64)(c >> 64);
    return fr_geq_p(s) ? fr_sub_p(s) : s;
}
BN_HD Fr fr_sub(const Fr& a, const Fr& b) {
    Fr d;
    u128 c = (u128)a.l[0] - b.l[0];
    d.l[0] = (u64)c;
    c = (u128)a.l[1] - b.l[1] - (u64)((c >> 64) & 1);
    d.l[1] = (u64)c;
    c = (u128)a.l[2] - b.l[2] - (u64)((c >> 64) & 1);
    d.l[2] = (u64)c;
    c = (u128)a.l[3] - b.l[3] - (u64)((c >> 64) & 1);
    d.l[3] = (u64)c;
    if ((c >> 64) & 1) {  // borrowed: add r back
        u128 e = (u128)d.l[0] + FR_P0;
        d.l[0] = (u64)e;
        e = (u128)d.l[1] + FR_P1 + (u64)(e >> 64);
        d.l[1] = (u64)e;
        e = (u128)d.l[2] + FR_P2 + (u64)(e >> 64);
        d.l[2] = (u64)e;
        d.l[3] = d.l[3] + FR_P3 + (u64)(e >> 64);
    }
    return d;
}
BN_HD Fr fr_dbl(const Fr& a) { return fr_add(a, a); }
// Montgomery product a b R^-1 mod r (CIOS; r has two spare top bits, so the running sum fits five limbs)
BN_HD Fr fr_mul(const Fr& a, const Fr& b) {
    u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 bi = b.l[i];
        u128 c = (u128)a.l[0] * bi + t0;
        t0 = (u64)c;
        c = (u128)a.l[1] * bi + t1 + (u64)(c >> 64);
        t1 = (u64)c;
        c = (u128)a.l[2] * bi + t2 + (u64)(c >> 64);
        t2 = (u64)c;
        c = (u128)a.l[3] * bi + t3 + (u64)(c >> 64);
        t3 = (u64)c;
        t4 += (u64)(c >> 64);
        const u64 m = t0 * FR_INV;
        c = (u128)m * FR_P0 + t0;
        c = (u128)m * FR_P1 + t1 + (u64)(c >> 64);
        t0 = (u64)c;
        c = (u128)m * FR_P2 + t2 + (u64)(c >> 64);
        t1 = (u64)c;
        c = (u128)m * FR_P3 + t3 + (u64)(c >> 64);
        t2 = (u64)c;
        c = (u128)t4 + (u64)(c >> 64);
        t3 = (u64)c;
        t4 = (u64)(c >> 64);
    }
    Fr r = fr_make(t0, t1, t2, t3);
    return (t4 || fr_geq_p(r)) ? fr_sub_p(r) : r;
}
// ---- deferred Montgomery reduction: sums of products stay as 512(+64)-bit integers and are reduced once --------------------
struct W512 {
    u64 l[9];  // little-endian limbs; l[8] collects the overflow of up to 2^64 accumulated products
};
BN_HD W512 w512_zero() { W512 w; for (int i = 0; i < 9; i++) w.l[i] = 0; return w; }
// acc += a * b (plain integer product of two residues < r)
BN_HD void w512_mac(W512& acc, const Fr& a, const Fr& b) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 bi = b.l[i];
        u128 c = (u128)a.l[0] * bi + acc.l[i];
        acc.l[i] = (u64)c;
        c = (u128)a.l[1] * bi + acc.l[i + 1] + (u64)(c >> 64);
        acc.l[i + 1] = (u64)c;
        c = (u128)a.l[2] * bi + acc.l[i + 2] + (u64)(c >> 64);
        acc.l[i + 2] = (u64)c;
        c = (u128)a.l[3] * bi + acc.l[i + 3] + (u64)(c >> 64);
        acc.l[i + 3] = (u64)c;
        // propagate the carry to the top
        u64 carry = (u64)(c >> 64);
#pragma unroll
        for (int k = i + 4; k < 9; k++) {
            const u128 d = (u128)acc.l[k] + carry;
            acc.l[k] = (u64)d;
            carry = (u64)(d >> 64);
        }
    }
}
// acc R^-1 mod r for acc < 2^10 r^2 (Montgomery reduction of the whole sum, then a few conditional subtractions)
BN_HD Fr w512_reduce(W512 t) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u64 m = t.l[i] * FR_INV;
        u128 c = (u128)m * FR_P0 + t.l[i];
        c = (u128)m * FR_P1 + t.l[i + 1] + (u64)(c >> 64);
        t.l[i + 1] = (u64)c;
        c = (u128)m * FR_P2 + t.l[i + 2] + (u64)(c >> 64);
        t.l[i + 2] = (u64)c;
        c = (u128)m * FR_P3 + t.l[i + 3] + (u64)(c >> 64);
        t.l[i + 3] = (u64)c;
        u64 carry = (u64)(c >> 64);
#pragma unroll
        for (int k = i + 4; k < 9; k++) {
            const u128 d = (u128)t.l[k] + carry;
            t.l[k] = (u64)d;
            carry = (u64)(d >> 64);
        }
    }
    // value = t.l[4..8] < 2^10 r: subtract r while it does not fit (top limb) or is >= r
    Fr r = fr_make(t.l[4], t.l[5], t.l[6], t.l[7]);
    u64 top = t.l[8];
    for (int it = 0; it < 1100 && (top || fr_geq_p(r)); it++) {
        const bool borrow = !fr_geq_p(r);  // r < p: the subtraction wraps and takes one from `top`
        r = fr_sub_p(r);
        if (borrow) top--;
    }
    return r;
}

BN_HD Fr fr_to_mont(const Fr& canonical) { return fr_mul(canonical, fr_r2()); }
BN_HD Fr fr_from_mont(const Fr& m) { return fr_mul(m, fr_make(1, 0, 0, 0)); }
BN_HD bool fr_eq(const Fr& a, const Fr& b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }

}  // namespace bn
}  // namespace hg
