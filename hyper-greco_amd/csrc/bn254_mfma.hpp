// The fold of a sum-check round over bn256::Fr, x + r (y - x), on the int8 matrix cores of gfx950 (device only; round 6).
//
// The reference folds every table entry with one field multiplication per round (gkr's fix_var over bn256::Fr, the same generic
// code as over Goldilocks: bfv-gkr/src/sk_encryption_circuit.rs:614-626). With r fixed for a launch the fold is LINEAR in the 64
// bytes of (x, y) as they lie in HBM (Montgomery residues, loose or canonical):
//     x + r (y - x) = sum_i x_i Cx_i + sum_i y_i Cy_i  (mod p),     Cx_i = (1 - r) 2^(8 i) mod p,  Cy_i = r 2^(8 i) mod p.
// With the constants' bytes as a 32 x 64 matrix A (row = output byte position, signed digits in [-128, 127]: MfA, one per (job,
// round), built on the device by k_bn_mf_consts) that is D[rho][n] = sum_kappa A[rho][kappa] B[kappa][n], B[.][n] = the 64 bytes of
// element n's (x, y) each XOR 0x80 (the bytes as signed values minus 128) - two v_mfma_i32_32x32x32_i8 per 32 elements. The -128
// offsets and a bias that keeps every D in [0, 2^21 + 256) sit in the C operand; since Cx_i + Cy_i = 2^(8 i) that operand does NOT
// depend on r (MF_T below). A wave then holds, per element, 32 column sums at byte weights 0..31, split over lanes n and n + 32:
// three shift-adds per four columns pack them into 64-bit columns at 32-bit strides, v_permlane32_swap brings an element's two halves
// into one lane, and a short quotient-estimate finish (q < 2^18: eight v_mad_u64_u32, no carry banks) leaves a loose residue in
// [0, 2p). About 80 VALU instructions per fold against ~230 for lz_fold (lz_sub + 64 mad + 64 addc + lz_finish); the products run on
// the matrix pipe (4 x 32 cycles per 64 folds). scripts/ub/bnmfmafold.hip: every result equal to lz_fold's, 289 against 136 G folds/s.
// Same field elements as any other fold (exact integer arithmetic), so the proof bytes do not change.
#pragma once
#include "bn254_lazy.hpp"

namespace hg {
namespace bn {

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

struct __attribute__((aligned(16))) MfA { signed char a[32][64]; };   // A[rho][kappa]: kappa < 32 the bytes of x, kappa >= 32 the bytes of y

// T = 128 sum_i 2^(8 i) - sum_rho 2^20 2^(8 rho) mod p, little-endian bytes: C[rho] = 2^20 + MF_T[rho]
__device__ constexpr unsigned char MF_T[32] = {119, 213, 112, 0, 237, 236, 8, 20, 223, 182, 75, 187, 78, 86, 189, 66,
                                               131, 67, 59, 213, 5, 150, 195, 156, 43, 169, 223, 255, 43, 211, 245, 10};

// A of one multiplier (Montgomery form): 64 threads, thread kappa writes column kappa. (r 2^(8 i) as a plain integer mod p is the
// Montgomery product of r R with the plain integer 2^(8 i).)
__device__ __forceinline__ void mf_consts_column(const Fr& r_mont, int kappa, MfA* out) {
    const int i = kappa & 31;
    Fr e = fr_zero();
    e.l[i >> 3] = 1ULL << (8 * (i & 7));
    const Fr coef = kappa < 32 ? fr_sub(fr_one_mont(), r_mont) : r_mont;
    const Fr c = fr_mul(coef, e);   // canonical, below p < 2^254
    int carry = 0;
#pragma unroll 1
    for (int rho = 0; rho < 32; rho++) {
        int b = (int)((c.l[rho >> 3] >> (8 * (rho & 7))) & 0xff) + carry;
        carry = b >= 128 ? 1 : 0;
        out->a[rho][kappa] = (signed char)(b - (carry << 8));   // (no carry out of the top digit: byte 31 of a value below p is at most 0x30)
    }
}

// what a lane keeps for a whole kernel: its 16-byte pieces of A for the two K steps, and its 16 values of the C operand
struct MfLane { mf_v4i a[2]; mf_v16i c; };
__device__ __forceinline__ MfLane mf_load(const MfA* __restrict__ A, int lane) {
    MfLane K;
    const int row = lane & 31, h = lane >> 5;
    const __attribute__((address_space(1))) mf_v4i* g = (const __attribute__((address_space(1))) mf_v4i*)(&A->a[row][16 * h]);
    K.a[0] = g[0];
    K.a[1] = g[2];
#pragma unroll
    for (int v = 0; v < 16; v++) K.c[v] = (1 << 20) + (int)(h ? MF_T[(v & 3) + 8 * (v >> 2) + 4] : MF_T[(v & 3) + 8 * (v >> 2)]);
    return K;
}

// V = sum_k C[k] 2^(32 k), C[k] < 2^47 (so V < 2^271 and V / p < 2^18)  ->  V mod p as a loose residue in [0, 2p)
__device__ __forceinline__ Fr mf_finish(const u64* C) {
    const double top = (double)C[7] + (double)(u32)(C[6] >> 32);          // V / 2^224, short by less than 1.001 units
    double qd = top * 1.2317090423844144e-09 - 0.0009765625;               // 2^224 / p; rounded down from slightly below: floor(V / p) or one less
    qd = qd > 0.0 ? qd : 0.0;
    const u32 q = (u32)qd;
    constexpr u32 NP[8] = {LZ_NP0, LZ_NP1, LZ_NP2, LZ_NP3, LZ_NP4, LZ_NP5, LZ_NP6, LZ_NP7};   // 2^256 - p: V - q p = V + q NP mod 2^256
    u32 L[8];
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        s += C[k] + (u64)q * NP[k];      // < 2^47 + 2^50 + carry
        L[k] = (u32)s;
        s >>= 32;
    }
    return fr_make((u64)L[0] | ((u64)L[1] << 32), (u64)L[2] | ((u64)L[3] << 32), (u64)L[4] | ((u64)L[5] << 32), (u64)L[6] | ((u64)L[7] << 32));
}
// the 16 column sums of a lane (rows (v & 3) + 8 (v >> 2) + 4 h) -> four 64-bit values u_g = sum_e D[4 g + e] 2^(8 e), weight 2^(64 g + 32 h)
__device__ __forceinline__ void mf_pack(const mf_v16i& D, u64 u[4]) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const u32 w0 = (u32)D[4 * g] + ((u32)D[4 * g + 1] << 8);       // < 2^30
        const u32 w1 = (u32)D[4 * g + 2] + ((u32)D[4 * g + 3] << 8);
        u[g] = (u64)w0 + ((u64)w1 << 16);
    }
}
// One wave folds 64 (x, y) pairs; EVERY lane of the wave must be here. bx0 / by0 = the 16 bytes this lane holds of element
// (lane & 31)'s x / y (bytes 16 h .. 16 h + 15, h = lane >> 5), bx1 / by1 = the same of element 32 + (lane & 31). Returns the fold of
// element `lane`, loose.
__device__ __forceinline__ Fr mf_fold(const MfLane& K, mf_v4i bx0, mf_v4i by0, mf_v4i bx1, mf_v4i by1) {
    const mf_v4i sgn = {(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
    mf_v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(K.a[0], bx0 ^ sgn, K.c, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(K.a[1], by0 ^ sgn, d0, 0, 0, 0);
    mf_v16i d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(K.a[0], bx1 ^ sgn, K.c, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(K.a[1], by1 ^ sgn, d1, 0, 0, 0);
    u64 P[4], Q[4];
    mf_pack(d0, P);
    mf_pack(d1, Q);
    u64 C[8];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        // lanes 32..63 of P (half 1 of elements 0..31) <-> lanes 0..31 of Q (half 0 of elements 32..63): afterwards P holds half 0 and
        // Q half 1 of element `lane`
        const auto lo = __builtin_amdgcn_permlane32_swap((u32)P[g], (u32)Q[g], false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((u32)(P[g] >> 32), (u32)(Q[g] >> 32), false, false);
        C[2 * g] = (u64)lo[0] | ((u64)hi[0] << 32);
        C[2 * g + 1] = (u64)lo[1] | ((u64)hi[1] << 32);
    }
    return mf_finish(C);
}
// the wave's 64 entries (T[2 j], T[2 j + 1]), j = jw .. jw + 63, read from HBM in the operand layout (four 16-byte loads per lane, as
// many as a lane-per-entry read) and folded; entries at and beyond `half` are read as entry 0 (their results are the caller's to drop)
__device__ __forceinline__ mf_v4i mf_gload16(const void* p) { return *(const __attribute__((address_space(1))) mf_v4i*)p; }
__device__ __forceinline__ Fr mf_fold_global(const MfLane& K, const Fr* __restrict__ tab, size_t jw, size_t half, int lane) {
    const int n = lane & 31, h = lane >> 5;
    const size_t j0 = jw + n < half ? jw + n : 0, j1 = jw + 32 + n < half ? jw + 32 + n : 0;
    const char* e0 = reinterpret_cast<const char*>(&tab[2 * j0]) + 16 * h;
    const char* e1 = reinterpret_cast<const char*>(&tab[2 * j1]) + 16 * h;
    return mf_fold(K, mf_gload16(e0), mf_gload16(e0 + 32), mf_gload16(e1), mf_gload16(e1 + 32));
}


}  // namespace bn
}  // namespace hg
