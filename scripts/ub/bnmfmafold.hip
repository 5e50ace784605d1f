// The fold of a sum-check round over bn256::Fr, x + r (y - x), as an int8 matrix product (round 6 experiment).
//
// With r fixed for a launch the fold is LINEAR in the 64 bytes of (x, y):  x + r (y - x) = sum_i x_i Cx_i + sum_i y_i Cy_i  (mod p),
// Cx_i = (1 - r) 2^(8 i) mod p, Cy_i = r 2^(8 i) mod p (x_i, y_i = the bytes of the Montgomery residues as they lie in HBM). Written
// with the constants' bytes as a 32 x 64 matrix A (row = output byte position, signed digits in [-128, 127]) that is
//     D[rho][n] = sum_kappa A[rho][kappa] B[kappa][n]        B[.][n] = the 64 bytes of element n's (x, y), each XOR 0x80 (-> signed)
// = two v_mfma_i32_32x32x32_i8 per 32 elements, with the -128 offsets and a bias that keeps every D non-negative in the C operand.
// The wave then holds, per element, 32 column sums < 2^22 at byte weights 0..31, split over lanes n and n + 32: four v_lshl_add_u64
// triples pack them into 64-bit columns at 32-bit strides, v_permlane32_swap brings an element's two halves into one lane, and a
// short quotient-estimate finish (q < 2^17: eight v_mad_u64_u32, no carry banks) leaves a loose residue in [0, 2p).
// Per fold: ~80 VALU instructions against ~230 for lz_fold (64 mad + 64 addc + lz_sub + lz_finish); the matrix pipe does the products.
//
// Checks every result against lz_fold (canonical forms equal) and times both on tables in HBM.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ihyper-greco_amd/csrc scripts/ub/bnmfmafold.hip -o scripts/ub/bnmfmafold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "bn254_lazy.hpp"
using namespace hg::bn;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// what a lane needs of the launch's constants: its rows of A for the two K steps (x bytes, y bytes) and its 16 C-operand values
struct MfLane { v4i a[2]; int c[16]; };
struct MfFoldK { MfLane lane[64]; };

// ---- host: constants ---------------------------------------------------------------------------------------------------------
static void signed_digits(const Fr& v, int8_t d[32]) {   // v < p < 2^254: 32 digits in [-128, 127], value preserved
    int carry = 0;
    for (int i = 0; i < 32; i++) {
        int b = (int)((v.l[i >> 3] >> (8 * (i & 7))) & 0xff) + carry;
        carry = 0;
        if (b >= 128) { b -= 256; carry = 1; }
        d[i] = (int8_t)b;
    }
    if (carry) { fprintf(stderr, "signed_digits: overflow\n"); exit(1); }
}
static Fr fr_times_256(Fr v) { for (int i = 0; i < 8; i++) v = fr_add(v, v); return v; }
static void mf_fold_consts(const Fr& r_mont, MfFoldK* out) {
    // constants as plain integers mod p (they multiply raw bytes): Cy_i = r 2^(8 i), Cx_i = (1 - r) 2^(8 i), r = r_mont R^-1
    const Fr r_plain = fr_mul(r_mont, fr_make(1, 0, 0, 0));
    const Fr omr = fr_sub(fr_make(1, 0, 0, 0), r_plain);
    int8_t A[32][64];
    Fr sum = fr_zero();   // sum over kappa of the constants
    Fr cx = omr, cy = r_plain;
    for (int i = 0; i < 32; i++) {
        int8_t d[32];
        signed_digits(cx, d);
        for (int rho = 0; rho < 32; rho++) A[rho][i] = d[rho];
        signed_digits(cy, d);
        for (int rho = 0; rho < 32; rho++) A[rho][32 + i] = d[rho];
        sum = fr_add(sum, fr_add(cx, cy));
        cx = fr_times_256(cx);
        cy = fr_times_256(cy);
    }
    // bytes b = s + 128 (s = b ^ 0x80 as a signed byte): sum_kappa A b = sum_kappa A s + 128 sum_kappa const_kappa. With |sum A s| <= 2^20 per
    // row, C[rho] = 2^20 + t[rho], t = the bytes of T = 128 sum_kappa const_kappa - sum_rho 2^20 2^(8 rho)  (mod p)
    Fr T = sum;
    for (int i = 0; i < 7; i++) T = fr_add(T, T);
    Fr w = fr_make(1ULL << 20, 0, 0, 0), bias = fr_zero();
    for (int rho = 0; rho < 32; rho++) { bias = fr_add(bias, w); w = fr_times_256(w); }
    T = fr_sub(T, bias);
    for (int l = 0; l < 64; l++) {
        const int row = l & 31, h = l >> 5;
        for (int s = 0; s < 2; s++) {
            uint32_t wds[4];
            for (int q = 0; q < 4; q++) {
                uint32_t x = 0;
                for (int e = 0; e < 4; e++) x |= (uint32_t)(uint8_t)A[row][32 * s + 16 * h + 4 * q + e] << (8 * e);
                wds[q] = x;
            }
            out->lane[l].a[s] = v4i{(int)wds[0], (int)wds[1], (int)wds[2], (int)wds[3]};
        }
        for (int v = 0; v < 16; v++) {
            const int rho = (v & 3) + 8 * (v >> 2) + 4 * h;
            out->lane[l].c[v] = (1 << 20) + (int)((T.l[rho >> 3] >> (8 * (rho & 7))) & 0xff);
        }
    }
}

// ---- device ------------------------------------------------------------------------------------------------------------------
// V = sum_k C[k] 2^(32 k), C[k] < 2^47 (so V < 2^271 and V / p < 2^18)  ->  V mod p as a loose residue in [0, 2p)
__device__ __forceinline__ Fr mf_finish(u64* C) {
    const double top = (double)C[7] + (double)(u32)(C[6] >> 32);          // V / 2^224, short by less than 2 units
    double qd = top * 1.2317090423844144e-09 - 0.0009765625;               // 2^224 / p; rounded down from slightly below
    qd = qd > 0.0 ? qd : 0.0;
    const u32 q = (u32)qd;
    constexpr u32 NP[8] = {LZ_NP0, LZ_NP1, LZ_NP2, LZ_NP3, LZ_NP4, LZ_NP5, LZ_NP6, LZ_NP7};   // 2^256 - p
    u32 L[8];
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        s += C[k] + (u64)q * NP[k];      // < 2^47 + 2^50 + carry: no overflow
        L[k] = (u32)s;
        s >>= 32;
    }
    return fr_make((u64)L[0] | ((u64)L[1] << 32), (u64)L[2] | ((u64)L[3] << 32), (u64)L[4] | ((u64)L[5] << 32), (u64)L[6] | ((u64)L[7] << 32));
}
// 16 column sums of a lane (rows (v & 3) + 8 (v >> 2) + 4 h) -> four 64-bit values u_g = sum_e D[4 g + e] 2^(8 e) at weight 2^(64 g + 32 h)
__device__ __forceinline__ void mf_pack(const v16i& D, u64 u[4]) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const u32 w0 = (u32)D[4 * g] + ((u32)D[4 * g + 1] << 8);       // < 2^31
        const u32 w1 = (u32)D[4 * g + 2] + ((u32)D[4 * g + 3] << 8);
        u[g] = (u64)w0 + ((u64)w1 << 16);
    }
}
// One wave folds 64 (x, y) pairs: bx0 / by0 = the 16 bytes this lane holds of element (lane & 31)'s x / y (bytes 16 h .. 16 h + 15,
// h = lane >> 5), bx1 / by1 = the same of element 32 + (lane & 31). Returns the fold of element `lane`.
__device__ __forceinline__ Fr mf_fold(const MfLane& K, v4i bx0, v4i by0, v4i bx1, v4i by1) {
    const v4i sgn = {(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
    v16i c;
#pragma unroll
    for (int v = 0; v < 16; v++) c[v] = K.c[v];
    v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(K.a[0], bx0 ^ sgn, c, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(K.a[1], by0 ^ sgn, d0, 0, 0, 0);
    v16i d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(K.a[0], bx1 ^ sgn, c, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(K.a[1], by1 ^ sgn, d1, 0, 0, 0);
    u64 P[4], Q[4];
    mf_pack(d0, P);
    mf_pack(d1, Q);
    u64 C[8];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        // lanes 32..63 of P (half 1 of elements 0..31) <-> lanes 0..31 of Q (half 0 of elements 32..63): afterwards P = half 0 and
        // Q = half 1 of element `lane`
        auto lo = __builtin_amdgcn_permlane32_swap((u32)P[g], (u32)Q[g], false, false);
        auto hi = __builtin_amdgcn_permlane32_swap((u32)(P[g] >> 32), (u32)(Q[g] >> 32), false, false);
        C[2 * g] = (u64)lo[0] | ((u64)hi[0] << 32);
        C[2 * g + 1] = (u64)lo[1] | ((u64)hi[1] << 32);
    }
    return mf_finish(C);
}

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i gload16(const void* p) {
    const __attribute__((address_space(1))) v4i* g = (const __attribute__((address_space(1))) v4i*)p;
    return *g;
}
// tables as the round kernels see them: entry j = (T[2 j], T[2 j + 1]) adjacent (64 bytes); out[j] = fold
template <int KIND>
__global__ __launch_bounds__(256) void k_fold(const Fr* __restrict__ in, Fr* __restrict__ out, size_t half, const MfFoldK* __restrict__ MK, FoldK fk, int reps) {
    const int lane = threadIdx.x & 63;
    const size_t wave0 = ((size_t)blockIdx.x * 256 + (threadIdx.x & ~63));
    if (KIND == 1) {
        const MfLane K = MK->lane[lane];
        for (size_t jw = wave0; jw < half; jw += (size_t)gridDim.x * 256) {
            const int n = lane & 31, h = lane >> 5;
            const char* e0 = reinterpret_cast<const char*>(&in[2 * (jw + n)]) + 16 * h;
            const char* e1 = reinterpret_cast<const char*>(&in[2 * (jw + 32 + n)]) + 16 * h;
            v4i bx0 = gload16(e0), by0 = gload16(e0 + 32), bx1 = gload16(e1), by1 = gload16(e1 + 32);
            Fr f;
            for (int rep = 0; rep < reps; rep++) {
                f = mf_fold(K, bx0, by0, bx1, by1);
                bx0.x ^= (int)f.l[0] & 0x01010101;   // (timing loop: keeps the repetitions dependent; reps = 1 in the check)
            }
            lz_gstore(&out[jw + lane], f);
        }
    } else {
        const LzK KK = lz_load_k(fk.k);
        for (size_t jw = wave0; jw < half; jw += (size_t)gridDim.x * 256) {
            Fr x = lz_gload(&in[2 * (jw + lane)]), y = lz_gload(&in[2 * (jw + lane) + 1]);
            Fr f;
            for (int rep = 0; rep < reps; rep++) {
                f = lz_fold(x, lz_sub(y, x), KK.k);
                x.l[0] ^= f.l[0] & 0x01010101;
            }
            lz_gstore(&out[jw + lane], f);
        }
    }
}
// raw MFMA layout probe: A[row][k] = (row == k) selects byte k of the B operand -> D[row][n] = B byte `row` of element n
__global__ void k_probe(int* out) {
    const int lane = threadIdx.x, row = lane & 31, h = lane >> 5;
    v4i a = {0, 0, 0, 0}, b;
    for (int j = 0; j < 16; j++) if (16 * h + j == row) a[j >> 2] |= 1 << (8 * (j & 3));
    for (int q = 0; q < 4; q++) {
        int x = 0;
        for (int e = 0; e < 4; e++) x |= ((((lane & 31) * 3 + (16 * h + 4 * q + e)) & 0x7f)) << (8 * e);   // B[k][n] = (3 n + k) & 127
        b[q] = x;
    }
    v16i c;
    for (int v = 0; v < 16; v++) c[v] = 0;
    const v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int v = 0; v < 16; v++) out[lane * 16 + v] = d[v];
}

static u64 rng_state = 0x9E3779B97F4A7C15ULL;
static u64 rnd64() { rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL; return rng_state ^ (rng_state >> 29); }
int main() {
    {   // layout probe
        int* d; hipMalloc(&d, 64 * 16 * 4);
        k_probe<<<1, 64>>>(d);
        std::vector<int> h(64 * 16);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 16; v++) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5), n = l & 31;
            if (h[l * 16 + v] != ((3 * n + row) & 127)) bad++;
        }
        printf("i8 32x32x32 layout probe: %d of 1024 values off (A: lane = row, bytes k = 16 h + j; B: lane = column, bytes k = 16 h + j; D: row = (v & 3) + 8 (v >> 2) + 4 h)\n", bad);
    }
    const size_t half = (size_t)1 << 22;
    std::vector<Fr> h_in(2 * half);
    for (size_t i = 0; i < 2 * half; i++) {
        Fr v = fr_make(rnd64(), rnd64(), rnd64(), rnd64() & 0x3FFFFFFFFFFFFFFFULL);
        // loose representatives: anything below 2p
        while (v.l[3] > LZ_2P3 - 1) v.l[3] >>= 1;
        if (i % 1024 == 0) v = fr_make(LZ_2P0 - 1, LZ_2P1, LZ_2P2, LZ_2P3);   // 2p - 1
        if (i % 1024 == 1) v = fr_zero();
        if (i % 1024 == 2) v = fr_make(~0ULL, ~0ULL, ~0ULL, 0);
        if (i % 1024 == 3) v = fr_make(0x8080808080808080ULL, 0x8080808080808080ULL, 0x7f7f7f7f7f7f7f7fULL, 0x2f7f7f7f80808080ULL);
        h_in[i] = v;
    }
    Fr *d_in, *d_a, *d_b;
    hipMalloc(&d_in, sizeof(Fr) * 2 * half); hipMalloc(&d_a, sizeof(Fr) * half); hipMalloc(&d_b, sizeof(Fr) * half);
    hipMemcpy(d_in, h_in.data(), sizeof(Fr) * 2 * half, hipMemcpyHostToDevice);
    MfFoldK* d_mk; hipMalloc(&d_mk, sizeof(MfFoldK));
    int total_bad = 0;
    for (int trial = 0; trial < 4; trial++) {
        Fr r = fr_make(rnd64(), rnd64(), rnd64(), rnd64() & 0x0FFFFFFFFFFFFFFFULL);
        if (trial == 1) r = fr_zero();
        if (trial == 2) r = fr_one_mont();
        if (trial == 3) r = fr_make(FR_P0 - 1, FR_P1, FR_P2, FR_P3);
        FoldK fk; fold_consts(r, &fk);
        MfFoldK mk; mf_fold_consts(r, &mk);
        hipMemcpy(d_mk, &mk, sizeof(mk), hipMemcpyHostToDevice);
        k_fold<0><<<2048, 256>>>(d_in, d_a, half, d_mk, fk, 1);
        k_fold<1><<<2048, 256>>>(d_in, d_b, half, d_mk, fk, 1);
        std::vector<Fr> a(half), b(half);
        hipMemcpy(a.data(), d_a, sizeof(Fr) * half, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), d_b, sizeof(Fr) * half, hipMemcpyDeviceToHost);
        size_t bad = 0, loose_bad = 0;
        for (size_t i = 0; i < half; i++) {
            Fr x = a[i], y = b[i];
            // both loose (< 2p): compare canonical forms
            auto canon = [](Fr v) { while (fr_geq_p(v)) v = fr_sub_p(v); return v; };
            const Fr twop = fr_make(LZ_2P0, LZ_2P1, LZ_2P2, LZ_2P3);
            auto lt = [](const Fr& u, const Fr& w) { for (int k = 3; k >= 0; k--) if (u.l[k] != w.l[k]) return u.l[k] < w.l[k]; return false; };
            if (!lt(y, twop)) loose_bad++;
            x = canon(x); y = canon(y);
            if (memcmp(&x, &y, sizeof(Fr)) != 0) { if (bad < 3) printf("  mismatch at %zu: %016llx.. vs %016llx..\n", i, (unsigned long long)x.l[0], (unsigned long long)y.l[0]); bad++; }
        }
        printf("trial %d: %zu of %zu folds differ from lz_fold, %zu results not below 2p\n", trial, bad, half, loose_bad);
        total_bad += (int)(bad + loose_bad);
    }
    for (int kind = 0; kind < 2; kind++) for (int reps : {1, 16}) {
        Fr r = fr_make(0x123456789abcdef1ULL, 0x0fedcba987654321ULL, 0x1111222233334444ULL, 0x2064aaaabbbbccccULL);
        FoldK fk; fold_consts(r, &fk);
        MfFoldK mk; mf_fold_consts(r, &mk);
        hipMemcpy(d_mk, &mk, sizeof(mk), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            hipEventRecord(e0);
            if (kind == 0) k_fold<0><<<2048, 256>>>(d_in, d_a, half, d_mk, fk, reps); else k_fold<1><<<2048, 256>>>(d_in, d_b, half, d_mk, fk, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-34s reps %2d: %8.3f ms  %7.1f G folds/s  (%.0f GB/s of table traffic)\n", kind == 0 ? "lz_fold (64 mad + 64 addc)" : "mf_fold (int8 MFMA)", reps, best,
               (double)half * reps / (best * 1e-3) / 1e9, (double)half * 96 / (best * 1e-3) / 1e9);
    }
    return total_bad ? 1 : 0;
}
