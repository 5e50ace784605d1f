// Micro-benchmark: per-item cost of "sum_i a_i*b_i" over Ext2 with (A) reduce-every-product Karatsuba e2_mul
// and (B) lazy 160-bit accumulators reduced once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../hyper-greco_amd/csrc/gl_field.hpp"
using namespace hg;

struct Acc { u64 lo, hi; u32 top; };
__device__ __forceinline__ void mul128(u64 a, u64 b, u64& lo, u64& hi) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 mid = (u64)a0 * b1 + (p00 >> 32);
    u64 mid2 = (u64)a1 * b0 + (u32)mid;
    hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
    lo = (mid2 << 32) | (u32)p00;
}
__device__ __forceinline__ void acc_mul(Acc& s, u64 a, u64 b) {
    u64 l, h; mul128(a, b, l, h);
    unsigned long long c1, c2;
    s.lo = __builtin_addcll(s.lo, l, 0, &c1);
    s.hi = __builtin_addcll(s.hi, h, c1, &c2);
    s.top += (u32)c2;
}
__device__ __forceinline__ u64 acc_reduce(const Acc& s) {
    u64 r = gl_reduce128(s.lo, s.hi);
    return gl_sub(r, (u64)s.top << 32);  // 2^128 = -2^32 (mod p)
}

template <int V> __global__ __launch_bounds__(256) void k(const E2* a, const E2* b, E2* out, int n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    if (V == 0) {
        E2 s = e2_zero();
        for (int i = 0; i < n; i++) {
            E2 x = a[j + i * stride], y = b[j + i * stride];
#pragma unroll
            for (int r = 0; r < 8; r++) { s = e2_add(s, e2_mul(x, y)); x.c0 ^= 1; y.c1 ^= 2; }
        }
        out[j] = s;
    } else if (V == 2) {
        Acc c00{0,0,0}, c11{0,0,0}, cm{0,0,0};
        for (int i = 0; i < n; i++) {
            E2 x = a[j + i * stride], y = b[j + i * stride];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                acc_mul(c00, x.c0, y.c0); acc_mul(c11, x.c1, y.c1); acc_mul(cm, gl_add(x.c0, x.c1), gl_add(y.c0, y.c1));
                x.c0 ^= 1; y.c1 ^= 2;
            }
        }
        u64 r00 = acc_reduce(c00), r11 = acc_reduce(c11), rm = acc_reduce(cm);
        out[j] = e2(gl_add(r00, gl_mul_small(r11, 7)), gl_sub(gl_sub(rm, r00), r11));
    } else {
        Acc c00{0,0,0}, c11{0,0,0}, c01{0,0,0};
        for (int i = 0; i < n; i++) {
            E2 x = a[j + i * stride], y = b[j + i * stride];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                acc_mul(c00, x.c0, y.c0); acc_mul(c11, x.c1, y.c1);
                acc_mul(c01, x.c0, y.c1); acc_mul(c01, x.c1, y.c0);
                x.c0 ^= 1; y.c1 ^= 2;
            }
        }
        u64 r00 = acc_reduce(c00), r11 = acc_reduce(c11), r01 = acc_reduce(c01);
        out[j] = e2(gl_add(r00, gl_mul_small(r11, 7)), r01);
    }
}
int main() {
    const int n = 50; const size_t T = 256 * 1024;  // threads
    std::vector<E2> h(T * n);
    u64 x = 88172645463325252ULL;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.c0 = x % GL_P; x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.c1 = x % GL_P; }
    E2 *da, *db, *d0, *d1, *d2; hipMalloc(&d2, T * 16); hipMalloc(&da, T * n * 16); hipMalloc(&db, T * n * 16); hipMalloc(&d0, T * 16); hipMalloc(&d1, T * 16);
    hipMemcpy(da, h.data(), T * n * 16, hipMemcpyHostToDevice);
    hipMemcpy(db, h.data() + 7, (T * n - 7) * 16, hipMemcpyHostToDevice);
    for (int v = 0; v < 3; v++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v == 0) k<0><<<T / 256, 256>>>(da, db, d0, n); else if (v == 1) k<1><<<T / 256, 256>>>(da, db, d1, n); else k<2><<<T / 256, 256>>>(da, db, d2, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("V%d: %.3f ms  %.1f G e2-products/s  %.1f GB/s\n", v, ms, T * n * 8.0 / (ms * 1e-3) / 1e9, T * n * 32.0 / (ms * 1e-3) / 1e9);
    }
    std::vector<E2> o0(T), o1(T);
    hipMemcpy(o0.data(), d0, T * 16, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, T * 16, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < T; i++) bad += (o0[i].c0 != o1[i].c0 || o0[i].c1 != o1[i].c1);
    printf("mismatches: %zu\n", bad);
    hipMemcpy(o1.data(), d2, T * 16, hipMemcpyDeviceToHost);
    bad = 0; for (size_t i = 0; i < T; i++) bad += (o0[i].c0 != o1[i].c0 || o0[i].c1 != o1[i].c1);
    printf("mismatches V2: %zu\n", bad);
}
