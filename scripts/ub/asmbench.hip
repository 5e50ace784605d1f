// Micro-benchmark: compiler-generated vs inline-asm carry-chain Goldilocks add / mul / Ext2 mul-accumulate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../hyper-greco_amd/csrc/gl_field.hpp"
using namespace hg;

// r = a + b mod p (a, b canonical): s = a + b (carry c1); t = s + EPS (carry c2: s >= p); r = (c1|c2) ? t : s
__device__ __forceinline__ u64 add_asm(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32), r0, r1, t0, t1;
    u64 c2;
    asm volatile(
        "v_add_co_u32 %0, vcc, %5, %7\n\t"
        "v_addc_co_u32 %1, vcc, %6, %8, vcc\n\t"
        "v_add_co_u32 %2, %4, -1, %0\n\t"
        "v_addc_co_u32 %3, %4, 0, %1, %4\n\t"
        "s_or_b64 vcc, vcc, %4\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %0, %0, %2, vcc\n\t"
        "v_cndmask_b32 %1, %1, %3, vcc\n\t"
        : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1), "=&s"(c2)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "vcc", "scc");
    return ((u64)r1 << 32) | r0;
}
// r = a - b mod p: d = a - b (borrow bw); r = bw ? d - EPS : d   (d - EPS == d + p mod 2^64)
__device__ __forceinline__ u64 sub_asm(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32), r0, r1, m;
    asm volatile(
        "v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc\n\t"      // m = borrow ? 0xFFFFFFFF : 0  (= EPS)
        "v_sub_co_u32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        : "=&v"(r0), "=&v"(r1), "=&v"(m)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "vcc");
    return ((u64)r1 << 32) | r0;
}
// 128-bit product then reduction, all carry chains
__device__ __forceinline__ u64 mul_asm(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 mid = (u64)a0 * b1 + (p00 >> 32);
    u64 mid2 = (u64)a1 * b0 + (u32)mid;
    u64 hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
    u32 l0 = (u32)p00, l1 = (u32)mid2, h0 = (u32)hi, h1 = (u32)(hi >> 32);
    u32 t0, t1, u0, u1, m;
    u64 c;
    asm volatile(
        // t = lo - hh ; borrow -> t -= EPS
        "v_sub_co_u32 %0, vcc, %6, %9\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %7, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %4, 0, -1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %4\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        // u = (hl << 32) - hl
        "v_sub_co_u32 %2, vcc, 0, %8\n\t"
        "v_subbrev_co_u32 %3, vcc, 0, %8, vcc\n\t"
        // r = t + u ; carry -> r += EPS
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "v_addc_co_u32 %1, vcc, %1, %3, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %4, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %4\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        // canonical: t' = r + EPS, carry <=> r >= p
        "v_add_co_u32 %2, %5, -1, %0\n\t"
        "v_addc_co_u32 %3, %5, 0, %1, %5\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %0, %0, %2, %5\n\t"
        "v_cndmask_b32 %1, %1, %3, %5\n\t"
        : "=&v"(t0), "=&v"(t1), "=&v"(u0), "=&v"(u1), "=&v"(m), "=&s"(c)
        : "v"(l0), "v"(l1), "v"(h0), "v"(h1)
        : "vcc");
    return ((u64)t1 << 32) | t0;
}
__device__ __forceinline__ u64 mul_small7_asm(u64 a) { return gl_mul_small(a, 7); }
__device__ __forceinline__ E2 e2_mul_asm(E2 a, E2 b) {
    u64 p0 = mul_asm(a.c0, b.c0), p1 = mul_asm(a.c1, b.c1);
    u64 m = mul_asm(add_asm(a.c0, a.c1), add_asm(b.c0, b.c1));
    return e2(add_asm(p0, mul_small7_asm(p1)), sub_asm(sub_asm(m, p0), p1));
}
__device__ __forceinline__ E2 e2_add_asm(E2 a, E2 b) { return e2(add_asm(a.c0, b.c0), add_asm(a.c1, b.c1)); }

template <int V> __global__ __launch_bounds__(256) void k(const E2* a, const E2* b, E2* out, int n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    E2 s = e2_zero();
    for (int i = 0; i < n; i++) {
        E2 x = a[j + i * stride], y = b[j + i * stride];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (V == 0) s = e2_add(s, e2_mul(x, y)); else s = e2_add_asm(s, e2_mul_asm(x, y));
            x = s; 
        }
    }
    out[j] = s;
}
int main() {
    const int n = 16; const size_t T = 256 * 4096;
    std::vector<E2> h(T * n);
    u64 x = 88172645463325252ULL;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.c0 = x % GL_P; x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.c1 = x % GL_P; }
    h[0] = e2(GL_P - 1, GL_P - 1); h[1] = e2(0, 1); h[2] = e2(GL_P - 1, 0);
    E2 *da, *db, *d0, *d1; hipMalloc(&da, T * n * 16); hipMalloc(&db, T * n * 16); hipMalloc(&d0, T * 16); hipMalloc(&d1, T * 16);
    hipMemcpy(da, h.data(), T * n * 16, hipMemcpyHostToDevice);
    hipMemcpy(db, h.data() + 7, (T * n - 7) * 16, hipMemcpyHostToDevice);
    for (int v = 0; v < 2; v++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v == 0) k<0><<<T / 256, 256>>>(da, db, d0, n); else k<1><<<T / 256, 256>>>(da, db, d1, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("status: %s / %s\n", hipGetErrorString(hipDeviceSynchronize()), hipGetErrorString(hipGetLastError()));
        printf("V%d: %.3f ms  %.1f G e2-mul-acc/s\n", v, ms, T * n * 8.0 / (ms * 1e-3) / 1e9);
    }
    std::vector<E2> o0(T), o1(T);
    hipMemcpy(o0.data(), d0, T * 16, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, T * 16, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < T; i++) bad += (o0[i].c0 != o1[i].c0 || o0[i].c1 != o1[i].c1);
    printf("mismatches: %zu  o0[5]=%llx,%llx o1[5]=%llx,%llx\n", bad, (unsigned long long)o0[5].c0, (unsigned long long)o0[5].c1, (unsigned long long)o1[5].c0, (unsigned long long)o1[5].c1);
}
