// Micro-benchmark: Goldilocks multiplication variants on gfx950 (throughput, G mul/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
constexpr u64 P = 0xFFFFFFFF00000001ULL, EPS = 0xFFFFFFFFULL;

__device__ __forceinline__ u64 red_v0(u64 lo, u64 hi) {
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh; t0 -= (lo < hh) ? EPS : 0;
    u64 t1 = (hl << 32) - hl;
    u64 r = t0 + t1; r += (r < t0) ? EPS : 0;
    r -= (r >= P) ? P : 0;
    return r;
}
__device__ __forceinline__ u64 mul_v0(u64 a, u64 b) { return red_v0(a * b, __umul64hi(a, b)); }

// V1: explicit 4 x (32x32+64) products
__device__ __forceinline__ void mul128(u64 a, u64 b, u64& lo, u64& hi) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 mid = (u64)a0 * b1 + (p00 >> 32);
    u64 mid2 = (u64)a1 * b0 + (u32)mid;
    hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);
    lo = (mid2 << 32) | (u32)p00;
}
__device__ __forceinline__ u64 mul_v1(u64 a, u64 b) { u64 lo, hi; mul128(a, b, lo, hi); return red_v0(lo, hi); }

// V2: reduction with explicit carry builtins
__device__ __forceinline__ u64 red_v2(u64 lo, u64 hi) {
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    // t0 = lo - hh (mod p)
    u64 t0 = lo - hh;
    if (lo < (u64)hh) t0 -= EPS;
    // t1 = hl * (2^32 - 1) as (hl<<32) - hl
    u64 t1 = ((u64)hl << 32) - hl;
    u64 r;
    bool c = __builtin_add_overflow(t0, t1, &r);
    if (c) r += EPS;
    if (r >= P) r -= P;
    return r;
}
__device__ __forceinline__ u64 mul_v2(u64 a, u64 b) { u64 lo, hi; mul128(a, b, lo, hi); return red_v2(lo, hi); }

// V3: lazy — no final canonicalisation (result in [0, 2^64)), inputs any u64
__device__ __forceinline__ u64 mul_v3(u64 a, u64 b) {
    u64 lo, hi; mul128(a, b, lo, hi);
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0 = lo - hh; if (lo < (u64)hh) t0 -= EPS;
    u64 t1 = ((u64)hl << 32) - hl;
    u64 r; bool c = __builtin_add_overflow(t0, t1, &r); if (c) r += EPS;
    return r;
}

template <int V> __device__ __forceinline__ u64 mulv(u64 a, u64 b) {
    if constexpr (V == 0) return mul_v0(a, b);
    else if constexpr (V == 1) return mul_v1(a, b);
    else if constexpr (V == 2) return mul_v2(a, b);
    else return mul_v3(a, b);
}

template <int V> __global__ __launch_bounds__(256) void k(const u64* in, u64* out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 x0 = in[i], x1 = in[i] ^ 0x1234567, x2 = in[i] + 77, x3 = in[i] * 3 % P;
    u64 y = in[i + 1] | 1;
    if (V != 3) { x1 %= P; x2 %= P; y %= P; }
    for (int it = 0; it < iters; it++) {
        x0 = mulv<V>(x0, y); x1 = mulv<V>(x1, y); x2 = mulv<V>(x2, y); x3 = mulv<V>(x3, y);
    }
    out[i] = x0 ^ x1 ^ x2 ^ x3;
}

template <int V> double run(const u64* din, u64* dout, size_t n, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<V><<<n / 256, 256>>>(din, dout, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<V><<<n / 256, 256>>>(din, dout, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)n * iters * 4 / (ms * 1e-3) / 1e9;
}
int main() {
    size_t n = 256 * 256 * 32;  // 2M threads
    std::vector<u64> h(n + 1);
    u64 x = 88172645463325252ULL;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x % P; }
    u64 *din, *dout; hipMalloc(&din, (n + 1) * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(din, h.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    int iters = 256;
    printf("V0 umul64hi      : %.1f Gmul/s\n", run<0>(din, dout, n, iters));
    std::vector<u64> o0(n), o1(n); hipMemcpy(o0.data(), dout, n * 8, hipMemcpyDeviceToHost);
    printf("V1 4xmad         : %.1f Gmul/s\n", run<1>(din, dout, n, iters));
    hipMemcpy(o1.data(), dout, n * 8, hipMemcpyDeviceToHost);
    printf("   V1 == V0: %d\n", (int)(o0 == o1));
    printf("V2 4xmad+carry   : %.1f Gmul/s\n", run<2>(din, dout, n, iters));
    hipMemcpy(o1.data(), dout, n * 8, hipMemcpyDeviceToHost);
    printf("   V2 == V0: %d\n", (int)(o0 == o1));
    printf("V3 lazy          : %.1f Gmul/s\n", run<3>(din, dout, n, iters));
    return 0;
}
