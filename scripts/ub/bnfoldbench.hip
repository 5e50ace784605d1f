// Fold x + r d over bn256::Fr with a launch-wide multiplier: fr_add(x, fr_mul_wide(r, d)) vs fr_fold_const (bn254_wide.hpp).
// Checks equality on pseudo-random residues (including edge values) and times both.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ihyper-greco_amd/csrc scripts/ub/bnfoldbench.hip -o scripts/ub/bnfoldbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bn254_wide.hpp"
using namespace hg::bn;
__device__ __forceinline__ Fr rnd(u64& s) {
    Fr v;
    for (int i = 0; i < 4; i++) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; v.l[i] = s ^ (s >> 29); }
    v.l[3] &= 0x3FFFFFFFFFFFFFFFULL;
    while (fr_geq_p(v)) v = fr_sub_p(v);
    return v;
}
__global__ void k_check(const FoldK* K, Fr r, unsigned long long* bad, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0x9E3779B97F4A7C15ULL * (t + 1);
    for (int it = 0; it < n; it++) {
        Fr x = rnd(s), d = rnd(s);
        if (it == 0) { x = fr_make(FR_P0 - 1, FR_P1, FR_P2, FR_P3); d = x; }   // p - 1
        if (it == 1) { x = fr_zero(); d = fr_zero(); }
        if (it == 2) { x = fr_make(FR_P0 - 1, FR_P1, FR_P2, FR_P3); d = fr_make(1, 0, 0, 0); }
        if (it == 3) { d = fr_make(~0ULL, ~0ULL, ~0ULL, 0x30644e72e131a028ULL); }
        const Fr a = fr_add(x, fr_mul_wide(r, d));
        const Fr b = fr_fold_const(x, d, K->k);
        if (a.l[0] != b.l[0] || a.l[1] != b.l[1] || a.l[2] != b.l[2] || a.l[3] != b.l[3]) atomicAdd(bad, 1ULL);
    }
}
template <int KIND>
__global__ __launch_bounds__(256) void k_time(const FoldK* K, Fr r, Fr* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    Fr x = fr_make(t + 1, 2 * t + 3, 5, 7), d = fr_make(11, t, 13, 1);
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) x = fr_add(x, fr_mul_wide(r, d));
        else x = fr_fold_const(x, d, K->k);
        d.l[0] ^= x.l[1];
    }
    out[t] = x;
}
int main() {
    Fr r = fr_make(0x123456789abcdef1ULL, 0x0fedcba987654321ULL, 0x1111222233334444ULL, 0x2064aaaabbbbccccULL);
    FoldK hk, *dk;
    fold_consts(r, &hk);
    hipMalloc(&dk, sizeof(FoldK)); hipMemcpy(dk, &hk, sizeof(FoldK), hipMemcpyHostToDevice);
    unsigned long long* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    k_check<<<1024, 256>>>(dk, r, bad, 64);
    unsigned long long hb = 0; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    printf("mismatches over %d folds: %llu\n", 1024 * 256 * 64, hb);
    Fr* out; const int blocks = 2048; hipMalloc(&out, sizeof(Fr) * blocks * 256);
    for (int kind = 0; kind < 2; kind++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 200;
        if (kind == 0) k_time<0><<<blocks, 256>>>(dk, r, out, 2); else k_time<1><<<blocks, 256>>>(dk, r, out, 2);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (kind == 0) k_time<0><<<blocks, 256>>>(dk, r, out, iters); else k_time<1><<<blocks, 256>>>(dk, r, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %8.3f ms  %8.1f G folds/s\n", kind == 0 ? "fr_add(x, fr_mul_wide(r, d))" : "fr_fold_const(x, d, K)", ms, (double)blocks * 256 * iters / (ms * 1e-3) / 1e9);
    }
    return hb != 0;
}
