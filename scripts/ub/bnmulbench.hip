// Throughput of the bn256::Fr products on gfx950: compiler CIOS (fr_mul), column accumulators (fr_mul_wide), and the two
// deferred-reduction dot products (w512_* vs wcol_*) over 25 terms. build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../hyper-greco_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bn254_wide.hpp"
using namespace hg::bn;
template <int KIND>
__global__ __launch_bounds__(256) void k(Fr* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    Fr a = fr_make(t + 1, 2 * t + 3, 5, 7), b = fr_make(11, t, 13, 1);
    Fr acc = fr_zero();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) { a = fr_mul(a, b); b = fr_add(b, a); }
        else if (KIND == 1) { a = fr_mul_wide(a, b); b = fr_add(b, a); }
        else if (KIND == 2) {
            W512 w = w512_zero();
            for (int i = 0; i < 25; i++) { w512_mac(w, a, b); a = fr_add(a, b); b.l[0] ^= (u64)i; }
            acc = fr_add(acc, w512_reduce(w));
        } else {
            WCol w = wcol_zero();
            for (int i = 0; i < 25; i++) { wcol_mac(w, a, b); a = fr_add(a, b); b.l[0] ^= (u64)i; }
            acc = fr_add(acc, wcol_reduce(w));
        }
    }
    out[t] = fr_add(fr_add(a, b), acc);
}
template <int KIND> static void run(const char* name, int iters, double per_iter) {
    Fr* out;
    const int blocks = 256 * 8;
    hipMalloc(&out, sizeof(Fr) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %8.1f G products/s\n", name, ms, (double)blocks * 256 * iters * per_iter / (ms * 1e-3) / 1e9);
    hipFree(out);
}
int main() {
    run<0>("fr_mul (CIOS, hipcc)", 200, 1);
    run<1>("fr_mul_wide (columns)", 200, 1);
    run<2>("w512 dot(25) + reduce", 8, 25);
    run<3>("wcol dot(25) + reduce", 8, 25);
    return 0;
}
