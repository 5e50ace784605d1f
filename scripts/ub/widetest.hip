// Correctness + throughput of the deferred-reduction Ext2 dot product (gl_wide.hpp) against e2_mul/e2_add.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../hyper-greco_amd/csrc/gl_wide.hpp"
using namespace hg;

template <int V> __global__ __launch_bounds__(256) void k(const E2* a, const E2* b, E2* out, int n, int reps) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    E2 tot = e2_zero();
    {
        if (V == 0) {
            E2 s = e2_zero();
            for (int i = 0; i < n; i++) {
                E2 x = a[j + i * stride], y = b[j + i * stride];
                for (int rp = 0; rp < reps; rp++) { s = e2_add(s, e2_mul(x, y)); x.c0 += 0x9E3779B97F4A7C15ULL; y.c1 += 0x2545F4914F6CDD1DULL; }
            }
            tot = e2_add(tot, s);
        } else {
            WE2 s = we2_zero();
            for (int i = 0; i < n; i++) {
                E2 x = a[j + i * stride], y = b[j + i * stride];
                for (int rp = 0; rp < reps; rp++) { we2_mac(s, x, y); x.c0 += 0x9E3779B97F4A7C15ULL; y.c1 += 0x2545F4914F6CDD1DULL; }
            }
            tot = e2_add(tot, we2_reduce(s));
        }
    }
    out[j] = tot;
}
int main() {
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000); hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    const int n = 25; const size_t T = 256 * 1024 * 2;
    std::vector<E2> h(T * n + 7);
    u64 x = 88172645463325252ULL;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.c0 = x % GL_P; x ^= x << 13; x ^= x >> 7; x ^= x << 17; v.c1 = x % GL_P; }
    // edge values
    h[0] = e2(GL_P - 1, GL_P - 1); h[7] = e2(GL_P - 1, GL_P - 1); h[1] = e2(0, 0); h[2] = e2(0xFFFFFFFFULL, 0xFFFFFFFF00000000ULL);
    for (int i = 0; i < n; i++) { h[i * T + 5] = e2(GL_P - 1, GL_P - 1); h[i * T + 5 + 7] = e2(GL_P - 1, GL_P - 1); }
    E2 *da, *db, *d0, *d1; hipMalloc(&da, (T * n + 7) * 16); hipMalloc(&db, T * n * 16); hipMalloc(&d0, T * 16); hipMalloc(&d1, T * 16);
    hipMemcpy(da, h.data(), (T * n + 7) * 16, hipMemcpyHostToDevice);
    hipMemcpy(db, h.data() + 7, T * n * 16, hipMemcpyHostToDevice);
    for (int reps : {1, 4, 10}) {
      for (size_t lds : {0, 40960, 81920, 160000}) {
        for (int v = 0; v < 2; v++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (v == 0) k<0><<<T / 256, 256, lds>>>(da, db, d0, n, reps); else k<1><<<T / 256, 256, lds>>>(da, db, d1, n, reps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("lds=%zu reps=%d V%d: %.3f ms  %.1f G e2-mac/s  (%.1f GB/s of operands)\n", lds, reps, v, ms, T * n * (double)reps / (ms * 1e-3) / 1e9, T * n * 32.0 * reps / (ms * 1e-3) / 1e9);
        }
      }
        std::vector<E2> o0(T), o1(T);
        hipMemcpy(o0.data(), d0, T * 16, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d1, T * 16, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < T; i++) bad += (o0[i].c0 != o1[i].c0 || o0[i].c1 != o1[i].c1);
        printf("reps=%d mismatches: %zu   sample %016llx %016llx\n", reps, bad, (unsigned long long)o1[5].c0, (unsigned long long)o1[5].c1);
    }
    return 0;
}
