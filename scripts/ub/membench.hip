// Memory-only floor of the sum-check round access pattern: per (pair, j) read 4 x 16 B (two tables, entries j and half + j),
// write 2 x 16 B, grid-stride over j, serial loop over the pairs. Variants: grid size, non-temporal stores, pairs per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
struct __attribute__((aligned(16))) E2 { u64 c0, c1; };
template <int NT>
__global__ __launch_bounds__(256) void k(const E2* __restrict__ in, E2* __restrict__ out, size_t half, int nb) {
    size_t ntiles = half >> 8;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        size_t j = (tile << 8) + threadIdx.x;
        size_t jo = (j & 1) * (half >> 1) + (j >> 1);
        u64 acc = 0;
        for (int i = 0; i < nb; i++) {
            const E2* tl = in + (size_t)(2 * i) * 2 * half;
            const E2* tr = in + (size_t)(2 * i + 1) * 2 * half;
            E2 xl = tl[j], yl = tl[half + j], xr = tr[j], yr = tr[half + j];
            E2 ol, orr;
            ol.c0 = xl.c0 + yl.c1; ol.c1 = xl.c1 ^ yl.c0; orr.c0 = xr.c0 + yr.c1; orr.c1 = xr.c1 ^ yr.c0;
            acc += ol.c0 * orr.c1;
            E2* pl = out + (size_t)(2 * i) * half + jo;
            E2* pr = out + (size_t)(2 * i + 1) * half + jo;
            if (NT) { __builtin_nontemporal_store(ol.c0, &pl->c0); __builtin_nontemporal_store(ol.c1, &pl->c1); __builtin_nontemporal_store(orr.c0, &pr->c0); __builtin_nontemporal_store(orr.c1, &pr->c1); }
            else { *pl = ol; *pr = orr; }
        }
        if (acc == 0x1234567) out[0].c0 = acc;
    }
}
int main() {
    const int nb = 50;
    for (int hl : {18, 16, 14}) {
        size_t half = (size_t)1 << hl;
        size_t in_bytes = (size_t)2 * nb * 2 * half * 16, out_bytes = (size_t)2 * nb * half * 16;
        E2 *in, *out; hipMalloc(&in, in_bytes); hipMalloc(&out, out_bytes); hipMemset(in, 1, in_bytes);
        for (int nt = 0; nt < 2; nt++)
            for (int blocks : {512, 1024, 2048, 4096}) {
                if ((size_t)blocks > (half >> 8)) continue;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                float ms = 0, best = 1e9;
                for (int rep = 0; rep < 5; rep++) {
                    hipEventRecord(e0);
                    if (nt) k<1><<<blocks, 256>>>(in, out, half, nb); else k<0><<<blocks, 256>>>(in, out, half, nb);
                    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf("half=2^%d nt=%d blocks=%d: %.1f us  %.0f GB/s (read %.0f MB write %.0f MB)\n", hl, nt, blocks, best * 1e3, (in_bytes + out_bytes) / (best * 1e-3) / 1e9, in_bytes / 1e6, out_bytes / 1e6);
            }
        hipFree(in); hipFree(out);
    }
    return 0;
}
