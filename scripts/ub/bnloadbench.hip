// Load-path microbenchmark for the BN254 grand-product round kernel's access pattern (scripts/ub): every lane needs 64 contiguous
// bytes (x, y of one pair index) from each of two tables per (pair, j) item; rows of a table are 2^19 * 64 B apart.
//   mode 0: four global_load_dwordx4 per table and lane (what the compiler emits for two 32-byte struct loads)
//   mode 1: the same bytes by LDS-DMA, lane-owned 16-byte pieces (64 lines per instruction, each line fetched four times)
//   mode 2: LDS-DMA of the wave's contiguous 4 KiB run as four coalesced 1 KiB pieces, read back 64 B per lane (bank conflicts)
//   mode 3: mode 2 without the read-back (DMA cost alone)
//   mode 4: coalesced plain loads of the run (lane L takes bytes 1024 k + 16 L), no redistribution (bandwidth reference)
// build: hipcc -O3 --offload-arch=gfx950 bnloadbench.hip -o bnloadbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
typedef u32 x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* gsrc, u32 lds_dst) {
    u32 keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_load(const char* __restrict__ L, const char* __restrict__ R, size_t half, int rows, u64* __restrict__ out, int work) {
    __shared__ x4 stage[4][8][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(&stage[wave][0][0]);
    const size_t pitch = half * 64;
    x4 acc = {0, 0, 0, 0};
    for (size_t jw = (size_t)blockIdx.x * 256 + 64 * wave; jw < half; jw += (size_t)gridDim.x * 256) {
        const size_t j = jw + lane;
        for (int i = 0; i < rows; i++) {
            const __attribute__((address_space(1))) x4* gl = (const __attribute__((address_space(1))) x4*)(L + (size_t)i * pitch + j * 64);
            const __attribute__((address_space(1))) x4* gr = (const __attribute__((address_space(1))) x4*)(R + (size_t)i * pitch + j * 64);
            x4 v[8];
            if (MODE == 0) {
                for (int k = 0; k < 4; k++) { v[k] = gl[k]; v[4 + k] = gr[k]; }
            } else if (MODE == 1) {
                for (int k = 0; k < 4; k++) { glds16((const char*)gl + 16 * k, lds0 + 1024u * k); glds16((const char*)gr + 16 * k, lds0 + 4096u + 1024u * k); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int k = 0; k < 8; k++) v[k] = stage[wave][k][lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else if (MODE == 2 || MODE == 3) {
                const char* wl = L + (size_t)i * pitch + jw * 64 + 16 * lane;
                const char* wr = R + (size_t)i * pitch + jw * 64 + 16 * lane;
                for (int k = 0; k < 4; k++) { glds16(wl + 1024 * k, lds0 + 1024u * k); glds16(wr + 1024 * k, lds0 + 4096u + 1024u * k); }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (MODE == 2) {
                    const x4* mine = &stage[wave][0][0] + 4 * lane;
                    for (int k = 0; k < 4; k++) { v[k] = mine[k]; v[4 + k] = mine[256 + k]; }
                } else for (int k = 0; k < 8; k++) v[k] = stage[wave][k][lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                const __attribute__((address_space(1))) x4* wl = (const __attribute__((address_space(1))) x4*)(L + (size_t)i * pitch + jw * 64 + 16 * lane);
                const __attribute__((address_space(1))) x4* wr = (const __attribute__((address_space(1))) x4*)(R + (size_t)i * pitch + jw * 64 + 16 * lane);
                for (int k = 0; k < 4; k++) { v[k] = wl[64 * k]; v[4 + k] = wr[64 * k]; }
            }
            for (int k = 0; k < 8; k++) acc ^= v[k];
            for (int w = 0; w < work; w++) acc = acc * acc + v[w & 7];   // optional VALU work per item (dependent chain of 4 mul + 4 add)
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[threadIdx.x] = acc.x;
}
// compute phase: NI dependent-chain steps on four independent 64-bit accumulators (v_mad_u64_u32), seeded by the item's data
__device__ __forceinline__ void compute(u64* a, const x4* v, int steps) {
    for (int w = 0; w < steps; w++) {
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] = (u64)(u32)a[q] * (u32)(v[q].x | 1u) + a[(q + 1) & 3] + v[4 + q].y;
    }
}
// modes 5..7: the round kernel's shape - load an item (128 B per lane), then ~1000 VALU instructions on it
//   5: plain loads, wait, compute;  6: LDS-DMA prefetch one item ahead;  7: two items ahead
template <int DEPTH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pipe(const char* __restrict__ L, const char* __restrict__ R, size_t half, int rows, u64* __restrict__ out, int steps) {
    __shared__ x4 stage[2][4][8][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 lds0 = (u32)(uintptr_t)(__attribute__((address_space(3))) void*)(&stage[0][wave][0][0]);
    const size_t pitch = half * 64;
    u64 a[4] = {1, 2, 3, 4};
    const size_t jw0 = (size_t)blockIdx.x * 256 + 64 * wave, jstep = (size_t)gridDim.x * 256;
    const u32 n_items = jw0 < half ? (u32)((half - jw0 + jstep - 1) / jstep) * rows : 0;
    u32 issued = 0, t = 0; int pf_i = 0; size_t pf_jw = jw0;
    auto issue_next = [&] {
        const char* gl = L + (size_t)pf_i * pitch + (pf_jw + lane) * 64;
        const char* gr = R + (size_t)pf_i * pitch + (pf_jw + lane) * 64;
        const u32 base = lds0 + (issued & 1) * 32768u;
        for (int k = 0; k < 4; k++) { glds16(gl + 16 * k, base + 1024u * k); glds16(gr + 16 * k, base + 4096u + 1024u * k); }
        issued++; pf_i++; if (pf_i >= rows) { pf_i = 0; pf_jw += jstep; }
    };
    if (DEPTH >= 1) { if (issued < n_items) issue_next(); if (DEPTH >= 2 && issued < n_items) issue_next(); }
    for (size_t jw = jw0; jw < half; jw += jstep) {
        const size_t j = jw + lane;
        for (int i = 0; i < rows; i++, t++) {
            x4 v[8];
            if (DEPTH == 0) {
                const __attribute__((address_space(1))) x4* gl = (const __attribute__((address_space(1))) x4*)(L + (size_t)i * pitch + j * 64);
                const __attribute__((address_space(1))) x4* gr = (const __attribute__((address_space(1))) x4*)(R + (size_t)i * pitch + j * 64);
                for (int k = 0; k < 4; k++) { v[k] = gl[k]; v[4 + k] = gr[k]; }
            } else {
                if (issued > t + 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int k = 0; k < 8; k++) v[k] = stage[t & 1][wave][k][lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (issued < n_items) issue_next();
            }
            compute(a, v, steps);
        }
    }
    if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345u) out[threadIdx.x] = a[0];
}
int main(int argc, char** argv) {
    const size_t half = (size_t)1 << 19;
    const int rows = 25;
    const size_t bytes = (size_t)rows * half * 64;
    char *L, *R; u64* out;
    hipMalloc(&L, bytes); hipMalloc(&R, bytes); hipMalloc(&out, 4096);
    hipMemset(L, 1, bytes); hipMemset(R, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0}) for (int mode = 0; mode < 5; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: k_load<0><<<1024, 256>>>(L, R, half, rows, out, work); break;
                case 1: k_load<1><<<1024, 256>>>(L, R, half, rows, out, work); break;
                case 2: k_load<2><<<1024, 256>>>(L, R, half, rows, out, work); break;
                case 3: k_load<3><<<1024, 256>>>(L, R, half, rows, out, work); break;
                case 4: k_load<4><<<1024, 256>>>(L, R, half, rows, out, work); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("work %3d mode %d: %.3f ms  %.2f TB/s\n", work, mode, best, 2.0 * bytes / best / 1e9);
    }
    for (int steps : {0, 60, 120, 250}) for (int depth = 0; depth < 3; depth++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (depth == 0) k_pipe<0><<<1024, 256>>>(L, R, half, rows, out, steps);
            else if (depth == 1) k_pipe<1><<<1024, 256>>>(L, R, half, rows, out, steps);
            else k_pipe<2><<<1024, 256>>>(L, R, half, rows, out, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("steps %3d (%4d mad per item) prefetch depth %d: %.3f ms  %.2f TB/s\n", steps, steps * 4, depth, best, 2.0 * bytes / best / 1e9);
    }
    return 0;
}
