// Keccak-f[1600] on ONE wave of a gfx950 CU: how long does a dependent chain of permutations take? (The absorbing transcript needs
// two permutations per extension-field challenge, strictly one after the other, 1800 challenges per proof at n=32768 k=16: the
// question is whether squeezing on the device can beat the 3.4 us mailbox round trip to a host core that does a permutation in 0.35 us.)
//  (a) uniform state: the compiler keeps the 25 lanes in SGPR pairs (s_xor_b64 / s_andn2_b64 / s_lshl_b64 ...), one wave issues at
//      most one instruction every 4-5 cycles;
//  (b) one VALU lane holding the whole state (what "a single thread does the transcript" compiles to when the state is divergent);
//  (c) 25 lanes, one state lane each, theta / pi / chi through ds_bpermute.
// Every variant is checked against the host implementation (Keccak-f of the zero state, iterated).
// build: hipcc -O3 --offload-arch=gfx950 scripts/ub/keccakbench.hip -o scripts/ub/keccakbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
typedef unsigned long long u64;
__host__ __device__ constexpr u64 RC(int i) {
    constexpr u64 rc[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                            0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                            0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                            0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    return rc[i];
}
__host__ __device__ constexpr int ROT(int i) {   // rho offsets, index x + 5y
    constexpr int r[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    return r[i];
}
__host__ __device__ inline u64 rotl(u64 v, int s) { return s ? (v << s) | (v >> (64 - s)) : v; }
// straight-line permutation on 25 named values (fully unrolled so that every lane stays in a register)
__host__ __device__ inline void keccak_f(u64* a) {
#pragma unroll
    for (int rd = 0; rd < 24; rd++) {
        u64 c[5], d[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
#pragma unroll
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], ROT(x + 5 * y));
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC(rd);
    }
}
// (a) uniform: the seed is a kernel argument, nothing depends on the lane
__global__ void k_uniform(u64 seed, int n, u64* out, long long* clk) {
    u64 a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = 0;
    a[0] = seed;
    const long long t0 = wall_clock64();
    for (int it = 0; it < n; it++) keccak_f(a);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { for (int i = 0; i < 25; i++) out[i] = a[i]; *clk = t1 - t0; }
}
// (b) divergent: the seed comes from a per-lane load
__global__ void k_lane(const u64* seeds, int n, u64* out, long long* clk) {
    u64 a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = 0;
    a[0] = seeds[threadIdx.x];
    const long long t0 = wall_clock64();
    for (int it = 0; it < n; it++) keccak_f(a);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { for (int i = 0; i < 25; i++) out[i] = a[i]; *clk = t1 - t0; }
}
// (c) lane i = x + 5y holds A[x][y]
__device__ __forceinline__ u64 bperm(int src_lane, u64 v) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane * 4, (int)(unsigned)v);
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane * 4, (int)(unsigned)(v >> 32));
    return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
__global__ void k_wave25(u64 seed, int n, u64* out, long long* clk, const int* rot_tab, const int* pi_src) {
    const int i = threadIdx.x < 25 ? threadIdx.x : 0;
    const int x = i % 5, y = i / 5;
    u64 a = threadIdx.x == 0 ? seed : 0;
    const int rot = rot_tab[i];          // rho offset of MY lane's value
    const int src = pi_src[i];           // lane whose rotated value lands here
    const int l_xm1 = (x + 4) % 5, l_xp1 = (x + 1) % 5;                       // column sums live in lanes 0..4
    const int l_c1 = (x + 1) % 5 + 5 * y, l_c2 = (x + 2) % 5 + 5 * y;
    const long long t0 = wall_clock64();
    for (int it = 0; it < n; it++) {
        for (int rd = 0; rd < 24; rd++) {
            // theta: c[x] = xor over the column; every lane fetches the four other rows of its column
            u64 c = a ^ bperm(x + 5 * ((y + 1) % 5), a) ^ bperm(x + 5 * ((y + 2) % 5), a) ^ bperm(x + 5 * ((y + 3) % 5), a) ^ bperm(x + 5 * ((y + 4) % 5), a);
            const u64 d = bperm(l_xm1, c) ^ rotl(bperm(l_xp1, c), 1);         // (row 0's lanes hold c[x]; every row holds the same)
            a ^= d;
            // rho + pi
            const u64 r = rot ? (a << rot) | (a >> (64 - rot)) : a;
            const u64 b = bperm(src, r);
            // chi
            a = b ^ (~bperm(l_c1, b) & bperm(l_c2, b));
            if (threadIdx.x == 0) a ^= RC(rd);
        }
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x < 25) out[threadIdx.x] = a;
    if (threadIdx.x == 0) *clk = t1 - t0;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 512;
    u64 ref[25];
    memset(ref, 0, sizeof(ref));
    ref[0] = 0x1234567ull;
    double h0 = now();
    for (int i = 0; i < N; i++) keccak_f(ref);
    double h1 = now();
    printf("host core: %.3f us per permutation\n", (h1 - h0) / N);
    u64 *d_out, *d_seeds; long long* d_clk; int *d_rot, *d_src;
    hipMalloc((void**)&d_out, 25 * 8); hipMalloc((void**)&d_seeds, 64 * 8); hipMalloc((void**)&d_clk, 8); hipMalloc((void**)&d_rot, 100); hipMalloc((void**)&d_src, 100);
    u64 seeds[64]; for (auto& s : seeds) s = 0x1234567ull;
    hipMemcpy(d_seeds, seeds, sizeof(seeds), hipMemcpyHostToDevice);
    int rot[25], src[25];
    for (int i = 0; i < 25; i++) rot[i] = ROT(i);
    for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) src[y + 5 * ((2 * x + 3 * y) % 5)] = x + 5 * y;
    hipMemcpy(d_rot, rot, sizeof(rot), hipMemcpyHostToDevice); hipMemcpy(d_src, src, sizeof(src), hipMemcpyHostToDevice);
    auto report = [&](const char* what) {
        hipDeviceSynchronize();
        u64 got[25]; long long clk;
        hipMemcpy(got, d_out, sizeof(got), hipMemcpyDeviceToHost); hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost);
        printf("%-46s %7.3f us per permutation (100 MHz device clock)  %s\n", what, clk / 100.0 / N, memcmp(got, ref, sizeof(got)) == 0 ? "state ok" : "STATE DIFFERS");
    };
    for (int rep = 0; rep < 2; rep++) {
        k_uniform<<<1, 64>>>(0x1234567ull, N, d_out, d_clk); report("(a) uniform state (scalar unit), one wave:");
        k_lane<<<1, 64>>>(d_seeds, N, d_out, d_clk); report("(b) whole state per VALU lane, one wave:");
        k_wave25<<<1, 64>>>(0x1234567ull, N, d_out, d_clk, d_rot, d_src); report("(c) 25 lanes + ds_bpermute, one wave:");
    }
    return 0;
}
