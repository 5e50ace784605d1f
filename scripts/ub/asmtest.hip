#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define main bench_main
#include "asmbench.hip"
#undef main
__global__ void kt(const u64* a, const u64* b, u64* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o[i] = add_asm(a[i], b[i]); o[n + i] = sub_asm(a[i], b[i]); o[2 * n + i] = mul_asm(a[i], b[i]);
    o[3 * n + i] = gl_add(a[i], b[i]); o[4 * n + i] = gl_sub(a[i], b[i]); o[5 * n + i] = gl_mul(a[i], b[i]);
}
int main() {
    int n = 1 << 16;
    std::vector<u64> a(n), b(n);
    u64 x = 88172645463325252ULL;
    for (int i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; a[i] = x % GL_P; x ^= x << 13; x ^= x >> 7; x ^= x << 17; b[i] = x % GL_P; }
    u64 edge[] = {0, 1, GL_P - 1, GL_P - 2, 0xFFFFFFFFULL, 0x100000000ULL, 0xFFFFFFFF00000000ULL};
    for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) { a[i * 7 + j] = edge[i]; b[i * 7 + j] = edge[j]; }
    u64 *da, *db, *dout; hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    kt<<<n / 256, 256>>>(da, db, dout, n);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    std::vector<u64> o(6 * n); hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    const char* nm[] = {"add", "sub", "mul"};
    for (int op = 0; op < 3; op++) {
        int bad = 0, first = -1;
        for (int i = 0; i < n; i++) if (o[op * n + i] != o[(3 + op) * n + i]) { if (first < 0) first = i; bad++; }
        printf("%s: %d mismatches", nm[op], bad);
        if (first >= 0) printf(" first i=%d a=%llx b=%llx asm=%llx ref=%llx", first, (unsigned long long)a[first], (unsigned long long)b[first], (unsigned long long)o[op * n + first], (unsigned long long)o[(3 + op) * n + first]);
        printf("\n");
    }
}
