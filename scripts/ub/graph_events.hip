// Do timing events recorded inside a stream capture work when the graph is replayed? (ROCm 7.2, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; i++) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 1024));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    spin<<<1, 64, 0, s>>>(d, 1000);
    CK(hipEventRecordWithFlags(a, s, hipEventRecordExternal));
    spin<<<1, 64, 0, s>>>(d, 2000000);
    CK(hipEventRecordWithFlags(b, s, hipEventRecordExternal));
    spin<<<1, 64, 0, s>>>(d, 1000);
    CK(hipStreamEndCapture(s, &g));
    size_t n = 0; CK(hipGraphGetNodes(g, nullptr, &n)); printf("graph nodes: %zu\n", n);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 3; it++) {
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        float ms = -1; hipError_t e = hipEventElapsedTime(&ms, a, b);
        printf("replay %d: elapsed(a,b) = %f ms (%s)\n", it, ms, hipGetErrorString(e));
    }
    return 0;
}
