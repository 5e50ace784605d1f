// Does a replayed stream-captured graph run THREE forked branches concurrently? (ROCm 7.2, gfx950)
// main: M0 M1 [fork] M2..M9 (long kernels) [join] J; branch B on s2 and branch C on s3: nb kernels each (medium), both forked after M1.
// mode 0: B and C depend on the fork only; mode 1: B's first node additionally depends on M2 (like the prover's counters on the limb kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void stamp(unsigned long long* t, int slot, int spin) {
    if (threadIdx.x == 0) t[slot] = wall_clock64();
    float v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) t[slot] = 0;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, nb = 20;
    unsigned long long* d; CK(hipMalloc(&d, 4096 * 8));
    hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    hipEvent_t fork, m2, j2, j3; CK(hipEventCreate(&fork)); CK(hipEventCreate(&m2)); CK(hipEventCreate(&j2)); CK(hipEventCreate(&j3));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    int slot = 0;
    std::vector<const char*> names;
    auto K = [&](hipStream_t s, const char* nm, int spin) { stamp<<<1, 64, 0, s>>>(d, slot++, spin); names.push_back(nm); };
    K(s1, "M", 2000); K(s1, "M", 2000);
    CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0)); CK(hipStreamWaitEvent(s3, fork, 0));
    K(s1, "M", 2000);
    CK(hipEventRecord(m2, s1));
    if (mode == 1) CK(hipStreamWaitEvent(s2, m2, 0));
    for (int i = 0; i < 7; i++) K(s1, "M", 2000);
    for (int i = 0; i < nb; i++) K(s2, "B", 700);
    for (int i = 0; i < nb; i++) K(s3, "C", 700);
    CK(hipEventRecord(j2, s2)); CK(hipEventRecord(j3, s3)); CK(hipStreamWaitEvent(s1, j2, 0)); CK(hipStreamWaitEvent(s1, j3, 0));
    K(s1, "J", 10);
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<unsigned long long> h(slot);
    for (int it = 0; it < 4; it++) { CK(hipGraphLaunch(ge, s1)); CK(hipStreamSynchronize(s1)); }
    CK(hipMemcpy(h.data(), d, slot * 8, hipMemcpyDeviceToHost));
    printf("mode %d: ", mode);
    for (int i = 0; i < slot; i++) printf("%s%d@%.0f ", names[i], i, (double)(long long)(h[i] - h[0]) / 100.0);
    printf("\n");
    return 0;
}
