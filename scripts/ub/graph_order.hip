// When does a graph node with a cross-stream dependency start on replay? (ROCm 7.2, gfx950)
// Stream capture of two chains: A0..A{na-1} on s1; B forks after A1 and runs B0..B{nb-1} on s2; X on s1 depends on A{na-1} and
// B{dep}; X is followed by a few more A nodes; s2 joins at the end. Every kernel stamps wall_clock64() when it starts.
// mode 0: X and the A tail are captured AFTER all of B (the prover's order); mode 1: they are captured right after B{dep}.
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
    const int spin = argc > 3 ? atoi(argv[3]) : 2000; const int mode = argc > 1 ? atoi(argv[1]) : 0, na = 9, nb = argc > 2 ? atoi(argv[2]) : 60, dep = 13, ntail = 5;
    unsigned long long* d; CK(hipMalloc(&d, 4096 * 8));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t fork, evdep, join; CK(hipEventCreate(&fork)); CK(hipEventCreate(&evdep)); CK(hipEventCreate(&join));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    int slot = 0;
    std::vector<const char*> names;
    const int spin_a = argc > 4 ? atoi(argv[4]) : spin;
    auto K = [&](hipStream_t s, const char* nm) { stamp<<<1, 64, 0, s>>>(d, slot++, s == s1 ? spin_a : spin); names.push_back(nm); };
    K(s1, "A"); K(s1, "A");
    CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0));
    for (int i = 2; i < na; i++) K(s1, "A");
    auto tail = [&] { (void)hipStreamWaitEvent(s1, evdep, 0); K(s1, "X"); for (int i = 0; i < ntail; i++) K(s1, "a"); };
    for (int i = 0; i < nb; i++) {
        K(s2, "B");
        if (i == dep) { CK(hipEventRecord(evdep, s2)); if (mode == 1) tail(); }
    }
    if (mode == 0) tail();
    CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s1, join, 0));
    K(s1, "J");
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<unsigned long long> h(slot);
    for (int it = 0; it < 4; it++) {
        CK(hipGraphLaunch(ge, s1));
        CK(hipStreamSynchronize(s1));
    }
    CK(hipMemcpy(h.data(), d, slot * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = h[0];
    printf("mode %d: start times in us (100 MHz clock), in capture order:\n", mode);
    for (int i = 0; i < slot; i++) printf("%s%d@%.1f ", names[i], i, (h[i] - t0) / 100.0);
    printf("\n");
    return 0;
}
