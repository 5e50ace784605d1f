// Mailbox round-trip microbenchmark (prover_seq.hip): device posts a sequence number into pinned host memory and spins on the host's answer.
//  (a) one persistent one-thread kernel doing N round trips: the pure PCIe round trip;
//  (b) one kernel per round trip, all enqueued up front: adds the dependent-launch gap.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ub/mailbench.hip -o scripts/ub/mailbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct Mail { unsigned long long gpu_seq, cpu_seq, pad0, pad1; unsigned long long chal[2]; };
__global__ void k_persist(Mail* m, int n, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (int i = 1; i <= n; i++) {
        __hip_atomic_store(&m->gpu_seq, (unsigned long long)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        while (__hip_atomic_load(&m->cpu_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)i) __builtin_amdgcn_s_sleep(1);
        acc += __hip_atomic_load(&m->chal[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *sink = acc;
}
__global__ void k_one(Mail* m, unsigned long long seq, unsigned long long* sink) {
    __hip_atomic_store(&m->gpu_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    while (__hip_atomic_load(&m->cpu_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) __builtin_amdgcn_s_sleep(1);
    *sink += __hip_atomic_load(&m->chal[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void serve(Mail* m, int n, unsigned long long base) {
    for (int i = 1; i <= n; i++) {
        while (__atomic_load_n(&m->gpu_seq, __ATOMIC_ACQUIRE) < base + i) __builtin_ia32_pause();
        m->chal[0] = i;
        __atomic_store_n(&m->cpu_seq, base + i, __ATOMIC_RELEASE);
    }
}
int main() {
    Mail* m; unsigned long long* sink;
    hipHostMalloc((void**)&m, sizeof(Mail), hipHostMallocDefault); memset(m, 0, sizeof(Mail));
    hipMalloc((void**)&sink, 8); hipMemset(sink, 0, 8);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int N = 2000;
    for (int rep = 0; rep < 3; rep++) {
        memset(m, 0, sizeof(Mail));
        double t0 = now();
        k_persist<<<1, 1, 0, st>>>(m, N, sink);
        serve(m, N, 0);
        hipStreamSynchronize(st);
        printf("persistent kernel: %.2f us per round trip\n", (now() - t0) / N);
    }
    for (int rep = 0; rep < 3; rep++) {
        memset(m, 0, sizeof(Mail));
        double t0 = now();
        for (int i = 1; i <= N; i++) k_one<<<1, 64, 0, st>>>(m, i, sink);
        double t1 = now();
        serve(m, N, 0);
        hipStreamSynchronize(st);
        printf("one kernel per round trip: %.2f us each (enqueue of %d launches took %.2f us each)\n", (now() - t1) / N, N, (t1 - t0) / N);
    }
    // sync-based round trip for comparison: launch, hipStreamSynchronize
    {
        double t0 = now();
        for (int i = 1; i <= 500; i++) { m->cpu_seq = ~0ull; k_one<<<1, 64, 0, st>>>(m, 1, sink); hipStreamSynchronize(st); }
        printf("launch + hipStreamSynchronize: %.2f us each\n", (now() - t0) / 500);
    }
    return 0;
}
