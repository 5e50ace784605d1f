// Multi-GPU exchange inside the library: one RCCL all-reduce per proof, device buffers only (no host staging, no framework).
//
// A sharded proof (prover.hip: plan_shards) leaves on every rank a result buffer of canonical Goldilocks lanes that is zero where
// the rank owns nothing and a PARTIAL sum where a sum-check was split by memory. The ranks' buffers must be added lane-wise
// mod p. RCCL has no field addition, so each 64-bit lane travels as two 32-bit halves in 64-bit lanes (ncclSum cannot overflow
// below 2^32 ranks) and is folded back mod p after the collective.
// RCCL is loaded at run time (dlopen) the first time a communicator is asked for: the single-GPU product has no link-time
// dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include "prover.hpp"

namespace hg {

namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.so, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.so, "ncclCommCount"));
    });
    if (!r.so || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce)
        throw Error("RCCL is not available (librccl.so could not be loaded): multi-GPU proving needs it");
    return r;
}
void nccl_check(ncclResult_t rc, const char* what) {
    if (rc != ncclSuccess) {
        Rccl& r = rccl();
        throw Error(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
    }
}

__global__ void k_split_limbs(const u64* __restrict__ res, size_t n, u64* __restrict__ x) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 v = res[i];
    x[2 * i] = v & 0xFFFFFFFFull;
    x[2 * i + 1] = v >> 32;
}
// lane = (sum lo) + 2^32 (sum hi) mod p; both sums are below 2^32 * world
__global__ void k_combine_limbs(const u64* __restrict__ x, size_t n, u64* __restrict__ res) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 lo = gl_from_u64(x[2 * i]), hi = gl_from_u64(x[2 * i + 1]);
    res[i] = gl_add(lo, gl_mul(hi, 1ull << 32));
}
// what ncclSum does to the limb lanes of two ranks (selftest only)
__global__ void k_add_lanes(u64* __restrict__ acc, const u64* __restrict__ x, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += x[i];
}
}  // namespace

// The exchange arithmetic without a communicator: `world` rank buffers of n canonical lanes each (host, rank-major) go through
// k_split_limbs, are added lane-wise as plain 64-bit integers (ncclSum on ncclUint64) and folded back by k_combine_limbs -> out[n].
// A one-GPU box cannot run a communicator of more than one rank, so this is how the world > 1 branch of the fold-back (limb sums
// up to world * (2^32 - 1)) is tested on hardware.
void comm_selftest(hg_ctx* ctx, const u64* bufs, int world, size_t n, u64* out) {
    if (world < 1 || n == 0) throw Error("hg_comm_selftest: bad argument");
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    u64 *d_in = nullptr, *d_x = nullptr, *d_acc = nullptr;
    hip_check(hipMalloc((void**)&d_in, n * 8), "hipMalloc");
    hip_check(hipMalloc((void**)&d_x, 2 * n * 8), "hipMalloc");
    hip_check(hipMalloc((void**)&d_acc, 2 * n * 8), "hipMalloc");
    hip_check(hipMemsetAsync(d_acc, 0, 2 * n * 8, ctx->stream), "memset");
    const unsigned grid = (unsigned)((n + 255) / 256), grid2 = (unsigned)((2 * n + 255) / 256);
    for (int r = 0; r < world; r++) {
        hip_check(hipMemcpyAsync(d_in, bufs + (size_t)r * n, n * 8, hipMemcpyHostToDevice, ctx->stream), "upload");
        k_split_limbs<<<grid, 256, 0, ctx->stream>>>(d_in, n, d_x);
        k_add_lanes<<<grid2, 256, 0, ctx->stream>>>(d_acc, d_x, 2 * n);
    }
    k_combine_limbs<<<grid, 256, 0, ctx->stream>>>(d_acc, n, d_in);
    hip_check(hipMemcpyAsync(out, d_in, n * 8, hipMemcpyDeviceToHost, ctx->stream), "download");
    hipError_t e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in); (void)hipFree(d_x); (void)hipFree(d_acc);
    hip_check(e, "hg_comm_selftest");
}
int comm_count(hg_ctx* ctx) {
    if (!ctx->comm) return 0;
    Rccl& r = rccl();
    if (!r.CommCount) return ctx->comm_world;
    int n = 0;
    nccl_check(r.CommCount(static_cast<ncclComm_t>(ctx->comm), &n), "ncclCommCount");
    return n;
}

void comm_unique_id(uint8_t out[128]) {
    ncclUniqueId id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out, &id, 128);
}
void comm_init(hg_ctx* ctx, const uint8_t id_bytes[128], int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) throw Error("hg_comm_init: bad rank / world");
    if (ctx->comm) throw Error("hg_comm_init: this context already has a communicator");
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ncclUniqueId id;
    memcpy(&id, id_bytes, 128);
    ncclComm_t comm = nullptr;
    nccl_check(rccl().CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
}
void comm_destroy(hg_ctx* ctx) {
    if (!ctx->comm) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)rccl().CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    if (ctx->d_xchg) { (void)hipFree(ctx->d_xchg); ctx->d_xchg = nullptr; ctx->xchg_cap = 0; }
}
// Enqueued on ctx->stream after this rank's last kernel: lane-wise sum mod p of the first n_e2 result slots over all ranks,
// written back in place (the result buffer is host-mapped: the combined lanes land where the transcript replay reads them).
void comm_allreduce_results(hg_ctx* ctx, size_t n_e2) {
    if (!ctx->comm) throw Error("sharded prove: no communicator (hg_comm_init)");
    const size_t n = 2 * n_e2;  // u64 lanes
    if (ctx->xchg_cap < 2 * n) {
        if (ctx->d_xchg) (void)hipFree(ctx->d_xchg);
        ctx->xchg_cap = 2 * std::max<size_t>(n, 2 * ctx->res_cap);
        hip_check(hipMalloc((void**)&ctx->d_xchg, ctx->xchg_cap * sizeof(u64)), "hipMalloc(exchange buffer)");
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    u64* lanes = reinterpret_cast<u64*>(ctx->d_res);
    k_split_limbs<<<grid, 256, 0, ctx->stream>>>(lanes, n, ctx->d_xchg);
    nccl_check(rccl().AllReduce(ctx->d_xchg, ctx->d_xchg, 2 * n, ncclUint64, ncclSum, static_cast<ncclComm_t>(ctx->comm), ctx->stream), "ncclAllReduce");
    k_combine_limbs<<<grid, 256, 0, ctx->stream>>>(ctx->d_xchg, n, lanes);
}

}  // namespace hg
