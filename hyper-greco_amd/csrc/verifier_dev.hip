// BfvEncrypt::verify with the table-sized work on the device [REF bfv-gkr/src/sk_encryption_circuit.rs:462-517; verify_gkr :509-510;
// lasso/src/memory_checking/verifier.rs:130-176]. The walk (verifier.cpp: proof parsing, round-polynomial checks, Lasso scalars) stays
// on the host; what scales with the tables goes through VerifyBackend (host.hpp) to the kernels the prover's bookkeeping uses:
//   eq tables of the claim points (runs of the challenge chain in HBM), the constant-gate sums, the wiring-predicate sums of the
//   Vanilla nodes as Libra gathers over the reverse CSR wiring followed by dot products with the eq table of the sum-check point,
//   the DFT-row tables of the FFT nodes, and the MLE evaluations of the public inputs.
// Everything is enqueued on one stream while the host keeps parsing; one synchronisation; then the deferred comparisons.
// Goldilocks, protocol mode 0 (the evaluation points are offsets into the fixed chain).
#include <cstring>
#include "prover.hpp"

namespace hg {
namespace {

__global__ __launch_bounds__(256) void k_dot_e2(const E2* __restrict__ a, const E2* __restrict__ b, size_t n, E2* __restrict__ partials) {
    __shared__ E2 sm[256];
    E2 acc = e2_zero();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc = e2_add(acc, e2_mul(a[i], b[i]));
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = e2_add(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sm[0];
}

struct DevBackend : VerifyBackend {
    hg_ctx* ctx;
    const hg_pk* pk;
    hipStream_t st;
    std::vector<const u64*> d_inputs;
    const u64* d_ct0is = nullptr;
    size_t res_used = 0;
    // the node being checked
    int node = -1;
    std::vector<size_t> mark;
    E2 *eqc = nullptr, *eqx = nullptr, *eqy = nullptr, *d_u = nullptr;
    dev::ClaimSet cs;

    DevBackend(hg_ctx* c, const hg_pk* k) : ctx(c), pk(k), st(c->stream) { memset(&cs, 0, sizeof(cs)); }
    int slot() {
        if (res_used + 1 > ctx->res_cap) throw Error("verifier: result buffer exhausted");
        return (int)res_used++;
    }
    template <typename T> T* upload(const T* src, size_t n) {
        const size_t bytes = n * sizeof(T), need = (bytes + 63) & ~(size_t)63;
        if (ctx->stage_used + need > ctx->stage_cap) {   // recycle the pinned staging: everything staged so far must have been copied
            hip_check(hipStreamSynchronize(st), "verifier: staging recycle");
            ctx->stage_used = 0;
        }
        void* h = ctx->h_stage + ctx->stage_used;
        ctx->stage_used += need;
        memcpy(h, src, bytes);
        T* d = ctx->alloc_n<T>(n ? n : 1);
        hip_check(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st), "verifier: upload descriptor");
        return d;
    }
    E2* eq_of(int nvars, const dev::ClaimSet& c) {
        const size_t N = (size_t)1 << nvars;
        E2* out = ctx->alloc_n<E2>(N);
        std::vector<dev::EqJob> jobs;
        E2* tmp = c.n > 1 ? ctx->alloc_n<E2>((size_t)c.n * N) : nullptr;
        for (int a = 0; a < c.n; a++) {
            dev::EqJob J;
            memset(&J, 0, sizeof(J));
            J.n = nvars; J.out = c.n > 1 ? tmp + (size_t)a * N : out;
            J.cs.n = 1; J.cs.unit_alpha = c.unit_alpha; J.cs.alpha_off = c.alpha_off + a; J.cs.point_off[0] = c.point_off[a];
            jobs.push_back(J);
        }
        dev::eq_jobs(st, upload(jobs.data(), jobs.size()), (int)jobs.size(), nvars, ctx->d_chal);
        if (c.n > 1) dev::sum_tables(st, out, tmp, c.n, N);
        return out;
    }
    E2* eq_single(int nvars, size_t off) {
        dev::ClaimSet c;
        memset(&c, 0, sizeof(c));
        c.n = 1; c.unit_alpha = 1; c.point_off[0] = off;
        return eq_of(nvars, c);
    }
    int dot_e2(const E2* a, const E2* b, size_t n) {
        const int grid = (int)std::min<size_t>((n + 255) / 256, 256);
        k_dot_e2<<<grid, 256, 0, st>>>(a, b, n, ctx->d_partials);
        const int t = slot();
        dev::reduce_partials(st, ctx->d_partials, grid, 1, ctx->d_res + t);
        return t;
    }

    void begin_node(int id, const ClaimOffs& cl) override {
        node = id;
        mark = ctx->arena_mark();
        const HNode& n = pk->circuit.nodes[id];
        if (cl.point_off.size() > (size_t)dev::MAX_CLAIMS) throw Error("verifier: too many claims on one node");
        memset(&cs, 0, sizeof(cs));
        cs.n = (int)cl.point_off.size();
        cs.unit_alpha = cl.unit ? 1 : 0;
        cs.alpha_off = cl.alpha_off;
        for (int a = 0; a < cs.n; a++) cs.point_off[a] = cl.point_off[a];
        eqc = n.kind == NK_VANILLA ? eq_of(n.log2_out(), cs) : nullptr;
        eqx = eqy = d_u = nullptr;
    }
    int const_sum() override {
        const HNode& n = pk->circuit.nodes[node];
        const hg_pk::NodeDev& nd = pk->node_dev[node];
        const int grid = dev::vanilla_const_sum(st, nd.const_gate, nd.const_coef, nd.nconst, eqc, n.log2_sub_out, n.log2_reps, ctx->d_partials);
        const int t = slot();
        dev::reduce_partials(st, ctx->d_partials, grid, 1, ctx->d_res + t);
        return t;
    }
    void set_x(size_t x_off) override {
        const HNode& n = pk->circuit.nodes[node];
        eqx = eq_single(n.kind == NK_VANILLA ? n.log2_sub_in + n.log2_reps : n.log2_size, x_off);
    }
    std::vector<int> lin_terms() override {
        const HNode& n = pk->circuit.nodes[node];
        const hg_pk::NodeDev& nd = pk->node_dev[node];
        const size_t SR = (size_t)1 << (n.log2_sub_in + n.log2_reps);
        std::vector<int> tk(n.arity, -1);
        for (int i = 0; i < n.arity; i++) {
            if (!n.left_use[i] || !nd.lin[i].ptr) continue;
            E2* T = ctx->alloc_n<E2>(SR);
            dev::GatherJob gj;
            memset(&gj, 0, sizeof(gj));
            gj.g.lin = nd.lin[i];   // (no mul part: the verifier's linear term has no input tables)
            gj.eqc = eqc; gj.log2_S = n.log2_sub_in; gj.log2_G = n.log2_sub_out; gj.log2_R = n.log2_reps; gj.T = T;
            dev::gather_jobs(st, upload(&gj, 1), 1, SR);
            tk[i] = dot_e2(T, eqx, SR);
        }
        return tk;
    }
    void set_y(size_t y_off, const std::vector<E2>& u) override {
        const HNode& n = pk->circuit.nodes[node];
        eqy = eq_single(n.log2_sub_in + n.log2_reps, y_off);
        d_u = upload(u.data(), u.size());
    }
    std::vector<int> mul_terms() override {
        const HNode& n = pk->circuit.nodes[node];
        const hg_pk::NodeDev& nd = pk->node_dev[node];
        const size_t SR = (size_t)1 << (n.log2_sub_in + n.log2_reps);
        std::vector<int> tk(n.arity, -1);
        for (int i = 0; i < n.arity; i++) {
            if (!n.right_use[i] || !nd.mulR[i].ptr) continue;
            E2* B = ctx->alloc_n<E2>(SR);
            dev::GatherBJob bj{nd.mulR[i], eqc, eqx, d_u, n.log2_sub_in, n.log2_sub_out, n.log2_reps, B};
            dev::gather_B_jobs(st, upload(&bj, 1), 1, SR);
            tk[i] = dot_e2(B, eqy, SR);
        }
        return tk;
    }
    int fft_term() override {
        const HNode& n = pk->circuit.nodes[node];
        const int L = n.log2_size;
        const size_t N = (size_t)1 << L;
        E2* F = ctx->alloc_n<E2>(N);
        const u64* W = (n.inverse ? pk->w_inv : pk->w_fwd).at(L);
        dev::FftJob fj{F, W, n.inverse ? gl_inv(gl_from_u64(N)) : 1, L, cs};
        E2* tab = ctx->alloc_n<E2>((size_t)cs.n * (N >> 4) + 1);
        dev::fft_jobs(st, upload(&fj, 1), 1, L, cs.n, ctx->d_chal, tab);
        return dot_e2(F, eqx, N);
    }
    void end_node() override {
        ctx->arena_rewind(mark);   // (one stream: the next node's kernels are ordered behind this node's)
        node = -1;
    }
    int mle_u64(const u64* tab, size_t point_off, int nvars) {
        const std::vector<size_t> m = ctx->arena_mark();
        E2* eq = eq_single(nvars, point_off);
        const int t = slot();
        const u64* tabs[8] = {tab};
        dev::dot_eq(st, eq, tabs, 1, (size_t)1 << nvars, ctx->d_partials, ctx->d_res + t);
        ctx->arena_rewind(m);
        return t;
    }
    int mle_input(size_t k, size_t point_off, int nvars) override {
        if (k >= d_inputs.size()) throw Error("verifier: no such input table");
        return mle_u64(d_inputs[k], point_off, nvars);
    }
    int mle_ct0is(size_t point_off, int nvars) override { return mle_u64(d_ct0is, point_off, nvars); }
    void finish() override {
        if (ctx->d_res != ctx->h_res && res_used)
            hip_check(hipMemcpyAsync(ctx->h_res, ctx->d_res, res_used * sizeof(E2), hipMemcpyDeviceToHost, st), "verifier: copy results");
        hip_check(hipStreamSynchronize(st), "verifier: synchronise");
        hip_check(hipGetLastError(), "verifier: kernels");
    }
    E2 value(int t) const override { return ctx->h_res[t]; }
};

}  // namespace

// public inputs and ct0is are uploaded (22 MB at n=32768 k=16), the proof is parsed on the host; "" = accepted
std::string verify_proof_device(hg_ctx* ctx, const hg_pk* pk, const Witness& w, const uint8_t* proof, size_t len) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    ctx->ensure_chain(16384);
    const Params& p = pk->params;
    DevBackend D(ctx, pk);
    const size_t SZ = p.SZ();
    auto up = [&](const u64* src, size_t n) {
        u64* d = ctx->alloc_n<u64>(n);
        hip_check(hipMemcpyAsync(d, src, n * 8, hipMemcpyHostToDevice, ctx->stream), "verifier: upload inputs");
        return (const u64*)d;
    };
    D.d_inputs.push_back(up(w.s.data(), SZ));
    D.d_inputs.push_back(up(w.e.data(), SZ));
    D.d_inputs.push_back(up(w.k1.data(), SZ));
    for (int i = 0; i < p.k; i++) D.d_inputs.push_back(up(&w.ais[(size_t)i * SZ], SZ));
    for (int i = 0; i < p.k; i++) D.d_inputs.push_back(up(&w.r1is[(size_t)i * SZ], SZ));
    D.d_inputs.push_back(up(w.r2is.data(), w.r2is.size()));
    D.d_ct0is = up(w.ct0is.data(), w.ct0is.size());
    // Rejection is a normal outcome and leaves kernels and staged descriptor copies queued (the walk returns from the middle of the
    // proof): drain the stream on EVERY way out - accept, Reject, hg::Error - before the caller may reuse the staging buffer and the
    // arena or free the witness whose uploads may still be pending.
    struct Drain {
        hipStream_t st;
        ~Drain() { (void)hipStreamSynchronize(st); }
    } drain{ctx->stream};
    return verify_proof_with(D, p, pk->lasso, pk->circuit, proof, len);
}

}  // namespace hg
