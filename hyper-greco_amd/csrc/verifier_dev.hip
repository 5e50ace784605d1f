// BfvEncrypt::verify with the table-sized work on the device [REF bfv-gkr/src/sk_encryption_circuit.rs:462-517; verify_gkr :509-510;
// lasso/src/memory_checking/verifier.rs:130-176]. The walk (verifier.cpp: proof parsing, round-polynomial checks, Lasso scalars) stays
// on the host; what scales with the tables goes through VerifyBackend (host.hpp) to the kernels the prover's bookkeeping uses:
//   eq tables of the claim points (runs of the challenge chain in HBM), the constant-gate sums, the wiring-predicate sums of the
//   Vanilla nodes as Libra gathers over the reverse CSR wiring followed by dot products with the eq table of the sum-check point,
//   the DFT-row tables of the FFT nodes, and the MLE evaluations of the public inputs.
// Everything is enqueued on one stream while the host keeps parsing; one synchronisation; then the deferred comparisons.
// Goldilocks, protocol mode 0 (the evaluation points are offsets into the fixed chain).
#include <cstring>
#include <omp.h>
#include "prover.hpp"

namespace hg {
namespace {

// every dot product of the verification in two launches: job q = sum_i a_q[i] * b_q[i] (a: E2 table or a table of base-field
// integers, b: an eq table), VD_BLOCKS workgroups per job, then one workgroup per job adds their partial sums into the job's slot
constexpr int VD_BLOCKS = 32;
struct DotJob { const void* a; const E2* b; size_t n; int slot; int a_is_u64; };
__global__ __launch_bounds__(256) void k_vdot_jobs(const DotJob* __restrict__ jobs, E2* __restrict__ partials) {
    __shared__ E2 sm[256];
    const DotJob& J = jobs[blockIdx.y];
    E2 acc = e2_zero();
    if (J.a_is_u64) {
        const u64* a = static_cast<const u64*>(J.a);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < J.n; i += (size_t)VD_BLOCKS * 256) { const u64 v = a[i]; if (v) acc = e2_add(acc, e2_mul_f(J.b[i], v)); }
    } else {
        const E2* a = static_cast<const E2*>(J.a);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < J.n; i += (size_t)VD_BLOCKS * 256) acc = e2_add(acc, e2_mul(a[i], J.b[i]));
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = e2_add(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * VD_BLOCKS + blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(64) void k_vdot_reduce(const DotJob* __restrict__ jobs, const E2* __restrict__ partials, E2* __restrict__ res) {
    __shared__ E2 sm[64];
    sm[threadIdx.x] = threadIdx.x < VD_BLOCKS ? partials[(size_t)blockIdx.x * VD_BLOCKS + threadIdx.x] : e2_zero();
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = e2_add(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) res[jobs[blockIdx.x].slot] = sm[0];
}

// The walk (verifier.cpp) only RECORDS what it needs - every evaluation point is a run of the challenge chain, so no table depends
// on a value the host would have to read back - and finish() launches it by kind over job arrays, as the prover's bookkeeping does:
// all eq tables (a node's claims combined inside the kernel), the constant-gate sums, all Libra gathers, all DFT-row tables, all
// phase-2 gathers, all dot products. Round 3 launched per node and per input as the walk went: ~700 launches, 17.6 ms at n=32768 k=16.
struct DevBackend : VerifyBackend {
    hg_ctx* ctx;
    const hg_pk* pk;
    hipStream_t st;
    std::vector<const u64*> d_inputs;
    const u64* d_ct0is = nullptr;
    size_t res_used = 0;
    // the node being checked
    int node = -1;
    E2 *eqc = nullptr, *eqx = nullptr, *eqy = nullptr;
    const E2* d_u = nullptr;
    dev::ClaimSet cs;
    // recorded work
    std::vector<dev::EqJob> eqs;
    int eq_max_n = 0;
    struct ConstSum { const hg_pk::NodeDev* nd; const E2* eqc; int log2_G, log2_R, slot; };
    std::vector<ConstSum> consts;
    std::vector<dev::GatherJob> gts;
    std::vector<dev::GatherBJob> gbs;
    size_t gt_max = 0, gb_max = 0;
    std::vector<dev::FftJob> ffts;
    int fft_max_L = 0, fft_max_claims = 0;
    std::vector<DotJob> dots;
    std::vector<E2> h_u;          // the phase-1 evaluations of every Vanilla node with a phase 2, back to back
    E2* d_u_all = nullptr;
    static constexpr size_t U_CAP = 8192;

    DevBackend(hg_ctx* c, const hg_pk* k) : ctx(c), pk(k), st(c->stream) {
        memset(&cs, 0, sizeof(cs));
        d_u_all = ctx->alloc_n<E2>(U_CAP);
    }
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
        E2* out = ctx->alloc_n<E2>((size_t)1 << nvars);
        dev::EqJob J;
        memset(&J, 0, sizeof(J));
        J.n = nvars; J.out = out; J.cs = c;
        eqs.push_back(J);
        eq_max_n = std::max(eq_max_n, nvars);
        return out;
    }
    E2* eq_single(int nvars, size_t off) {
        dev::ClaimSet c;
        memset(&c, 0, sizeof(c));
        c.n = 1; c.unit_alpha = 1; c.point_off[0] = off;
        return eq_of(nvars, c);
    }
    int dot(const void* a, bool a_is_u64, const E2* b, size_t n) {
        const int t = slot();
        dots.push_back(DotJob{a, b, n, t, a_is_u64 ? 1 : 0});
        return t;
    }

    void begin_node(int id, const ClaimOffs& cl) override {
        node = id;
        const HNode& n = pk->circuit.nodes[id];
        if (cl.point_off.size() > (size_t)dev::MAX_CLAIMS) throw Error("verifier: too many claims on one node");
        memset(&cs, 0, sizeof(cs));
        cs.n = (int)cl.point_off.size();
        cs.unit_alpha = cl.unit ? 1 : 0;
        cs.alpha_off = cl.alpha_off;
        for (int a = 0; a < cs.n; a++) cs.point_off[a] = cl.point_off[a];
        eqc = n.kind == NK_VANILLA ? eq_of(n.log2_out(), cs) : nullptr;
        eqx = eqy = nullptr;
        d_u = nullptr;
    }
    int const_sum() override {
        const HNode& n = pk->circuit.nodes[node];
        const int t = slot();
        consts.push_back(ConstSum{&pk->node_dev[node], eqc, n.log2_sub_out, n.log2_reps, t});
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
            gts.push_back(gj);
            gt_max = std::max(gt_max, SR);
            tk[i] = dot(T, false, eqx, SR);
        }
        return tk;
    }
    void set_y(size_t y_off, const std::vector<E2>& u) override {
        const HNode& n = pk->circuit.nodes[node];
        eqy = eq_single(n.log2_sub_in + n.log2_reps, y_off);
        if (h_u.size() + u.size() > U_CAP) throw Error("verifier: too many phase-1 evaluations");
        d_u = d_u_all + h_u.size();
        h_u.insert(h_u.end(), u.begin(), u.end());
    }
    std::vector<int> mul_terms() override {
        const HNode& n = pk->circuit.nodes[node];
        const hg_pk::NodeDev& nd = pk->node_dev[node];
        const size_t SR = (size_t)1 << (n.log2_sub_in + n.log2_reps);
        std::vector<int> tk(n.arity, -1);
        for (int i = 0; i < n.arity; i++) {
            if (!n.right_use[i] || !nd.mulR[i].ptr) continue;
            E2* B = ctx->alloc_n<E2>(SR);
            gbs.push_back(dev::GatherBJob{nd.mulR[i], eqc, eqx, d_u, n.log2_sub_in, n.log2_sub_out, n.log2_reps, B});
            gb_max = std::max(gb_max, SR);
            tk[i] = dot(B, false, eqy, SR);
        }
        return tk;
    }
    int fft_term() override {
        const HNode& n = pk->circuit.nodes[node];
        const int L = n.log2_size;
        const size_t N = (size_t)1 << L;
        E2* F = ctx->alloc_n<E2>(N);
        const u64* W = (n.inverse ? pk->w_inv : pk->w_fwd).at(L);
        ffts.push_back(dev::FftJob{F, W, n.inverse ? gl_inv(gl_from_u64(N)) : 1, L, cs});
        fft_max_L = std::max(fft_max_L, L);
        fft_max_claims = std::max(fft_max_claims, cs.n);
        return dot(F, false, eqx, N);
    }
    void end_node() override { node = -1; }
    int mle_u64(const u64* tab, size_t point_off, int nvars) { return dot(tab, true, eq_single(nvars, point_off), (size_t)1 << nvars); }
    int mle_input(size_t k, size_t point_off, int nvars) override {
        if (k >= d_inputs.size()) throw Error("verifier: no such input table");
        return mle_u64(d_inputs[k], point_off, nvars);
    }
    int mle_ct0is(size_t point_off, int nvars) override { return mle_u64(d_ct0is, point_off, nvars); }
    void finish() override {
        const bool times = hg_times("verify");   // read at every call (host.hpp)
        const double t0 = times ? omp_get_wtime() : 0;
        if (times) { hip_check(hipStreamSynchronize(st), "sync"); fprintf(stderr, "[hg] verify_device: uploads drained %.2f ms after the walk ended; %zu eq tables, %zu gathers, %zu + %zu, %zu dots\n", (omp_get_wtime() - t0) * 1e3, eqs.size(), gts.size(), gbs.size(), ffts.size(), dots.size()); }
        auto lap = [&](const char* what) { if (times) { hip_check(hipStreamSynchronize(st), "sync"); fprintf(stderr, "[hg] verify_device: %8.2f ms  %s\n", (omp_get_wtime() - t0) * 1e3, what); } };
        // descriptors first (one staging area; the copies are stream-ordered ahead of the launches), then one launch per kind
        const dev::EqJob* d_eqs = eqs.empty() ? nullptr : upload(eqs.data(), eqs.size());
        const dev::GatherJob* d_gts = gts.empty() ? nullptr : upload(gts.data(), gts.size());
        const dev::GatherBJob* d_gbs = gbs.empty() ? nullptr : upload(gbs.data(), gbs.size());
        const dev::FftJob* d_ffts = ffts.empty() ? nullptr : upload(ffts.data(), ffts.size());
        const DotJob* d_dots = dots.empty() ? nullptr : upload(dots.data(), dots.size());
        if (!h_u.empty()) {
            const E2* staged = upload(h_u.data(), h_u.size());
            hip_check(hipMemcpyAsync(d_u_all, staged, h_u.size() * sizeof(E2), hipMemcpyDeviceToDevice, st), "verifier: phase-1 evaluations");
        }
        lap("descriptors uploaded");
        if (d_eqs) dev::eq_jobs(st, d_eqs, (int)eqs.size(), eq_max_n, ctx->d_chal);
        lap("eq tables");
        for (const ConstSum& c : consts) {
            const int grid = dev::vanilla_const_sum(st, c.nd->const_gate, c.nd->const_coef, c.nd->nconst, c.eqc, c.log2_G, c.log2_R, ctx->d_partials);
            dev::reduce_partials(st, ctx->d_partials, grid, 1, ctx->d_res + c.slot);
        }
        lap("constant sums");
        if (d_gts) dev::gather_jobs(st, d_gts, (int)gts.size(), gt_max);
        lap("phase-1 gathers");
        if (d_ffts) {
            E2* tab = ctx->alloc_n<E2>(ffts.size() * (size_t)fft_max_claims * ((size_t)1 << (fft_max_L > 4 ? fft_max_L - 4 : 0)) + 1);
            dev::fft_jobs(st, d_ffts, (int)ffts.size(), fft_max_L, fft_max_claims, ctx->d_chal, tab);
        }
        lap("DFT-row tables");
        if (d_gbs) dev::gather_B_jobs(st, d_gbs, (int)gbs.size(), gb_max);
        lap("phase-2 gathers");
        if (d_dots) {
            E2* part = ctx->alloc_n<E2>(dots.size() * (size_t)VD_BLOCKS);
            k_vdot_jobs<<<dim3(VD_BLOCKS, (unsigned)dots.size()), 256, 0, st>>>(d_dots, part);
            k_vdot_reduce<<<(unsigned)dots.size(), 64, 0, st>>>(d_dots, part, ctx->d_res);
        }
        lap("dot products");
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
    const double tv0 = omp_get_wtime();
    struct Total { double t0; ~Total() { if (hg_times("verify")) fprintf(stderr, "[hg] verify_device: %.2f ms in all\n", (omp_get_wtime() - t0) * 1e3); } } total{tv0};
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    // Rejection is a normal outcome and leaves kernels and staged descriptor copies queued (the walk returns from the middle of the
    // proof), and an hg::Error may leave from any of the uploads below: drain the stream on EVERY way out - accept, Reject, hg::Error -
    // before the caller may reuse the staging buffer and the arena or free the witness whose uploads may still be pending. (The guard
    // stands ahead of the first enqueue. Peak arena use: every node's tables stay until the batched finish - the per-node rewind
    // went with the per-kind launches - about the size of the node tables, 0.13 GB at n=32768 k=16.)
    struct Drain {
        hipStream_t st;
        ~Drain() { (void)hipStreamSynchronize(st); }
    } drain{ctx->stream};
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
    if (hg_times("verify")) fprintf(stderr, "[hg] verify_device: inputs enqueued at %.2f ms\n", (omp_get_wtime() - tv0) * 1e3);
    return verify_proof_with(D, p, pk->lasso, pk->circuit, proof, len);
}

}  // namespace hg
