// Round-by-round prover for the protocol modes of SURVEY.md 8(f) f-4 (hg_prove_mode):
//   bit 0  absorbing transcript: write_felt also hashes the element (the in-tree plonkish-trait writer's rule,
//          transcript.rs:205-208, 224-233), so every challenge depends on every earlier prover message;
//   bit 1  extension-field memory checking: gamma, tau stay in E instead of being truncated to base limb 0
//          (lasso/src/memory_checking/prover.rs:36-39; README.md:108 "Known issues").
// Mode 0 (the reference as it is) never comes here: prover.hip enqueues that whole proof behind one synchronisation, because
// its challenges are known up front. With an absorbing transcript they are not, so this prover walks the protocol in
// transcript order and synchronises once per sum-check round:
//   pass 1  the round kernel with a placeholder challenge: its hypercube sums do not depend on the challenge;
//   host    round polynomial -> transcript (absorbed) -> squeeze r -> one 16-byte upload into the device challenge table;
//   pass 2  the same launch again: now the fold it writes is the real one.
// Every kernel is the product kernel of the fast path (kernels.hip), launched for one sum-check at a time; the only extra
// kernels are the Ext2 forms of the hash / product-tree steps that mode bit 1 needs. This is a measured mode, not a tuned
// one: DESIGN.md gives its time next to the headline and says what a tuned version would change (device-side Keccak, the
// fold of round i fused with the sums of round i+1).
#include <algorithm>
#include <chrono>
#include <cstring>
#include "prover.hpp"

namespace hg {
namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

const u64 INV2 = gl_inv(2), INV3 = gl_inv(3), INV6 = gl_inv(6);
void interpolate(const E2* ev, int d, E2* c) {  // evaluations at 0..d -> coefficients low..high (d = 2 or 3)
    E2 d1 = e2_sub(ev[1], ev[0]);
    E2 d2 = e2_add(e2_sub(ev[2], e2_dbl(ev[1])), ev[0]);
    if (d == 2) { c[0] = ev[0]; c[2] = e2_mul_f(d2, INV2); c[1] = e2_sub(d1, c[2]); return; }
    E2 d3 = e2_sub(e2_sub(ev[3], ev[0]), e2_mul_f(e2_sub(ev[2], ev[1]), 3));
    c[0] = ev[0];
    c[3] = e2_mul_f(d3, INV6);
    c[2] = e2_mul_f(e2_sub(d2, d3), INV2);
    c[1] = e2_add(e2_sub(d1, e2_mul_f(d2, INV2)), e2_mul_f(d3, INV3));
}
E2 horner(const E2* c, int d, E2 x) {
    E2 r = c[d];
    for (int i = d - 1; i >= 0; i--) r = e2_add(e2_mul(r, x), c[i]);
    return r;
}

// ---- Ext2 forms of the memory-checking steps (mode bit 1) -----------------------------------------------------------
// h = a + v gamma + t gamma^2 - tau with gamma, tau in E (a, v, t small integers)
__device__ __forceinline__ E2 hash_e(u64 a, u64 v, u64 t, E2 g, E2 g2, E2 tau) {
    return e2_sub(e2_add(e2_add_f(e2_mul_f(g, v), a), e2_mul_f(g2, t)), tau);
}
__global__ void k_hash_rw_e2(size_t n, const u64* __restrict__ dim, const u64* __restrict__ ts, const u64* __restrict__ ep, E2 g, E2 g2, E2 tau,
                             E2* __restrict__ rd, E2* __restrict__ wr) {
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const E2 a = hash_e(dim[j], ep[j], ts[j], g, g2, tau);
        rd[j] = a;
        wr[j] = e2_add(a, g2);  // t + 1
    }
}
__global__ void k_hash_if_e2(u32 cutoff, const u64* __restrict__ fc, E2 g, E2 g2, E2 tau, E2* __restrict__ init, E2* __restrict__ fin) {
    const u32 a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= 65536) return;
    const u64 tv = a < cutoff ? (u64)a : 0;
    const E2 h0 = hash_e(a, tv, 0, g, g2, tau);
    init[a] = h0;
    fin[a] = e2_add(h0, e2_mul_f(g2, gl_from_u64(fc[a])));
}
// product tree level on Ext2 rows: out[b][i] = in[b][i] * in[b][i + h]
__global__ void k_prod_level_e2(const E2* __restrict__ in, size_t in_len, E2* __restrict__ out) {
    const size_t h = in_len >> 1;
    const E2* src = in + (size_t)blockIdx.y * in_len;
    E2* dst = out + (size_t)blockIdx.y * h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < h; i += (size_t)gridDim.x * blockDim.x) dst[i] = e2_mul(src[i], src[i + h]);
}
__global__ void k_gp_top_e2(const E2* __restrict__ top, int nb, E2* __restrict__ roots, E2* __restrict__ evals) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const E2 l = top[2 * b], r = top[2 * b + 1];
    roots[b] = e2_mul(l, r);
    evals[2 * b] = l;
    evals[2 * b + 1] = r;
}

struct Claim {  // evaluation claim whose point is a run of this proof's challenge table
    size_t point_off;
    int len;
    E2 value;
};

struct SeqProver {
    hg_ctx* ctx;
    const hg_pk* pk;
    int mode;
    hipStream_t st;
    FsTranscript tr;
    std::vector<E2> chain;  // challenges squeezed so far (host copy)
    E2* d_chain = nullptr;  // the same in HBM: kernels take challenge positions, exactly like the fast path
    size_t chain_cap = 0;
    size_t res_used = 0;
    size_t n_sync = 0;
    std::vector<const u64*> d_vals;
    std::vector<std::vector<Claim>> claims;

    SeqProver(hg_ctx* c, const hg_pk* k, int m) : ctx(c), pk(k), mode(m), st(c->stream) {
        tr.absorb = (m & 1) != 0;
        chain_cap = 1 << 15;
        d_chain = ctx->alloc_n<E2>(chain_cap);
        hip_check(hipMemsetAsync(d_chain, 0, chain_cap * sizeof(E2), st), "clear challenge table");
        for (E2* pbuf : {ctx->d_partials, ctx->d_partials2})
            hip_check(hipMemsetAsync(reinterpret_cast<char*>(pbuf) + dev::PARTIALS_E2 * sizeof(E2), 0, dev::PARTIALS_TICKETS * sizeof(unsigned), st), "clear reduction tickets");
    }
    bool ext_mc() const { return (mode & 2) != 0; }
    E2* d_res() { return ctx->d_res; }
    const E2* h_res() { return ctx->h_res; }
    size_t slot(size_t n) {
        if (res_used + n > ctx->res_cap) throw Error("result buffer exhausted");
        size_t s = res_used;
        res_used += n;
        return s;
    }
    size_t epos() const { return chain.size(); }
    void sync() {
        hip_check(hipStreamSynchronize(st), "round synchronisation");
        n_sync++;
    }
    // squeeze_challenge (transcript.rs:146-157): the value goes into the host chain and into the device table
    E2 squeeze() {
        if (chain.size() >= chain_cap) throw Error("challenge table exhausted");
        const E2 r = tr.squeeze();
        chain.push_back(r);
        E2* h = static_cast<E2*>(stage(&r, sizeof(E2)));
        hip_check(hipMemcpyAsync(d_chain + chain.size() - 1, h, sizeof(E2), hipMemcpyHostToDevice, st), "upload challenge");
        return r;
    }
    void* stage(const void* src, size_t bytes) {
        size_t need = (bytes + 63) & ~(size_t)63;
        if (ctx->stage_used + need > ctx->stage_cap) {  // a proof in this mode uploads thousands of tiny descriptors: recycle
            sync();
            ctx->stage_used = 0;
        }
        void* p = ctx->h_stage + ctx->stage_used;
        ctx->stage_used += need;
        memcpy(p, src, bytes);
        return p;
    }
    template <typename T> T* upload(const T* src, size_t n) {
        T* d = ctx->alloc_n<T>(n ? n : 1);
        if (n) hip_check(hipMemcpyAsync(d, stage(src, n * sizeof(T)), n * sizeof(T), hipMemcpyHostToDevice, st), "upload descriptor");
        return d;
    }
    void write_slots(size_t s, size_t n) { for (size_t i = 0; i < n; i++) tr.write_e(h_res()[s + i]); }

    // transcript side of one round: d+1 coefficients, eval(1) derived from the running claim (C1), then the challenge
    E2 round_message(const E2* sums, int deg, E2& claim) {
        E2 ev[4], c[4];
        ev[0] = sums[0];
        ev[1] = e2_sub(claim, sums[0]);
        ev[2] = sums[1];
        if (deg == 3) ev[3] = sums[2];
        interpolate(ev, deg, c);
        for (int k = 0; k <= deg; k++) tr.write_e(c[k]);
        const E2 r = squeeze();
        claim = horner(c, deg, r);
        return r;
    }

    // ---- prove_sum_check, stride layout (collation / grand-product shapes) ------------------------------------------------
    // tables: ntab rows at `in + t * in_stride` (u64 if base else E2). final evaluations land in d_res[evals_slot ..).
    size_t sumcheck_stride(int kind, const void* in, bool base, size_t in_stride, int ntab, int nvars, const dev::Powers& pw, E2& claim, size_t evals_slot) {
        const int nv = kind == dev::SC_GRANDPROD ? 3 : 2, deg = nv;
        const size_t point_off = epos();
        const size_t sums_slot = slot((size_t)nvars * nv);
        const size_t N = (size_t)1 << nvars;
        dev::StJob J;
        memset(&J, 0, sizeof(J));
        J.in = in; J.in_stride = in_stride;
        J.buf[0] = ctx->alloc_n<E2>((size_t)ntab * std::max<size_t>(N / 2, 1));
        J.buf[1] = ctx->alloc_n<E2>((size_t)ntab * std::max<size_t>(N / 4, 1));
        J.final_out = d_res() + evals_slot;
        J.kind = kind; J.ntab = ntab; J.nvars = nvars; J.base = base ? 1 : 0;
        J.r_off = point_off; J.sums_slot = sums_slot;
        memcpy(J.pw, pw.v, sizeof(J.pw));
        dev::StJob* d_job = upload(&J, 1);
        const void* cur_in = in;
        size_t cur_stride = in_stride;
        for (int rd = 0; rd < nvars; rd++) {
            const int h = nvars - 1 - rd;
            dev::StItem it;
            memset(&it, 0, sizeof(it));
            it.job = 0; it.h_log2 = h; it.in = cur_in; it.in_stride = cur_stride;
            it.out = rd == nvars - 1 ? J.final_out : (cur_in == (const void*)J.buf[0] ? J.buf[1] : J.buf[0]);
            const int grid = dev::st_plan_blocks(&it, 1, false);
            dev::StItem* d_it = upload(&it, 1);
            if (rd == 0) {
                // the first-round kernels store the weighted fold pw[i] * (x + r d) as pw[i] x + pwr[i] d with pwr = pw * r_0: r_0 is
                // not known yet, so the first pass runs with pwr = 0 (sums unaffected) and the job is re-uploaded before pass 2
                dev::st_step(st, kind, base, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res());
            } else dev::st_step(st, kind, false, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res());
            sync();
            const E2 r = round_message(h_res() + sums_slot + (size_t)rd * nv, deg, claim);
            if (rd == 0) {
                for (int i = 0; i < dev::PW_MAX; i++) J.pwr[i] = e2_mul(pw.v[i], r);
                hip_check(hipMemcpyAsync(d_job, stage(&J, sizeof(J)), sizeof(J), hipMemcpyHostToDevice, st), "upload job");
            }
            dev::st_step(st, kind, base && rd == 0, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res());  // pass 2: the real fold
            cur_in = it.out; cur_stride = (size_t)1 << h;
        }
        return point_off;
    }

    // ---- prove_sum_check, sum of pair products (Libra / zkCNN reductions) --------------------------------------------------
    size_t sumcheck_prodsum(const std::vector<const u64*>& a, const std::vector<const E2*>& b, int nvars, const std::vector<E2*>& fin_a,
                            const std::vector<E2*>& fin_b, E2& claim) {
        const size_t point_off = epos();
        const size_t sums_slot = slot((size_t)nvars * 2);
        const int np = (int)a.size();
        if (np > dev::PS_MAX_PAIRS) throw Error("prodsum: too many table pairs");
        const size_t N = (size_t)1 << nvars;
        dev::PsJob J;
        memset(&J, 0, sizeof(J));
        J.npairs = np; J.nvars = nvars; J.r_off = point_off; J.sums_slot = sums_slot; J.tail_rd = nvars; J.tail_buf = -1;
        for (int q = 0; q < 2; q++) {
            J.bufa[q] = ctx->alloc_n<E2>((size_t)np * std::max<size_t>(N >> (q + 1), 1));
            J.bufb[q] = ctx->alloc_n<E2>((size_t)np * std::max<size_t>(N >> (q + 1), 1));
        }
        for (int i = 0; i < np; i++) { J.a[i] = a[i]; J.b[i] = b[i]; J.fin_a[i] = fin_a[i]; J.fin_b[i] = fin_b[i]; }
        dev::PsJob* d_job = upload(&J, 1);
        int cur = -1;
        for (int rd = 0; rd < nvars; rd++) {
            dev::PsItem it;
            memset(&it, 0, sizeof(it));
            it.job = 0; it.rd = rd; it.in_buf = cur;
            it.out_buf = rd == nvars - 1 ? -1 : (cur == 0 ? 1 : 0);
            const int grid = dev::ps_plan_blocks(&it, 1, &J, false);
            dev::PsItem* d_it = upload(&it, 1);
            dev::ps_round(st, false, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res());
            sync();
            round_message(h_res() + sums_slot + 2 * (size_t)rd, 2, claim);
            dev::ps_round(st, false, d_job, d_it, 1, grid, d_chain, ctx->d_partials, d_res());
            cur = it.out_buf;
        }
        return point_off;
    }

    // ---- small device helpers ---------------------------------------------------------------------------------------------------
    void eq_table(E2* out, int n, const dev::ClaimSet& cs) {
        const size_t N = (size_t)1 << n;
        std::vector<dev::EqJob> jobs;
        E2* tmp = cs.n > 1 ? ctx->alloc_n<E2>((size_t)cs.n * N) : nullptr;
        for (int a = 0; a < cs.n; a++) {
            dev::EqJob J;
            memset(&J, 0, sizeof(J));
            J.n = n; J.out = cs.n > 1 ? tmp + (size_t)a * N : out;
            J.cs.n = 1; J.cs.unit_alpha = cs.unit_alpha; J.cs.alpha_off = cs.alpha_off + a; J.cs.point_off[0] = cs.point_off[a];
            jobs.push_back(J);
        }
        dev::eq_jobs(st, upload(jobs.data(), jobs.size()), (int)jobs.size(), n, d_chain);
        if (cs.n > 1) dev::sum_tables(st, out, tmp, cs.n, N);
    }
    void eq_single(E2* out, int n, size_t point_off) {
        dev::ClaimSet cs;
        memset(&cs, 0, sizeof(cs));
        cs.n = 1; cs.unit_alpha = 1; cs.point_off[0] = point_off;
        eq_table(out, n, cs);
    }
    // out[t] = sum_j eq[j] tabs[t][j]
    void dots(const E2* eq, const std::vector<const u64*>& tabs, size_t n, size_t out_slot) {
        for (size_t o = 0; o < tabs.size(); o += 8) {
            const int cnt = (int)std::min<size_t>(8, tabs.size() - o);
            const u64* t8[8] = {nullptr};
            for (int t = 0; t < cnt; t++) t8[t] = tabs[o + t];
            dev::dot_eq(st, eq, t8, cnt, n, ctx->d_partials, d_res() + out_slot + o);
        }
    }

    // ---- prove_grand_product (prover.rs:183-266) on nb rows of `len` entries (u64, or E2 when `ext`) ---------------------
    size_t grand_product(const void* H, bool ext, size_t len, int nb) {
        int nv = 0;
        while (((size_t)1 << nv) < len) nv++;
        const size_t el = ext ? sizeof(E2) : sizeof(u64);
        std::vector<const void*> lev(nv, nullptr);
        lev[0] = H;
        for (int k = 1; k < nv; k++) {  // Layer::bottom / Layer::up
            const size_t in_len = len >> (k - 1);
            void* out = ctx->alloc((size_t)nb * (len >> k) * el);
            if (ext) k_prod_level_e2<<<dim3((unsigned)std::min<size_t>((in_len / 2 + 255) / 256, 256), (unsigned)nb), 256, 0, st>>>(static_cast<const E2*>(lev[k - 1]), in_len, static_cast<E2*>(out));
            else dev::prod_level(st, static_cast<const u64*>(lev[k - 1]), in_len, static_cast<u64*>(out), nb);
            lev[k] = out;
        }
        const size_t roots = slot(nb), ev0 = slot(2 * (size_t)nb);
        if (ext) k_gp_top_e2<<<(nb + 63) / 64, 64, 0, st>>>(static_cast<const E2*>(lev[nv - 1]), nb, d_res() + roots, d_res() + ev0);
        else dev::gp_top(st, static_cast<const u64*>(lev[nv - 1]), nb, d_res() + roots, d_res() + ev0);
        sync();
        std::vector<E2> cl(nb);
        for (int b = 0; b < nb; b++) { cl[b] = h_res()[roots + b]; tr.write_e(cl[b]); }  // root products (prover.rs:197-221)
        write_slots(ev0, 2 * (size_t)nb);                                                // layer 0: v_l, v_r (prover.rs:257)
        size_t point_off = epos();
        auto layer_down = [&](size_t evals_slot) {  // prover.rs:259, 288-294
            const E2 mu = squeeze();
            const E2* ev = h_res() + evals_slot;
            for (int b = 0; b < nb; b++) cl[b] = e2_add(ev[2 * b], e2_mul(mu, e2_sub(ev[2 * b + 1], ev[2 * b])));
        };
        layer_down(ev0);
        for (int n = 1; n < nv; n++) {
            const int k = nv - 1 - n;
            const size_t h = (size_t)1 << n;
            const E2 gamma = squeeze();  // prover.rs:238
            dev::Powers pw;
            memset(&pw, 0, sizeof(pw));
            if (nb > dev::PW_MAX) throw Error("grand product: too many batched tables");
            E2 g = e2_one(), claim = e2_zero();
            for (int b = 0; b < nb; b++) { pw.v[b] = g; claim = e2_add(claim, e2_mul(cl[b], g)); g = e2_mul(g, gamma); }  // prover.rs:281-286
            const size_t evals = slot(2 * (size_t)nb);
            point_off = sumcheck_stride(dev::SC_GRANDPROD, lev[k], !ext, h, 2 * nb, n, pw, claim, evals);
            sync();
            // the kernels leave the final LEFT evaluation of pair b multiplied by pw[b] = gamma^b
            if (nb >= 2) {
                if (pw.v[1].c0 == 0 && pw.v[1].c1 == 0) throw Error("grand product: zero batching weight");
                const E2 ginv = e2_inv(pw.v[1]);
                E2 w = ginv;
                for (int b = 1; b < nb; b++) { ctx->h_res[evals + 2 * b] = e2_mul(ctx->h_res[evals + 2 * b], w); w = e2_mul(w, ginv); }
            }
            write_slots(evals, 2 * (size_t)nb);  // prover.rs:257
            layer_down(evals);
        }
        return point_off;
    }

    // ---- LassoNode::prove_claim_reduction (lasso.rs:57-114) ------------------------------------------------------------------
    Claim lasso_node(const u64* d_input) {
        const LassoPlan& lp = pk->lasso;
        const dev::LassoDev& L = pk->lasso_dev;
        const int nu = lp.nu, A = lp.alpha;
        const size_t N = (size_t)1 << nu, M = 65536;
        u64* dims = ctx->alloc_n<u64>(4 * N);
        u64* ep = ctx->alloc_n<u64>((size_t)A * N);
        dev::lasso_split(st, L, d_input, dims, ep, dev::ep_rows_all(A));  // polynomialize (lasso.rs:157-250)
        const size_t r_off = epos();
        for (int i = 0; i < nu; i++) squeeze();       // lasso.rs:85
        E2* eq = ctx->alloc_n<E2>(N);
        eq_single(eq, nu, r_off);
        const size_t claim_slot = slot(1);
        {
            int grid = dev::lasso_claim(st, L, eq, ep, dev::ep_rows_all(A), ctx->d_partials);
            dev::reduce_partials(st, ctx->d_partials, grid, 1, d_res() + claim_slot);
        }
        sync();
        const E2 claimed = h_res()[claim_slot];
        tr.write_e(claimed);  // lasso.rs:100-107
        {   // collation sum-check (lasso.rs:271-279), result dropped (:97)
            dev::Powers pw;
            memset(&pw, 0, sizeof(pw));
            if (A > dev::PW_MAX) throw Error("lasso: too many memories");
            u64 c = 1;
            for (int i = 0; i < A; i++) { pw.v[i] = e2(c, 0); c = gl_mul(c, M); }
            E2 claim = claimed;
            sumcheck_stride(dev::SC_COLLATION, ep, true, N, A, nu, pw, claim, slot(A));
        }
        const E2 gamma_e = squeeze(), tau_e = squeeze();  // lasso.rs:99
        // counters of the chunk-indexed memories (lasso.rs:317-319)
        std::map<int, u64*> read_ts, final_cts;
        {
            size_t tb = dev::lasso_counter_temp_bytes(N);
            void* temp = ctx->alloc(tb);
            u32* keys = ctx->alloc_n<u32>(N); u32* keys2 = ctx->alloc_n<u32>(N);
            u32* rows = ctx->alloc_n<u32>(N); u32* rows2 = ctx->alloc_n<u32>(N);
            u32* starts = ctx->alloc_n<u32>(65537);
            for (auto& chk : lp.chunks) {
                int c = chk.first;
                if (c < 0 || c >= 4) continue;
                read_ts[c] = ctx->alloc_n<u64>(N);
                final_cts[c] = ctx->alloc_n<u64>(M);
                dev::lasso_counters(st, L, c, dims, read_ts[c], final_cts[c], temp, tb, keys, keys2, rows, rows2, starts);
            }
        }
        // MemoryCheckingProver::new (prover.rs:35-89): reads | writes of every memory in memory-GKR order, then inits | finals
        const int G = (int)lp.gkr_order.size();
        if (G > 32) throw Error("lasso: more than 32 memories");
        size_t xoff, yoff;
        if (!ext_mc()) {
            const u64 gamma = gamma_e.c0, tau = tau_e.c0;  // prover.rs:38-39
            u64* H1 = ctx->alloc_n<u64>((size_t)2 * G * N);
            u64* H2 = ctx->alloc_n<u64>((size_t)2 * G * M);
            for (int i = 0; i < G; i++) {
                dev::HashRwArgs ha;
                memset(&ha, 0, sizeof(ha));
                ha.ep[0] = ep + (size_t)lp.gkr_order[i] * N; ha.rd[0] = H1 + (size_t)i * N; ha.wr[0] = H1 + (size_t)(G + i) * N;
                const int c = lp.gkr_chunk[i];
                dev::lasso_hash_rw(st, N, dims + (size_t)c * N, read_ts[c], ha, 1, gamma, tau);
            }
            dev::HashIfArgs hi;
            memset(&hi, 0, sizeof(hi));
            for (int i = 0; i < G; i++) { hi.cutoff[i] = (u32)lp.mems[lp.gkr_order[i]].cutoff; hi.fc[i] = final_cts[lp.gkr_chunk[i]]; hi.row_init[i] = i; hi.row_fin[i] = G + i; }
            dev::lasso_hash_if(st, hi, G, gamma, tau, H2);
            xoff = grand_product(H1, false, N, 2 * G);   // prover.rs:161-165
            yoff = grand_product(H2, false, M, 2 * G);   // prover.rs:167-171
        } else {
            const E2 g2 = e2_mul(gamma_e, gamma_e);
            E2* H1 = ctx->alloc_n<E2>((size_t)2 * G * N);
            E2* H2 = ctx->alloc_n<E2>((size_t)2 * G * M);
            for (int i = 0; i < G; i++) {
                const int c = lp.gkr_chunk[i];
                k_hash_rw_e2<<<1024, 256, 0, st>>>(N, dims + (size_t)c * N, read_ts[c], ep + (size_t)lp.gkr_order[i] * N, gamma_e, g2, tau_e,
                                                   H1 + (size_t)i * N, H1 + (size_t)(G + i) * N);
                k_hash_if_e2<<<256, 256, 0, st>>>((u32)lp.mems[lp.gkr_order[i]].cutoff, final_cts[c], gamma_e, g2, tau_e, H2 + (size_t)i * M, H2 + (size_t)(G + i) * M);
            }
            xoff = grand_product(H1, true, N, 2 * G);
            yoff = grand_product(H2, true, M, 2 * G);
        }
        // openings (prover.rs:173-178, mod.rs:80-93)
        E2* eqx = eq;
        E2* eqy = ctx->alloc_n<E2>(M);
        eq_single(eqx, nu, xoff);
        eq_single(eqy, 16, yoff);
        std::vector<std::pair<size_t, size_t>> wires;  // (slot, count) in wire order
        for (auto& chk : lp.chunks) {
            const int c = chk.first;
            const size_t base = slot(3 + chk.second.size());
            std::vector<const u64*> xs = {dims + (size_t)c * N, read_ts[c]};
            const size_t tmp = slot(2 + chk.second.size());
            for (int m : chk.second) xs.push_back(ep + (size_t)m * N);
            dots(eqx, xs, N, tmp);
            dots(eqy, {final_cts[c]}, M, base + 2);
            wires.push_back({base, tmp});
        }
        sync();
        size_t q = 0;
        for (auto& chk : lp.chunks) {  // dim(x), read_ts(x), final_cts(y), then E_m(x)
            const size_t base = wires[q].first, tmp = wires[q].second;
            q++;
            tr.write_e(h_res()[tmp]); tr.write_e(h_res()[tmp + 1]); tr.write_e(h_res()[base + 2]);
            for (size_t i = 0; i < chk.second.size(); i++) tr.write_e(h_res()[tmp + 2 + i]);
        }
        return Claim{r_off, nu, claimed};  // lasso.rs:97,113
    }

    // ---- prove_gkr (sk_encryption_circuit.rs:455-457), conventions G1-G4 of the fast path ---------------------------------
    dev::ClaimSet claim_set(const std::vector<Claim>& cl, std::vector<E2>* alphas) {
        dev::ClaimSet cs;
        memset(&cs, 0, sizeof(cs));
        if (cl.empty()) throw Error("gkr: node without claim");
        if (cl.size() > (size_t)dev::MAX_CLAIMS) throw Error("gkr: too many claims on one node");
        cs.n = (int)cl.size();
        cs.unit_alpha = cl.size() == 1;
        cs.alpha_off = epos();
        alphas->clear();
        if (cl.size() > 1) for (size_t a = 0; a < cl.size(); a++) alphas->push_back(squeeze());
        else alphas->push_back(e2_one());
        for (size_t a = 0; a < cl.size(); a++) cs.point_off[a] = cl[a].point_off;
        return cs;
    }
    static E2 combined(const std::vector<Claim>& cl, const std::vector<E2>& alphas) {
        E2 s = e2_zero();
        for (size_t a = 0; a < cl.size(); a++) s = e2_add(s, e2_mul(cl[a].value, alphas[a]));
        return s;
    }

    void vanilla_node(int id) {
        const HNode& n = pk->circuit.nodes[id];
        const hg_pk::NodeDev& nd = pk->node_dev[id];
        const int nin = n.log2_sub_in + n.log2_reps;
        const size_t SR = (size_t)1 << nin;
        std::vector<E2> alphas;
        dev::ClaimSet cs = claim_set(claims[id], &alphas);
        for (auto& c : claims[id]) if (c.len != n.log2_out()) throw Error("gkr: claim arity mismatch");
        E2 claim = combined(claims[id], alphas);
        E2* eqc = ctx->alloc_n<E2>((size_t)1 << n.log2_out());
        eq_table(eqc, n.log2_out(), cs);
        if (nd.nconst) {  // claim -= sum_g eqc[g] w0_g
            const size_t s = slot(1);
            int grid = dev::vanilla_const_sum(st, nd.const_gate, nd.const_coef, nd.nconst, eqc, n.log2_sub_out, n.log2_reps, ctx->d_partials);
            dev::reduce_partials(st, ctx->d_partials, grid, 1, d_res() + s);
            sync();
            claim = e2_sub(claim, h_res()[s]);
        }
        std::vector<int> li, ri;
        for (int i = 0; i < n.arity; i++) { if (n.left_use[i]) li.push_back(i); if (n.right_use[i]) ri.push_back(i); }
        if (n.arity > dev::PS_MAX_PAIRS) throw Error("vanilla: arity too large");
        dev::GatherT gt;
        memset(&gt, 0, sizeof(gt));
        for (int i = 0; i < n.arity; i++) gt.in_vals[i] = d_vals[n.preds[i]];
        std::vector<const u64*> a;
        std::vector<const E2*> b;
        std::vector<E2*> fa, fb;
        const size_t u_base = slot(n.arity);
        E2* scratch = ctx->alloc_n<E2>(n.arity);
        std::vector<dev::GatherJob> gj;
        for (int i : li) {
            E2* T = ctx->alloc_n<E2>(SR);
            gt.lin = nd.lin[i];
            gt.mul = nd.mulL[i];
            gj.push_back(dev::GatherJob{gt, eqc, n.log2_sub_in, n.log2_sub_out, n.log2_reps, T});
            a.push_back(d_vals[n.preds[i]]);
            b.push_back(T);
            fa.push_back(d_res() + u_base + i);
            fb.push_back(scratch + i);
        }
        if (!gj.empty()) dev::gather_jobs(st, upload(gj.data(), gj.size()), (int)gj.size(), SR);
        const size_t rx_off = sumcheck_prodsum(a, b, nin, fa, fb, claim);  // Libra phase 1
        sync();
        for (int i : li) {
            const E2 v = h_res()[u_base + i];
            tr.write_e(v);
            claims[n.preds[i]].push_back(Claim{rx_off, nin, v});
        }
        if (n.mul.empty()) return;
        if (!n.lin.empty()) throw Error("vanilla: nodes mixing linear and mul gates are not on this path");
        // phase 2: sum_y sum_i in_i(y) B_i(y); the claim carries over (no linear part)
        E2* eqx = ctx->alloc_n<E2>(SR);
        eq_single(eqx, nin, rx_off);
        std::vector<const u64*> a2;
        std::vector<const E2*> b2;
        std::vector<E2*> fa2, fb2;
        const size_t w_base = slot(n.arity);
        std::vector<dev::GatherBJob> bj;
        for (int i : ri) {
            E2* B = ctx->alloc_n<E2>(SR);
            bj.push_back(dev::GatherBJob{nd.mulR[i], eqc, eqx, d_res() + u_base, n.log2_sub_in, n.log2_sub_out, n.log2_reps, B});
            a2.push_back(d_vals[n.preds[i]]);
            b2.push_back(B);
            fa2.push_back(d_res() + w_base + i);
            fb2.push_back(scratch + i);
        }
        dev::gather_B_jobs(st, upload(bj.data(), bj.size()), (int)bj.size(), SR);
        const size_t ry_off = sumcheck_prodsum(a2, b2, nin, fa2, fb2, claim);
        sync();
        for (int i : ri) {
            const E2 v = h_res()[w_base + i];
            tr.write_e(v);
            claims[n.preds[i]].push_back(Claim{ry_off, nin, v});
        }
    }

    void fft_node(int id) {
        const HNode& n = pk->circuit.nodes[id];
        const int L = n.log2_size;
        const size_t N = (size_t)1 << L;
        std::vector<E2> alphas;
        dev::ClaimSet cs = claim_set(claims[id], &alphas);
        E2 claim = combined(claims[id], alphas);
        E2* F = ctx->alloc_n<E2>(N);
        const u64* W = (n.inverse ? pk->w_inv : pk->w_fwd).at(L);
        dev::FftJob fj{F, W, n.inverse ? gl_inv(gl_from_u64(N)) : 1, L, cs};
        E2* tab = ctx->alloc_n<E2>((size_t)cs.n * (N >> 4) + 1);
        dev::fft_jobs(st, upload(&fj, 1), 1, L, cs.n, d_chain, tab);
        const size_t u = slot(1);
        E2* scratch = ctx->alloc_n<E2>(1);
        const size_t off = sumcheck_prodsum({d_vals[n.preds[0]]}, {F}, L, {d_res() + u}, {scratch}, claim);
        sync();
        const E2 v = h_res()[u];
        tr.write_e(v);
        claims[n.preds[0]].push_back(Claim{off, L, v});
    }

    void gkr(const Claim& sum_claim) {
        const HCircuit& c = pk->circuit;
        claims.assign(c.nodes.size(), {});
        claims[c.lasso_id].push_back(Claim{epos(), 0, e2_zero()});  // EvalClaim::new(vec![], E::ZERO) (:450)
        claims[c.sum_id].push_back(sum_claim);
        for (size_t q = c.topo.size(); q-- > 0;) {
            const int id = c.topo[q];
            const HNode& n = c.nodes[id];
            switch (n.kind) {
                case NK_INPUT: break;
                case NK_VANILLA: vanilla_node(id); break;
                case NK_FFT: fft_node(id); break;
                case NK_LASSO: claims[n.preds[0]].push_back(lasso_node(d_vals[n.preds[0]])); break;
            }
        }
    }
};

}  // namespace

// BfvEncrypt::prove on resident node values in a non-default protocol mode; gpu_ms = wall time of the whole walk (the device
// is synchronised every round, so there is no separate device span)
ProveResult prove_resident_mode(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode) {
    if (mode == 0) return prove_resident(ctx, pk, v);
    if (mode < 0 || mode > 3) throw Error("hg_prove_mode: unknown mode bits");
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    const double t0 = now_ms();
    SeqProver P(ctx, pk, mode);
    P.d_vals = v->d_vals;
    const Params& p = pk->params;
    // "eval output" (sk_encryption_circuit.rs:444-448)
    const int ov = p.ct0is_log2();
    const size_t point_off = P.epos();
    for (int i = 0; i < ov; i++) P.squeeze();
    const size_t vslot = P.slot(1);
    E2* eq = ctx->alloc_n<E2>((size_t)1 << ov);
    P.eq_single(eq, ov, point_off);
    P.dots(eq, {v->d_ct0is}, (size_t)1 << ov, vslot);
    P.sync();
    P.gkr(Claim{point_off, ov, ctx->h_res[vslot]});
    hip_check(hipGetLastError(), "prove (mode)");
    ProveResult res;
    res.prove_ms = now_ms() - t0;
    res.gpu_ms = res.prove_ms;
    res.sync_ms = (double)P.n_sync;  // number of synchronisations (reported through hg_timings::sync_ms in this mode)
    res.proof = std::move(P.tr.bytes);
    return res;
}

}  // namespace hg
