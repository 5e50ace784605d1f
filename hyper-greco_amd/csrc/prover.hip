#include "prover.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <tuple>

namespace hg {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Error(std::string(what) + ": " + hipGetErrorString(e));
}
static double wall_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace hg

using namespace hg;

// ------------------------------------------------------------------------------------------------
// context
void* hg_ctx::alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    auto note = [this] { size_t u = 0; for (auto& c : chunks) u += c.used; arena_high = std::max(arena_high, u); };
    for (auto& c : chunks)
        if (c.cap - c.used >= bytes) { void* p = c.p + c.used; c.used += bytes; c.high = std::max(c.high, c.used); note(); return p; }
    if (arena_fixed) throw Error("arena: a recorded prove needs more workspace than the walked prove it follows");
    size_t cap = std::max<size_t>(bytes, (size_t)512 << 20);
    char* p = nullptr;
    hip_check(hipMalloc((void**)&p, cap), "hipMalloc(arena chunk)");
    chunks.push_back({p, cap, bytes, bytes});
    arena_total += cap;
    note();
    return p;
}
std::vector<size_t> hg_ctx::arena_mark() const {
    std::vector<size_t> m;
    for (auto& c : chunks) m.push_back(c.used);
    return m;
}
void hg_ctx::arena_rewind(const std::vector<size_t>& mark) {
    for (size_t i = 0; i < chunks.size(); i++) chunks[i].used = i < mark.size() ? mark[i] : 0;
}
void hg_ctx::arena_skip_to_high() {
    for (auto& c : chunks) c.used = std::max(c.used, c.high);
}
void hg_ctx::arena_reset() {
    arena_epoch++;
    // coalesce into one chunk once the high-water mark is known, so later proves never call hipMalloc
    size_t used = 0;
    for (auto& c : chunks) used += c.used;
    used = std::max(used, arena_high);
    if (chunks.size() > 1) {
        hip_check(hipStreamSynchronize(stream), "sync before arena coalesce");
        for (auto& c : chunks) (void)hipFree(c.p);
        chunks.clear();
        size_t cap = used + used / 8 + ((size_t)64 << 20);
        char* p = nullptr;
        hip_check(hipMalloc((void**)&p, cap), "hipMalloc(arena)");
        chunks.push_back({p, cap, 0, 0});
        arena_total = cap;
        arena_high = 0;
    }
    for (auto& c : chunks) { c.used = 0; c.high = 0; }
    stage_used = 0;
    bn_flushed = 0;
    bn_res_used = 0;
}
void hg_ctx::ensure_chain(size_t n_e) {
    if (n_e <= chal_e) return;
    size_t want = std::max<size_t>(n_e, chal_e ? chal_e * 2 : 16384);
    const u64* host = challenge_chain(2 * want);
    hip_check(hipStreamSynchronize(stream), "sync before chain growth");
    if (d_chal) (void)hipFree(d_chal);
    hip_check(hipMalloc((void**)&d_chal, want * sizeof(E2)), "hipMalloc(chain)");
    hip_check(hipMemcpy(d_chal, host, want * sizeof(E2), hipMemcpyHostToDevice), "upload chain");
    chal_e = want;
}
int hg_ctx::prof_class(const char* name, bool dominant) {
    for (size_t i = 0; i < prof_stats.size(); i++) if (prof_stats[i].name == name) return (int)i;
    ProfStat s; s.name = name; s.dominant = dominant;
    prof_stats.push_back(s);
    return (int)prof_stats.size() - 1;
}
void hg_ctx::prof_begin(int cls, double bytes, double model_bytes, double design_bytes) {
    cur_cls = -1;
    if (prof_level == 0) return;
    if (prof_level == 1 && !prof_stats[cls].dominant) return;
    hipEvent_t a, b;
    auto get = [&]() { if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; } hipEvent_t e; (void)hipEventCreate(&e); return e; };
    a = get(); b = get();
    (void)hipEventRecord(a, prof_stream);
    cur_cls = cls; cur_a = a;
    prof_events.push_back({cls, a, b});
    prof_stats[cls].launches++;
    prof_stats[cls].bytes += bytes;
    prof_stats[cls].model += model_bytes < 0 ? bytes : model_bytes;
    prof_stats[cls].design += design_bytes < 0 ? bytes : design_bytes;
}
void hg_ctx::prof_end() {
    if (cur_cls < 0) return;
    (void)hipEventRecord(prof_events.back().b, prof_stream);
    cur_cls = -1;
}
void hg_ctx::prof_collect() {
    for (auto& e : prof_events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) prof_stats[e.cls].ms += ms;
        event_pool.push_back(e.a); event_pool.push_back(e.b);
    }
    prof_events.clear();
}
hg_ctx::~hg_ctx() {
    if (stream) (void)hipStreamSynchronize(stream);
    hg::ctx_register(this, false);
    hg::pending_shard_drop(this);
    hg::prove_cache_drop(this);
    if (scratch_values) hg::values_free(scratch_values);
    for (auto& v : stream_values) if (v) hg::values_free(v);
    for (auto& p : stream_pinned) if (p) (void)hipHostFree(p);
    if (stream3) { (void)hipStreamSynchronize(stream3); (void)hipStreamDestroy(stream3); }
    for (auto e : ev_ready) if (e) (void)hipEventDestroy(e);
    for (auto& c : chunks) (void)hipFree(c.p);
    if (d_chal) (void)hipFree(d_chal);
    if (d_res && d_res != h_res) (void)hipFree(d_res);
    if (h_res) (void)hipHostFree(h_res);
    if (h_stage) (void)hipHostFree(h_stage);
    if (bn_dstage) (void)hipFree(bn_dstage);
    if (h_mailbox) (void)hipHostFree(h_mailbox);
    if (d_partials) (void)hipFree(d_partials);
    if (d_partials2) (void)hipFree(d_partials2);
    if (d_partials3) (void)hipFree(d_partials3);
    if (stream_col) { (void)hipStreamSynchronize(stream_col); (void)hipStreamDestroy(stream_col); }
    if (ev_col) (void)hipEventDestroy(ev_col);
    if (comm) { try { hg::comm_destroy(this); } catch (...) {} }
    if (d_xchg) (void)hipFree(d_xchg);
    if (stream2) { (void)hipStreamSynchronize(stream2); (void)hipStreamDestroy(stream2); }
    if (bn_stream_hi) (void)hipStreamDestroy(bn_stream_hi);
    if (bn_stream_lo) (void)hipStreamDestroy(bn_stream_lo);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    for (auto e : ev_aux) if (e) (void)hipEventDestroy(e);
    for (auto e : event_pool) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
}

namespace hg {

// ------------------------------------------------------------------------------------------------
// host-side scalar glue (round-polynomial interpolation, Horner)
static const u64 INV2 = gl_inv(2), INV3 = gl_inv(3), INV6 = gl_inv(6);

// coefficients (low -> high) of the degree-d polynomial through (0,e0) (1,e1) .. (d,ed), d in {2,3}
static void interpolate(const E2* ev, int d, E2* c) {
    E2 d1 = e2_sub(ev[1], ev[0]);
    E2 d2 = e2_add(e2_sub(ev[2], e2_dbl(ev[1])), ev[0]);  // second finite difference
    if (d == 2) {
        c[0] = ev[0];
        c[2] = e2_mul_f(d2, INV2);
        c[1] = e2_sub(d1, c[2]);
        return;
    }
    // third finite difference e3 - 3 e2 + 3 e1 - e0
    E2 d3 = e2_sub(e2_sub(ev[3], ev[0]), e2_mul_f(e2_sub(ev[2], ev[1]), 3));
    c[0] = ev[0];
    c[3] = e2_mul_f(d3, INV6);
    c[2] = e2_mul_f(e2_sub(d2, d3), INV2);
    c[1] = e2_add(e2_sub(d1, e2_mul_f(d2, INV2)), e2_mul_f(d3, INV3));
}
static E2 horner(const E2* c, int d, E2 x) {
    E2 r = c[d];
    for (int i = d - 1; i >= 0; i--) r = e2_add(e2_mul(r, x), c[i]);
    return r;
}

// ---- single-proof sharding over `world` GPUs: who owns what (shared by the prover and by the sharded witness generation) -------------
ShardPlan shard_plan(const hg_pk* pk, int rank, int world) {
    ShardPlan sp;
    const HCircuit& c = pk->circuit;
    const int nu = pk->lasso.nu;
    sp.node_owner.assign(c.nodes.size(), 0);
    sp.own_out_claim = 0;
    if (world <= 1) return sp;
    const int G = (int)pk->lasso.gkr_order.size();
    const double N = (double)((size_t)1 << nu);
    sp.gp1_mem_owner.assign(G, 0);
    for (int i = 0; i < G; i++) sp.gp1_mem_owner[i] = (int)(((long long)i * world) / G);
    // Load model in "table entries touched", calibrated on MI355X at n=32768 k=16. Per owned memory: two grand-product tables
    // through every layer (~9 passes each), its E table in the split, the claim, the collation sum-check and the opening (~6).
    // Per rank that owns any: the limb split of the input, the counters of the chunks it needs, and the p_0 tables.
    std::vector<double> load(world, 0.0);
    std::vector<int> nmem(world, 0);
    for (int i = 0; i < G; i++) { load[sp.gp1_mem_owner[i]] += N * (2.0 * 9.0 + 6.0); nmem[sp.gp1_mem_owner[i]]++; }
    for (int r = 0; r < world; r++) if (nmem[r]) load[r] += N * (5.0 + 3.5 * std::min(4, nmem[r]) + (r == sp.gp1_mem_owner[0] ? 0.0 : 11.0));
    struct Item { double cost; int idx; };
    std::vector<Item> items;
    for (size_t id = 0; id < c.nodes.size(); id++) {
        const HNode& n = c.nodes[id];
        if (n.kind == NK_FFT) items.push_back({(double)((size_t)1 << n.log2_size) * 18.0, (int)id});
        if (n.kind == NK_VANILLA) {
            int np = 0;
            for (int i = 0; i < n.arity; i++) np += n.left_use[i] + n.right_use[i];
            // (the eq-factored nodes - hg_pk::NodeDev::EqForm - are cheaper than this says, but pricing them so dealt a rank nodes of
            // every modulus chain: no faster in the one-GPU projection, 1.31 / 1.42 against 1.38 / 1.34 ms on two ranks, and the rank's
            // evaluation cone became the whole circuit)
            items.push_back({(double)np * (double)((size_t)1 << (n.log2_sub_in + n.log2_reps)) * 12.0 + 3.0 * (double)((size_t)1 << n.log2_out()), (int)id});
        }
    }
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.cost > b.cost; });
    for (auto& it : items) {   // longest-processing-time-first
        int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        load[r] += it.cost;
        sp.node_owner[it.idx] = r;
    }
    sp.own_out_claim = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    if (hg_debug("shard") && rank == 0) {
        fprintf(stderr, "[hg] shard plan world %d: %d memories; out-claim -> %d; loads", world, G, sp.own_out_claim);
        for (int r = 0; r < world; r++) fprintf(stderr, " %.1fM", load[r] / 1e6);
        fprintf(stderr, "\n");
    }
    return sp;
}

typedef std::shared_ptr<E2> Cell;
static Cell cell(E2 v = e2_zero()) { return std::make_shared<E2>(v); }

struct ClaimRef {  // an evaluation claim whose point is a run of the challenge chain
    size_t point_off;
    int len;
    Cell value;
};

struct ScHandle {
    size_t sums_slot = 0;
    int nv = 2;  // sums per round: t = 0,2[,3]
    int nvars = 0;
    size_t point_off = 0;
    std::vector<E2> rs;
    bool scaled = false;   // the device sums are the round sums divided by `scale` (mirrored grand product, StJob::mirror)
    E2 scale = {0, 0};
};
// Slot form of a mirrored top layer (GpHashSrc::slot_of in kernels.hpp): the job runs on `2 nslots + 1` tables until its tables are short
// enough for the single-workgroup tail, which runs on the `tail_ntab` = 2 nrows + 1 per-memory tables gathered from them.
struct SlotPlan {
    int tail_ntab = 0; const uint8_t* d_slot_of = nullptr; const E2* d_ratio = nullptr; int nrows = 0, nslots = 0, npairs = 0, max_rd = 0;
    const E2* job_slotw = nullptr; const u64* job_emit = nullptr;   // layers below the top one: StJob::slotw / emit_mask (the top layer's are in its GpHashSrc)
};
struct MirrorSpec { E2 k1, k2; int credit_ntab; };  // StJob::mk1 / mk2; the table count the launch is credited with (the unmirrored batch)

struct Prover {
    hg_ctx* ctx;
    const hg_pk* pk;
    hipStream_t st;   // stream currently being enqueued to
    E2* partials;     // its per-workgroup partial-sum scratch
    bool forked = false;
    ChallengeSource ch;
    ProofStream proof;
    std::vector<std::function<void()>> ops;  // transcript steps, replayed after the single synchronisation
    // Out-of-order replay (cached launch graphs, one rank): the steps of the node reductions that do not descend from the Lasso node
    // (55 of the 65 nodes at n=32768 k=16, two thirds of the replay's arithmetic) read only results the second stream has written by
    // the time it raises `early_slot`, 0.15 ms before the prove ends (the Lasso node's openings follow): the host runs them while it
    // waits, each at the byte offset the first in-order replay of this object recorded for it, and only the rest after the
    // synchronisation. The transcript is the same sequence of bytes; only the order in which the host fills it in changes.
    std::vector<char> op_early;      // parallel to ops
    std::vector<size_t> op_off;      // byte offset of step i's output; op_off[ops.size()] = proof length (valid once offsets_known)
    bool offsets_known = false, cur_early = false, early_done = false;
    size_t early_slot = (size_t)-1;
    void push_op(std::function<void()> f) { ops.push_back(std::move(f)); op_early.push_back(cur_early ? 1 : 0); }
    static bool early_replay_on() { return true; }
    bool early_ready() const {
        if (early_done || !offsets_known || early_slot == (size_t)-1) return false;
        return __atomic_load_n(&ctx->h_res[early_slot].c0, __ATOMIC_ACQUIRE) == 1;
    }
    void run_early() {
        proof.bytes.resize(op_off[ops.size()]);
        for (size_t i = 0; i < ops.size(); i++)
            if (op_early[i]) { proof.pos = op_off[i]; ops[i](); }
        proof.pos = (size_t)-1;
        early_done = true;
    }
    size_t res_used = 0;
    int cls_gp_hash, cls_gp_base, cls_gp_ext, cls_gp_ext2, cls_col_ext2, cls_col_base, cls_col_ext, cls_ps, cls_ps2, cls_reduce, cls_aux, cls_tree, cls_hash, cls_gather, cls_tail, cls_ps_tail;

    // ---- single-proof sharding over `world` GPUs ---------------------------------------------------
    // Every rank walks the whole protocol (same challenges, same result slots) but only enqueues the work it owns; unowned
    // slots stay zero and ONE sum-all-reduce of the result buffer at the end gives every rank the complete buffer.
    //  * The Lasso node (most of the proof) is split BY MEMORY over all ranks: a rank runs the limb tables, the share of the
    //    claimed sum, of the collation sum-check and of both grand products, and the openings, of its own memories only. Those
    //    batched sum-checks are linear in their batch items, so the ranks' round sums are partial sums (lasso_node).
    //  * The Vanilla / FFT node reductions are independent jobs, dealt to the ranks longest-first on top of that load.
    int rank = 0, world = 1;
    std::vector<int> node_owner;       // Vanilla / FFT node reductions
    std::vector<int> gp1_owner;        // grand product #1 layers (all `rank`: every rank runs every layer on its memories)
    std::vector<int> gp1_mem_owner;    // world > 1: memory-GKR index i -> owning rank
    int own_out_claim = 0;
    bool mine(int owner) const { return owner == rank; }
    void plan_shards() {
        if (!pk) return;
        ShardPlan sp = shard_plan(pk, rank, world);
        node_owner = std::move(sp.node_owner);
        gp1_owner.assign(pk->lasso.nu, rank);
        gp1_mem_owner = std::move(sp.gp1_mem_owner);
        own_out_claim = sp.own_out_claim;
    }

    Prover(hg_ctx* c, const hg_pk* k, int rank_ = 0, int world_ = 1) : ctx(c), pk(k), st(c->stream), partials(c->d_partials), rank(rank_), world(world_) {
        ctx->prof_stream = st;
        plan_shards();
        cls_gp_ext2 = ctx->prof_class("sc_round2<grand_product,ext>", true);
        cls_gp_ext = ctx->prof_class("sc_round<grand_product,ext>", false);
        cls_gp_base = ctx->prof_class("sc_round<grand_product,base>", false);
        cls_gp_hash = ctx->prof_class("sc_round<grand_product,hash>", false);
        cls_col_base = ctx->prof_class("sc_round<collation,base>", false);
        cls_col_ext = ctx->prof_class("sc_round<collation,ext>", false);
        cls_col_ext2 = ctx->prof_class("sc_round2<collation,ext>", false);
        cls_ps = ctx->prof_class("sc_round<prodsum>", false);
        cls_ps2 = ctx->prof_class("sc_round2<prodsum>", false);
        cls_tail = ctx->prof_class("sc_tail<single-workgroup>", false);
        cls_ps_tail = ctx->prof_class("ps_tail<single-workgroup>", false);
        cls_reduce = ctx->prof_class("reduce_partials", false);
        cls_tree = ctx->prof_class("prod_level", false);
        cls_hash = ctx->prof_class("lasso_hash", false);
        cls_gather = ctx->prof_class("vanilla_gather", false);
        cls_aux = ctx->prof_class("aux", false);
        ctx->ensure_chain(16384);
        // arrival tickets of the last-workgroup reductions: a launch that died mid-way (fault, abort) would leave them non-zero
        // and every later prove on this context would silently lose round sums, so each prove starts from cleared tickets
        // (one kernel, together with the result-buffer prefix a sharded prove clears: memset nodes cost a launch each)
        dev::ClearSet cs;
        memset(&cs, 0, sizeof(cs));
        int nr = 0;
        for (E2* pbuf : {ctx->d_partials, ctx->d_partials2}) {
            cs.p[nr] = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(pbuf) + dev::PARTIALS_E2 * sizeof(E2));
            cs.n[nr++] = dev::PARTIALS_TICKETS;
        }
        if (world <= 1) {   // (one rank: the third region is free for the third stream's tickets)
            cs.p[2] = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ctx->d_partials3) + dev::PARTIALS_E2 * sizeof(E2));
            cs.n[2] = dev::PARTIALS_TICKETS;
        }
        if (world > 1 && pk) {
            // un-owned result slots must read zero. The buffer is host memory across PCIe: clear only what a prove of this key uses
            // (known from the previous walk of the same key; the first one clears everything)
            const bool hinted = ctx->res_hint_serial == pk->serial && ctx->res_hint > 0 && ctx->res_hint <= ctx->res_cap;
            cs.p[2] = reinterpret_cast<unsigned*>(ctx->d_res);
            cs.n[2] = (hinted ? ctx->res_hint : ctx->res_cap) * (sizeof(E2) / sizeof(unsigned));
        }
        dev::clear_words(ctx->stream, cs);
    }
    E2* d_res() { return ctx->d_res; }
    const E2* h_res() { return ctx->h_res; }
    size_t slot(size_t n) {
        if (res_used + n > ctx->res_cap) throw Error("result buffer exhausted");
        size_t s = res_used;
        res_used += n;
        return s;
    }
    size_t epos() const { return ch.pos / 2; }
    E2 squeeze() {
        E2 r = ch.squeeze();
        if (ch.pos / 2 > ctx->chal_e) throw Error("challenge chain in HBM too short");
        return r;
    }
    void reduce(int grid, int nv, size_t out_slot) {
        ctx->prof_begin(cls_reduce, 0);
        dev::reduce_partials(st, partials, grid, nv, d_res() + out_slot);
        ctx->prof_end();
    }

    // ---- sum-check drivers ---------------------------------------------------------------------
    // A sum-check whose remaining work is this small ((table pairs) x (pairs per table) items) finishes all
    // remaining rounds in one single-workgroup launch; anything larger is ALU-bound on a single CU.
    static constexpr size_t TAIL_ITEMS = 4096;   // (the first tail round of such a job does not fit the LDS regions and goes through HBM: still faster than one more launch, 2.93 vs 2.98 ms)

    // Stride-layout sum-checks (collation, every grand-product layer) are queued as jobs and executed by
    // flush_stride() in a round-synchronised schedule (launch k = every job's next round(s)): they are independent on the device.
    bool hash_recomp = false;   // the queued hash-source job recomputes its E values (lasso_node: lean form)
    std::vector<dev::StJob> st_jobs;
    // Grand-product jobs whose FIRST round also produces a product-tree level (or reads recomputed hashes) must run one after
    // the other, deepest layer first: job q with st_seq[q] = s > 0 gets its own first-round launch, in ascending s, before
    // the shared first-round launch of everything else; `st_after_seq` then builds the remaining (small) tree levels.
    std::vector<int> st_seq;
    std::vector<SlotPlan> st_slot;     // per queued job
    std::vector<int> st_credit_ntab;   // per queued job: table count of the reference's batch (traffic model of SURVEY.md 8(d)); = ntab without a shortcut
    std::vector<double> st_fused_bytes;  // algorithmic bytes of the passes a job's first round absorbs (hash build, tree level)
    std::vector<std::function<void()>> st_after_seq;
    // what a job's first round moves to or from HBM beside its own tables (hg_kernel_stat::hbm_bytes): the tree level it writes
    // (rows actually written x their length) and, for the hash-source job, the integer tables it reads INSTEAD of its input rows
    double pending_level_design = 0, pending_hash_reads = -1, hash_design_reads = 0;
    std::vector<double> st_level_design, st_hash_reads;
    double pending_fused_bytes = 0;      // set by the caller right before sc_stride (the hash build a hash-source job absorbs)
    double pending_fused_model_extra = 0, hash_model_extra = 0;   // ... and what only the reference's traffic model counts of it (E reads)
    std::vector<double> st_fused_model_extra;
    std::vector<dev::ScatterEnt> scatter;  // locally produced scalars -> global result slots (batch-subset grand products)

    ScHandle sc_stride(int kind, const void* in, bool base, size_t in_stride, int ntab, int nvars, const dev::Powers& pw, E2* final_out,
                       bool enqueue = true, bool p0_only = false, int seq = 0, u64* next_level = nullptr, const dev::GpHashSrc* hash_src = nullptr,
                       const MirrorSpec* mirror = nullptr, int model_ntab = 0, const SlotPlan* slots = nullptr) {
        ScHandle h;
        h.nv = kind == dev::SC_GRANDPROD ? 3 : 2;
        h.nvars = nvars;
        h.point_off = epos();
        h.sums_slot = slot((size_t)nvars * h.nv);
        const size_t N = (size_t)1 << nvars;
        if (!enqueue) {  // another rank runs this job: transcript bookkeeping only
            pending_fused_bytes = 0; pending_fused_model_extra = 0; pending_level_design = 0; pending_hash_reads = -1;
            for (int i = 0; i < nvars; i++) h.rs.push_back(squeeze());
            return h;
        }
        dev::StJob J;
        memset(&J, 0, sizeof(J));
        J.in = in; J.in_stride = in_stride;
        J.buf[0] = ctx->alloc_n<E2>((size_t)ntab * std::max<size_t>(N / 2, 1));
        J.buf[1] = ctx->alloc_n<E2>((size_t)ntab * std::max<size_t>(N / 4, 1));
        J.final_out = final_out ? final_out : ctx->alloc_n<E2>(ntab);
        J.kind = kind; J.ntab = ntab; J.nvars = nvars; J.base = base ? 1 : 0; J.p0_only = p0_only ? 1 : 0;
        J.r_off = h.point_off; J.sums_slot = h.sums_slot;
        J.next_level = next_level; J.hash_src = hash_src;
        if (mirror) { J.mirror = 1; J.mk1 = mirror->k1; J.mk2 = mirror->k2; }
        if (slots && slots->job_slotw) { J.slotw = slots->job_slotw; J.emit_mask = slots->job_emit; J.slot_ng = slots->npairs; J.slot_shift = slots->max_rd; }
        memcpy(J.pw, pw.v, sizeof(J.pw));
        for (int i = 0; i < nvars; i++) h.rs.push_back(squeeze());
        if (nvars > 0)  // weight * r_0: the first round stores the weighted fold (kernels.hip)
            for (int i = 0; i < dev::PW_MAX; i++) J.pwr[i] = e2_mul(pw.v[i], h.rs[0]);
        if (nvars > 0) {
            st_slot.push_back(slots ? *slots : SlotPlan());
            st_jobs.push_back(J); st_seq.push_back(seq); st_credit_ntab.push_back(mirror ? mirror->credit_ntab : (model_ntab ? model_ntab : ntab));
            // a level-writing first round replaces prod_level on its input level: (nb rows of 2N entries) x 8 B x 1.5 (read +
            // write), as prod_level is credited; the hash-source job's level 1 is credited with its write only, as the hash kernel
            // it replaces was (round-1 accounting: the totals stay comparable)
            double fused = next_level ? (double)(st_credit_ntab.back() / 2) * (double)(2 * N) * 8.0 * (hash_src ? 0.5 : 1.5) : 0.0;
            st_fused_bytes.push_back(fused + pending_fused_bytes);
            st_fused_model_extra.push_back(pending_fused_model_extra);
            st_level_design.push_back(next_level ? pending_level_design : 0.0);
            st_hash_reads.push_back(hash_src ? pending_hash_reads : -1.0);
            pending_fused_bytes = 0; pending_fused_model_extra = 0; pending_level_design = 0; pending_hash_reads = -1;
        }
        return h;
    }

    // half-length (log2) at which a slot-form job's tail starts: what its PER-MEMORY tables allow (HG_TAIL_H does not apply)
    static int slot_tail_h(int tail_ntab, int nvars) { return std::min(dev::st_tail_h(tail_ntab, nvars), nvars - 2); }
    void flush_stride() {
        if (st_jobs.empty()) return;
        const int nj = (int)st_jobs.size();
        // launch plan: (kind, base? | fused pair?, h_log2 | tail) -> items (job, where the round reads and writes).
        // Folded tables ping-pong between the job's two buffers; the host tracks where each job's live tables are.
        constexpr int fuse_min_h = 15;   // fused pairs of rounds from half = 2^15 up (13 until the slot form: 1.96-1.99 against 1.99-2.01 ms; round 5: 13 / 11 / 9 = 1.864 / 1.905 / 1.980 against 1.831)
        struct Launch { int kind; bool base; int h_log2; bool tail; int nrounds; std::vector<dev::StItem> items; bool hash = false; bool after_seq = false; };
        std::vector<Launch> plan;
        struct Regroup { int job; const E2* in; E2* out; int len_log2; };
        std::vector<Regroup> regroups;
        std::vector<const void*> cur_in(nj);
        std::vector<size_t> cur_stride(nj);
        std::vector<int> next_h(nj);  // half-length (log2) of the job's next unscheduled round
        for (int q = 0; q < nj; q++) { cur_in[q] = st_jobs[q].in; cur_stride[q] = st_jobs[q].in_stride; next_h[q] = st_jobs[q].nvars - 1; }
        auto next_out = [&](int q) { return cur_in[q] == (const void*)st_jobs[q].buf[0] ? st_jobs[q].buf[1] : st_jobs[q].buf[0]; };
        // The rounds with half <= 2^h_small[q] of job q run in ONE single-workgroup launch with the folded tables in LDS (st_tail);
        // how many fit depends on the job's table count (12 rounds for the two collation tables, 6 for a full grand-product layer).
        std::vector<int> h_small(nj);
        for (int q = 0; q < nj; q++) {
            const SlotPlan& sp = st_slot[q];
            h_small[q] = sp.tail_ntab ? slot_tail_h(sp.tail_ntab, st_jobs[q].nvars) : dev::st_tail_h(st_jobs[q].ntab, st_jobs[q].nvars);
            if (st_seq[q] > 0) h_small[q] = std::min(h_small[q], st_jobs[q].nvars - 2);   // a sequenced first round has its own kernel
            if (sp.tail_ntab && st_jobs[q].nvars - 1 - h_small[q] > sp.max_rd) throw Error("slot-form job: the tail starts below the segment pairs");
        }
        for (int kind : {dev::SC_COLLATION, dev::SC_GRANDPROD}) {
            int max_h = -1;
            for (auto& J : st_jobs) if (J.kind == kind) max_h = std::max(max_h, J.nvars - 1);
            if (max_h < 0) continue;
            // sequenced first rounds (each produces the tree level the next one reads), then the remaining tree levels
            if (kind == dev::SC_GRANDPROD) {
                int max_seq = 0;
                for (int q = 0; q < nj; q++) max_seq = std::max(max_seq, st_seq[q]);
                for (int sq = 1; sq <= max_seq; sq++)
                    for (int q = 0; q < nj; q++) {
                        const dev::StJob& J = st_jobs[q];
                        if (st_seq[q] != sq || J.kind != kind) continue;
                        if (!J.base || next_h[q] <= h_small[q]) throw Error("sequenced first round on a job that has none");
                        Launch ls{kind, true, -1, false, 1, {}};
                        ls.hash = J.hash_src != nullptr;
                        dev::StItem it;
                        memset(&it, 0, sizeof(it));
                        it.job = q; it.h_log2 = next_h[q]; it.in = cur_in[q]; it.in_stride = cur_stride[q]; it.out = J.buf[0];
                        ls.items.push_back(it);
                        cur_in[q] = it.out; cur_stride[q] = (size_t)1 << next_h[q];
                        next_h[q]--;
                        plan.push_back(ls);
                    }
                if (max_seq > 0 || !st_after_seq.empty()) { Launch la{kind, true, -1, false, 0, {}}; la.after_seq = true; plan.push_back(la); }
            }
            // the first rounds on base-field rows only read finished tree levels / node tables: one launch for all of them
            {
                Launch lall{kind, true, -1, false, 1, {}};
                for (int q = 0; q < nj; q++) {
                    const dev::StJob& J = st_jobs[q];
                    if (J.kind != kind || !J.base || next_h[q] <= h_small[q] || next_h[q] != J.nvars - 1) continue;  // (sequenced jobs are past their first round)
                    dev::StItem it;
                    memset(&it, 0, sizeof(it));
                    it.job = q; it.h_log2 = next_h[q]; it.in = cur_in[q]; it.in_stride = cur_stride[q]; it.out = J.buf[0];
                    lall.items.push_back(it);
                    cur_in[q] = it.out; cur_stride[q] = (size_t)1 << next_h[q];
                    next_h[q]--;
                }
                if (!lall.items.empty()) plan.push_back(lall);
            }
            // then round-synchronised: launch k runs every job's next round (or, grand product with a long enough table,
            // its next TWO rounds) whatever the sizes; the jobs only depend on their own previous launch
            for (;;) {
                Launch le{kind, false, -1, false, 1, {}}, l2{kind, false, -1, false, 2, {}};
                for (int q = 0; q < nj; q++) {
                    const dev::StJob& J = st_jobs[q];
                    const int h = next_h[q];
                    if (J.kind != kind || h <= h_small[q]) continue;
                    const bool first = J.nvars - 1 == h;
                    if (first && J.hash_src) throw Error("hash-source job without a sequenced first round");
                    dev::StItem it;
                    memset(&it, 0, sizeof(it));
                    it.job = q; it.h_log2 = h; it.in = cur_in[q]; it.in_stride = cur_stride[q];
                    it.out = first ? J.buf[0] : next_out(q);
                    const bool pair = !first && h - 1 > h_small[q] && h >= std::max(dev::ST_STEP2_MIN_H, fuse_min_h);
                    (pair ? l2 : le).items.push_back(it);
                    next_h[q] = h - (pair ? 2 : 1);
                    cur_in[q] = it.out; cur_stride[q] = (size_t)1 << (pair ? h - 1 : h);
                }
                if (le.items.empty() && l2.items.empty()) break;
                if (!l2.items.empty()) plan.push_back(l2);
                if (!le.items.empty()) plan.push_back(le);
            }
            // the tail launch: every job's remaining rounds
            {
                Launch lc{kind, false, 0, true, 0, {}};
                for (int q = 0; q < nj; q++) {
                    const dev::StJob& J = st_jobs[q];
                    if (J.kind != kind || next_h[q] < 0) continue;
                    const int hs = next_h[q];
                    dev::StItem it;
                    memset(&it, 0, sizeof(it));
                    it.job = q; it.in = cur_in[q]; it.in_stride = cur_stride[q];
                    it.rd = J.nvars - 1 - hs; it.nrounds = hs + 1; it.out = J.final_out;
                    if (st_slot[q].tail_ntab) {   // the tail reads the per-memory tables (gathered right before it is launched)
                        const SlotPlan& sp = st_slot[q];
                        if (cur_stride[q] != (size_t)2 << hs) throw Error("slot-form job: unexpected table length at the tail");
                        E2* rg = ctx->alloc_n<E2>((size_t)sp.tail_ntab << (hs + 1));
                        regroups.push_back(Regroup{q, (const E2*)cur_in[q], rg, hs + 1});
                        it.in = rg; it.ntab = sp.tail_ntab;
                    }
                    lc.items.push_back(it);
                    next_h[q] = -1;
                }
                if (!lc.items.empty()) plan.push_back(lc);
            }
        }
        dev::StJob* d_jobs = ctx->alloc_n<dev::StJob>(nj);
        upload(d_jobs, st_jobs.data(), (size_t)nj * sizeof(dev::StJob), "upload jobs");
        std::vector<dev::StItem> flat;
        std::vector<size_t> offs;
        std::vector<std::vector<int>> grids(plan.size());
        for (size_t li = 0; li < plan.size(); li++) {
            Launch& L = plan[li];
            if (!L.tail && !L.after_seq)
                for (size_t o = 0; o < L.items.size(); o += MAX_BATCH)
                    grids[li].push_back(dev::st_plan_blocks(L.items.data() + o, (int)std::min<size_t>(MAX_BATCH, L.items.size() - o), L.nrounds == 2));
            offs.push_back(flat.size());
            flat.insert(flat.end(), L.items.begin(), L.items.end());
        }
        dev::StItem* d_items = ctx->alloc_n<dev::StItem>(flat.size());
        upload(d_items, flat.data(), flat.size() * sizeof(dev::StItem), "upload step items");
        // algorithmic bytes of one round of job q: every live table read once, every folded table written once.
        // model = true: the tables of the reference's batch (a mirrored grand product is modelled with its read AND write tables, the
        // collation sum-check with its alpha tables), false: the tables this implementation streams
        auto round_bytes = [&](int q, int rd, bool model) {
            const dev::StJob& J = st_jobs[q];
            size_t half = (size_t)1 << (J.nvars - 1 - rd);
            return (double)(model ? st_credit_ntab[q] : J.ntab) * (2.0 * half * ((J.base && rd == 0) ? 8 : 16) + half * 16.0);
        };
        // the same launch in bytes moved to or from HBM by design (hg_kernel_stat::hbm_bytes): the tables of round rd read once, the
        // folds of round rd + nrounds - 1 written once (the tail: nothing written), plus the tree level a first round emits
        auto design_bytes = [&](int q, int rd, int nrounds, bool tail) {
            const dev::StJob& J = st_jobs[q];
            const size_t half = (size_t)1 << (J.nvars - 1 - rd);
            double b = (double)J.ntab * 2.0 * half * ((J.base && rd == 0) ? 8 : 16);
            if (rd == 0 && st_hash_reads[q] >= 0) b = st_hash_reads[q];
            if (!tail) b += (double)J.ntab * (double)(half >> (nrounds - 1)) * 16.0;
            if (rd == 0) b += st_level_design[q];
            return b;
        };
        for (size_t li = 0; li < plan.size(); li++) {
            const Launch& L = plan[li];
            if (L.kind == dev::SC_GRANDPROD && st_before_gp) { stamp("collation done"); st_before_gp(); st_before_gp = nullptr; stamp("grand products may start"); }
            if (L.after_seq) {
                for (auto& f : st_after_seq) f();
                st_after_seq.clear();
                if (st_before_gp2) { st_before_gp2(); st_before_gp2 = nullptr; }   // the launches from here on contain grand product #2's jobs
                continue;
            }
            for (size_t o = 0; o < L.items.size(); o += MAX_BATCH) {
                const int cnt = (int)std::min<size_t>(MAX_BATCH, L.items.size() - o);
                double bytes = 0, model = 0;
                if (L.hash) {
                    const dev::StItem& it = L.items[o];
                    // algorithmic bytes: the sum-check round plus the passes this launch absorbs - the hash build
                    // (dims, read_ts per chunk; E read, read / write hashes written per memory) and product-tree level 1
                    bytes = round_bytes(it.job, 0, false) + st_fused_bytes[it.job];
                    model = round_bytes(it.job, 0, true) + st_fused_bytes[it.job] + st_fused_model_extra[it.job];
                    ctx->prof_begin(cls_gp_hash, bytes, model, design_bytes(it.job, 0, 1, false));
                    dev::st_first_hash(st, d_jobs + it.job, d_items + offs[li] + o, grids[li][0], st_jobs[it.job].mirror != 0, hash_recomp, ctx->d_chal, partials, d_res(), st_slot[it.job].tail_ntab != 0);
                    ctx->prof_end();
                    stamp("first hash round done");
                    continue;
                }
                if (L.tail) {
                    size_t table_bytes = 0;
                    double design = 0;
                    for (int q = 0; q < cnt; q++) {
                        const dev::StItem& it = L.items[o + q];
                        const dev::StJob& J = st_jobs[it.job];
                        for (int k = 0; k < it.nrounds; k++) { bytes += round_bytes(it.job, it.rd + k, false); model += round_bytes(it.job, it.rd + k, true); }
                        design += (double)(it.ntab > 0 ? it.ntab : J.ntab) * 2.0 * (double)((size_t)1 << (J.nvars - 1 - it.rd)) * ((J.base && it.rd == 0) ? 8 : 16);
                        const int h0 = J.nvars - 1 - it.rd;
                        table_bytes = std::max(table_bytes, (size_t)(it.ntab > 0 ? it.ntab : J.ntab) * (((size_t)1 << h0) + (((size_t)1 << h0) >> 1)) * sizeof(E2));
                        for (const Regroup& g : regroups) if (g.job == it.job) {
                            const SlotPlan& sp = st_slot[it.job];
                            dev::gp_slot_regroup(st, g.in, g.out, sp.d_slot_of, sp.d_ratio, sp.nrows, sp.nslots, sp.npairs, g.len_log2, (sp.tail_ntab & 1) != 0);
                        }
                    }
                    ctx->prof_begin(cls_tail, bytes, model, design);
                    dev::st_tail(st, L.kind, d_jobs, d_items + offs[li] + o, cnt, table_bytes, ctx->d_chal, d_res());
                    ctx->prof_end();
                } else {
                    double design = 0;
                    for (int q = 0; q < cnt; q++) {
                        const dev::StItem& it = L.items[o + q];
                        const dev::StJob& J = st_jobs[it.job];
                        design += design_bytes(it.job, J.nvars - 1 - it.h_log2, L.nrounds, false);
                        // algorithmic bytes in the per-round accounting of SURVEY.md 8(d): a fused launch is credited with both of
                        // its rounds although the intermediate folded tables never reach HBM
                        for (int k = 0; k < L.nrounds; k++) { bytes += round_bytes(it.job, J.nvars - 1 - it.h_log2 + k, false); model += round_bytes(it.job, J.nvars - 1 - it.h_log2 + k, true); }
                        if (L.base && it.h_log2 == J.nvars - 1) { bytes += st_fused_bytes[it.job]; model += st_fused_bytes[it.job]; }  // the tree level a first round also writes
                    }
                    const int grid = grids[li][o / MAX_BATCH];
                    int cls = L.kind == dev::SC_GRANDPROD ? (L.base ? cls_gp_base : (L.nrounds == 2 ? cls_gp_ext2 : cls_gp_ext)) : (L.base ? cls_col_base : (L.nrounds == 2 ? cls_col_ext2 : cls_col_ext));
                    ctx->prof_begin(cls, bytes, model, design);
                    if (L.nrounds == 2) dev::st_step2(st, L.kind, d_jobs, d_items + offs[li] + o, cnt, grid, ctx->d_chal, partials, d_res());
                    else dev::st_step(st, L.kind, L.base, d_jobs, d_items + offs[li] + o, cnt, grid, ctx->d_chal, partials, d_res(),
                                      L.base && st_jobs[L.items[o].job].slotw != nullptr);
                    ctx->prof_end();
                }
            }
        }
        if (st_before_gp) { st_before_gp(); st_before_gp = nullptr; }
        if (st_before_gp2) { st_before_gp2(); st_before_gp2 = nullptr; }
        st_jobs.clear();
        st_seq.clear();
        st_credit_ntab.clear();
        st_slot.clear();
        st_fused_bytes.clear();
        st_fused_model_extra.clear();
        st_level_design.clear();
        st_hash_reads.clear();
        for (auto& f : st_after_seq) f();  // (no grand-product job was queued: nothing can depend on these, but keep the order)
        st_after_seq.clear();
        if (!scatter.empty()) {
            dev::ScatterEnt* d = ctx->alloc_n<dev::ScatterEnt>(scatter.size());
            upload(d, scatter.data(), scatter.size() * sizeof(dev::ScatterEnt), "upload scatter list");
            dev::scatter_e2(st, d, scatter.size(), d_res());
            scatter.clear();
        }
    }

    // PRODSUM instances are queued and executed in batches of equal nvars (grid.y = instance): the node
    // reductions have no device-side dependencies on each other, only the transcript order matters.
    std::map<int, std::vector<dev::PsJob>> ps_queue;
    std::vector<std::function<void()>> second_wave;  // device work that needs first-wave results (Libra phase 2)

    ScHandle sc_prodsum(const std::vector<const u64*>& a, const std::vector<const E2*>& b, int nvars,
                        const std::vector<E2*>& fin_a, const std::vector<E2*>& fin_b, bool enqueue = true) {
        ScHandle h;
        h.nv = 2;
        h.nvars = nvars;
        h.point_off = epos();
        h.sums_slot = slot((size_t)nvars * 2);
        if (!enqueue) {
            for (int i = 0; i < nvars; i++) h.rs.push_back(squeeze());
            return h;
        }
        const int np = (int)a.size();
        if (np > dev::PS_MAX_PAIRS) throw Error("prodsum: too many table pairs");
        const size_t N = (size_t)1 << nvars;
        dev::PsJob J;
        memset(&J, 0, sizeof(J));
        J.npairs = np; J.nvars = nvars; J.r_off = h.point_off; J.sums_slot = h.sums_slot;
        for (int q = 0; q < 2; q++) {
            J.bufa[q] = ctx->alloc_n<E2>((size_t)np * std::max<size_t>(N >> (q + 1), 1));
            J.bufb[q] = ctx->alloc_n<E2>((size_t)np * std::max<size_t>(N >> (q + 1), 1));
        }
        for (int i = 0; i < np; i++) { J.a[i] = a[i]; J.b[i] = b[i]; J.fin_a[i] = fin_a[i]; J.fin_b[i] = fin_b[i]; }
        for (int i = 0; i < nvars; i++) h.rs.push_back(squeeze());
        ps_queue[nvars].push_back(J);
        return h;
    }
    // ---- eq-factored PRODSUM jobs (kernels.hpp PsJob::eq_n) ---------------------------------------------------------------
    // HG_NO_PS_EQ=1: every Libra table materialised (the general form, which small or non-affine nodes take anyway)
    static bool ps_eq_on() { static const bool v = !hg_env_on("HG_NO_PS_EQ"); return v; }
    static size_t ps_tail_items() { return std::min<size_t>(TAIL_ITEMS, dev::ps_tail_items_max()); }   // (the tail keeps a job's folds in LDS)
    // First tail round of an eq-factored job, -1: the job is too small for the form. Every round ahead of the tail runs in a fused
    // pair (the tail may start one round later than TAIL_ITEMS says), the last pair at half >= 2^9, and the table handed to the
    // tail must be one of the point's stored suffix tables.
    static int eq_tail_rd(int npairs, int nvars) {
        if (npairs < 1 || npairs > dev::PS_MAX_PAIRS || nvars > dev::PS_EQ_MAX_VARS) return -1;
        const size_t N = (size_t)1 << nvars;
        int rd = 0;
        while (rd < nvars && ((N >> rd) / 2) * (size_t)npairs > ps_tail_items()) rd++;
        if (rd & 1) rd++;
        if (rd < 2 || rd > nvars - 8 || rd < dev::ps_eq_kmin(nvars)) return -1;
        return rd;
    }
    std::vector<E2> eq_scal_host;   // prefactors / kappa of the queued eq-factored jobs (PsJob::eq_scal), one upload per flush
    std::vector<dev::PsEqPoint> eqpt_queue;
    std::map<std::tuple<size_t, int, int, unsigned>, std::pair<E2*, E2*>> eqpt_shared;   // (point, w, nvars, hib) -> the point's (lo, suf) tables
    // g = sum_i a_i(x) kappa_i eq(z', x): zp = z' (host copy of the coordinates), the first w of them the chain run at z_off
    ScHandle sc_prodsum_eq(const std::vector<const u64*>& a, const std::vector<E2>& kappa, const std::vector<E2>& zp, size_t z_off, int w, unsigned hib,
                           int nvars, int tail_rd, const std::vector<E2*>& fin_a, const std::vector<E2*>& fin_b, bool enqueue) {
        ScHandle h;
        h.nv = 2;
        h.nvars = nvars;
        h.point_off = epos();
        h.sums_slot = slot((size_t)nvars * 2);
        for (int i = 0; i < nvars; i++) h.rs.push_back(squeeze());
        if (!enqueue) return h;
        const int np = (int)a.size();
        const size_t N = (size_t)1 << nvars;
        dev::PsJob J;
        memset(&J, 0, sizeof(J));
        J.npairs = np; J.nvars = nvars; J.r_off = h.point_off; J.sums_slot = h.sums_slot;
        J.tail_rd = tail_rd; J.eq_n = np; J.eq_single = np == 1 ? 1 : 0;
        for (int q = 0; q < 2; q++) {
            J.bufa[q] = ctx->alloc_n<E2>((size_t)np * (N >> (q + 1)));
            if (np > 1) J.bufA[q] = ctx->alloc_n<E2>(N >> (q + 1));
        }
        if (np > 1) J.eqA0 = ctx->alloc_n<E2>(N);
        for (int i = 0; i < np; i++) { J.a[i] = a[i]; J.fin_a[i] = fin_a[i]; J.fin_b[i] = fin_b[i]; }
        // prefactors of the rounds ahead of the tail, P at the hand-off, kappa
        std::vector<E2> scal((size_t)2 * nvars + 1 + np + nvars + 4 * (size_t)(tail_rd / 2), e2_zero());
        E2 P = e2_one();
        for (int rd = 0; rd <= tail_rd && rd < nvars; rd++) {
            if (rd == tail_rd) { scal[(size_t)2 * nvars] = P; break; }
            const E2 z = zp[rd], omz = e2_sub(e2_one(), z);
            E2 p0 = e2_mul(P, omz), p2 = e2_mul(P, e2_sub(e2_add(e2_dbl(z), z), e2_one()));
            if (np == 1) { p0 = e2_mul(p0, kappa[0]); p2 = e2_mul(p2, kappa[0]); }
            scal[(size_t)2 * rd] = p0; scal[(size_t)2 * rd + 1] = p2;
            P = e2_mul(P, e2_add(e2_mul(z, h.rs[rd]), e2_mul(omz, e2_sub(e2_one(), h.rs[rd]))));
        }
        for (int i = 0; i < np; i++) scal[(size_t)2 * nvars + 1 + i] = kappa[i];
        for (int k = 0; k < nvars; k++) scal[(size_t)2 * nvars + 1 + np + k] = zp[k];
        for (int rd = 0; rd + 1 < tail_rd; rd += 2) {   // the four-term double fold of the pass at (rd, rd + 1): entry 4j + p, p = b0 + 2 b1
            const E2 ra = h.rs[rd], rb = h.rs[rd + 1], na = e2_sub(e2_one(), ra), nb = e2_sub(e2_one(), rb);
            E2* c = &scal[(size_t)2 * nvars + 1 + np + nvars + 4 * (size_t)(rd / 2)];
            c[0] = e2_mul(na, nb); c[1] = e2_mul(ra, nb); c[2] = e2_mul(na, rb); c[3] = e2_mul(ra, rb);
        }
        // (uploaded by flush_prodsum on the stream the rounds run on - this walk is still enqueueing to the main one; until then
        // eq_scal holds the job's offset into eq_scal_host)
        J.eq_scal = reinterpret_cast<const E2*>(eq_scal_host.size() * sizeof(E2));
        eq_scal_host.insert(eq_scal_host.end(), scal.begin(), scal.end());
        const auto key = std::make_tuple(z_off, w, nvars, hib);
        auto hit = eqpt_shared.find(key);
        if (hit == eqpt_shared.end()) {
            dev::PsEqPoint pt;
            memset(&pt, 0, sizeof(pt));
            pt.point_off = z_off; pt.w = w; pt.nvars = nvars; pt.hib = hib; pt.kmin = dev::ps_eq_kmin(nvars);
            pt.lo = ctx->alloc_n<E2>(dev::ps_eq_lo_entries(nvars));
            pt.suf = ctx->alloc_n<E2>(dev::ps_eq_suf_entries(nvars));
            eqpt_queue.push_back(pt);
            hit = eqpt_shared.emplace(key, std::make_pair(pt.lo, pt.suf)).first;
        }
        J.eq_lo = hit->second.first; J.eq_suf = hit->second.second;
        ps_queue[nvars].push_back(J);
        return h;
    }
    void flush_prodsum() {
        // round-synchronised: launch rd runs round rd of every queued job whatever its size; the last rounds of each
        // job (TAIL_ITEMS work items or fewer) run in one single-workgroup-per-job launch
        std::vector<dev::PsJob> jobs;
        for (auto& kv : ps_queue) { jobs.insert(jobs.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
        if (jobs.empty()) return;
        const int nj = (int)jobs.size();
        if (hg_debug("eq")) {
            int ne = 0; size_t ee = 0, et = 0;
            for (auto& J : jobs) { const size_t e = (size_t)J.npairs << J.nvars; et += e; if (J.eq_n) { ne++; ee += e; } }
            fprintf(stderr, "[hg eq] %d of %d queued node reductions eq-factored, %zu of %zu table entries\n", ne, nj, ee, et);
        }
        if (!eq_scal_host.empty()) {
            E2* d_scal = ctx->alloc_n<E2>(eq_scal_host.size());
            upload(d_scal, eq_scal_host.data(), eq_scal_host.size() * sizeof(E2), "upload eq-form scalars");
            for (auto& J : jobs) if (J.eq_n) J.eq_scal = d_scal + reinterpret_cast<size_t>(J.eq_scal) / sizeof(E2);
            eq_scal_host.clear();
        }
        int max_rd = 0;
        for (auto& J : jobs) {
            const size_t N = (size_t)1 << J.nvars;
            int rd = 0;
            if (J.eq_n) rd = J.tail_rd;   // (planned when the job was queued: eq_tail_rd)
            else while (rd < J.nvars && ((N >> rd) / 2) * (size_t)J.npairs > ps_tail_items()) rd++;
            J.tail_rd = rd;
            max_rd = std::max(max_rd, rd);
        }
        auto round_bytes = [&](const dev::PsJob& J, int rd) {
            size_t half = ((size_t)1 << J.nvars) >> (rd + 1);
            if (J.eq_n && rd < J.tail_rd)   // no b tables; the folded A beside the a_i (formed in round 0, not read)
                return (double)J.npairs * (2.0 * half * (rd == 0 ? 8 : 16) + half * 16.0) + (J.eq_single ? 0.0 : (rd == 0 ? 0.0 : 2.0 * half * 16) + half * 16.0);
            return (double)J.npairs * (2.0 * half * ((rd == 0 ? 8 : 16) + 16) + half * 32.0);
        };
        // plan every step first (step s = every job's next round, or its next two rounds when its table is long
        // enough), upload all items in one copy, then launch; the host tracks each job's ping-pong buffer
        constexpr int ps_fuse_min_h = 9;   // fused pairs from half = 2^9 (11 until the node reductions moved to the third stream: 1.92-1.94 against 1.96 ms)
        // ... and in bytes moved to or from HBM by design: a pass reads its tables once and writes the folds of its LAST round
        auto design_bytes = [&](const dev::PsJob& J, int rd, int nrounds) {
            const size_t half = ((size_t)1 << J.nvars) >> (rd + 1), out = half >> (nrounds - 1);
            const double ea = rd == 0 ? 8 : 16;
            if (J.eq_n) return (double)J.npairs * (2.0 * half * ea + out * 16.0) + (J.eq_single ? 0.0 : 2.0 * half * 16.0 + out * 16.0);
            return (double)J.npairs * (2.0 * half * (ea + 16) + out * 32.0);
        };
        struct PsLaunch { bool two, eq; int cnt, grid; size_t off; double bytes, design; };
        std::vector<PsLaunch> launches;
        std::vector<dev::PsItem> all_items;
        std::vector<int> next_rd(nj, 0), cur_buf(nj, -1);
        for (;;) {
            std::vector<dev::PsItem> one, two, two_eq;   // (eq-factored jobs have a kernel of their own)
            for (int q = 0; q < nj; q++) {
                const dev::PsJob& J = jobs[q];
                const int rd = next_rd[q];
                if (rd >= J.tail_rd) continue;
                const int h = J.nvars - 1 - rd;
                const bool pair = J.eq_n ? true : rd + 1 < J.tail_rd && h >= std::max(9, ps_fuse_min_h);
                dev::PsItem it;
                memset(&it, 0, sizeof(it));
                it.job = q; it.rd = rd; it.in_buf = cur_buf[q]; it.out_buf = cur_buf[q] == 1 ? 0 : 1;
                if (cur_buf[q] < 0) it.out_buf = pair ? 1 : 0;  // sizes: bufa[0] holds N/2 entries per table, bufa[1] N/4
                if (J.eq_n && rd + 2 >= J.tail_rd) it.pad = 1;   // (PS_EQ_OUT_TAIL: this pass writes the layout the tail reads)
                (J.eq_n ? two_eq : pair ? two : one).push_back(it);
                next_rd[q] = rd + (pair ? 2 : 1);
                cur_buf[q] = it.out_buf;
            }
            if (one.empty() && two.empty() && two_eq.empty()) break;
            for (int kind = 0; kind < 3; kind++) {   // the eq-factored jobs first: the largest tables of the first wave
                std::vector<dev::PsItem>& items = kind == 0 ? two_eq : kind == 1 ? one : two;
                const bool fused = kind != 1;
                for (size_t o = 0; o < items.size(); o += MAX_BATCH) {
                    const int cnt = (int)std::min<size_t>(MAX_BATCH, items.size() - o);
                    const int grid = dev::ps_plan_blocks(items.data() + o, cnt, jobs.data(), fused);
                    double bytes = 0, design = 0;
                    for (int q = 0; q < cnt; q++) {
                        for (int k = 0; k <= (fused ? 1 : 0); k++) bytes += round_bytes(jobs[items[o + q].job], items[o + q].rd + k);
                        design += design_bytes(jobs[items[o + q].job], items[o + q].rd, fused ? 2 : 1);
                    }
                    launches.push_back({fused, kind == 0, cnt, grid, all_items.size(), bytes, design});
                    all_items.insert(all_items.end(), items.begin() + o, items.begin() + o + cnt);
                }
            }
        }
        for (int q = 0; q < nj; q++) jobs[q].tail_buf = cur_buf[q];
        dev::PsJob* d_jobs = ctx->alloc_n<dev::PsJob>(nj);
        upload(d_jobs, jobs.data(), (size_t)nj * sizeof(dev::PsJob), "upload jobs");
        {   // A = sum_i kappa_i a_i of the eq-factored jobs with several tables
            std::vector<int> ids;
            size_t max_quads = 0; double ab = 0;
            for (int q = 0; q < nj; q++) if (jobs[q].eq_n > 1) {
                ids.push_back(q);
                const size_t N = (size_t)1 << jobs[q].nvars;
                max_quads = std::max(max_quads, N >> 2);
                ab += (double)N * (8.0 * jobs[q].eq_n + 16.0);
            }
            if (!ids.empty()) {
                int* d_ids = ctx->alloc_n<int>(ids.size());
                upload(d_ids, ids.data(), ids.size() * sizeof(int), "upload eq-form job list");
                ctx->prof_begin(cls_aux, ab);
                dev::ps_eq_A(st, d_jobs, d_ids, (int)ids.size(), max_quads);
                ctx->prof_end();
            }
        }
        if (!all_items.empty()) {
            dev::PsItem* d_items = ctx->alloc_n<dev::PsItem>(all_items.size());
            upload(d_items, all_items.data(), all_items.size() * sizeof(dev::PsItem), "upload items");
            for (auto& L : launches) {
                ctx->prof_begin(L.two ? cls_ps2 : cls_ps, L.bytes, -1.0, L.design);
                dev::ps_round(st, L.two, d_jobs, d_items + L.off, L.cnt, L.grid, ctx->d_chal, partials, d_res(), L.eq);
                ctx->prof_end();
            }
        }
        double tb = 0, td = 0;
        for (auto& J : jobs) {
            for (int rd = J.tail_rd; rd < J.nvars; rd++) tb += round_bytes(J, rd);
            if (J.tail_rd < J.nvars) {   // the tail reads its first round's tables (an eq-factored job: the a tables and a suffix table), the rest runs in LDS
                const size_t half = ((size_t)1 << J.nvars) >> (J.tail_rd + 1);
                td += (double)J.npairs * 2.0 * half * (J.tail_rd == 0 ? 8 : 16) + (J.eq_n ? 2.0 * half * 16 : (double)J.npairs * 2.0 * half * 16);
            }
        }
        ctx->prof_begin(cls_ps_tail, tb, -1.0, td);
        dev::ps_tail(st, d_jobs, nj, ctx->d_chal, d_res());
        ctx->prof_end();
    }
    static constexpr size_t MAX_BATCH = 64;
    // pinned staging for small host->device descriptor copies (kept alive until the final synchronisation)
    // small host->device descriptor copy through the staging buffer, on the current stream
    // Descriptor uploads (jobs, items, hash sources ...). Their contents depend on the key and the share only - the challenges are
    // known up front - so a prove that is being recorded into a launch graph does not record them: they are copied ONCE before the
    // first replay (prove_capture) and stay in the arena, which nothing else touches while the cached graph is valid (arena_epoch).
    // As graph nodes they cost about 5 us each on the stream they sit on, ten to twenty per prove.
    bool defer_uploads = false;
    struct Upload { void* dst; const void* src; size_t bytes; };
    std::vector<Upload> deferred_uploads;
    void upload(void* dst, const void* src, size_t bytes, const char* what) {
        const void* staged = stage(src, bytes);
        if (defer_uploads) { deferred_uploads.push_back(Upload{dst, staged, bytes}); return; }
        hip_check(hipMemcpyAsync(dst, staged, bytes, hipMemcpyHostToDevice, st), what);
    }
    void* stage(const void* src, size_t bytes) {
        size_t need = (bytes + 63) & ~(size_t)63;
        if (ctx->stage_used + need > ctx->stage_cap) throw Error("staging buffer exhausted");
        void* p = ctx->h_stage + ctx->stage_used;
        ctx->stage_used += need;
        memcpy(p, src, bytes);
        return p;
    }

    // transcript side of prove_sum_check: d+1 coefficients per round, eval(1) derived from the running claim
    void defer_sumcheck(const ScHandle& h, int deg, Cell claim_in, Cell claim_out) {
        push_op([this, h, deg, claim_in, claim_out] {
            E2 claim = *claim_in;
            if (h.scaled)   // mirrored grand product: the kernels summed everything but the common factor 1 + kappa
                for (size_t q = 0; q < (size_t)h.nvars * h.nv; q++) ctx->h_res[h.sums_slot + q] = e2_mul(ctx->h_res[h.sums_slot + q], h.scale);
            for (int i = 0; i < h.nvars; i++) {
                const E2* s = h_res() + h.sums_slot + (size_t)i * h.nv;
                E2 ev[4], c[4];
                ev[0] = s[0];
                ev[1] = e2_sub(claim, s[0]);
                ev[2] = s[1];
                if (deg == 3) ev[3] = s[2];
                interpolate(ev, deg, c);
                for (int k = 0; k <= deg; k++) proof.write_e(c[k]);
                claim = horner(c, deg, h.rs[i]);
            }
            if (claim_out) *claim_out = claim;
        });
    }
    // the grand-product kernels leave the final LEFT evaluation of pair b multiplied by pw[b] (see kernels.hip)
    void defer_gp_unscale(size_t evals_slot, int nb, const dev::Powers& pw) {
        if (nb < 2) return;
        if (pw.v[1].c0 == 0 && pw.v[1].c1 == 0) throw Error("grand product: zero batching weight");
        // the inverse weights depend on the challenges only: computed here, during the walk (which a cached launch graph does not
        // repeat), not in the replay that every prove runs after its synchronisation
        auto winv = std::make_shared<std::vector<E2>>(nb);
        const E2 ginv = e2_inv(pw.v[1]);  // pw[b] = gamma^b
        E2 w = ginv;
        for (int b = 1; b < nb; b++) { (*winv)[b] = w; w = e2_mul(w, ginv); }
        push_op([this, evals_slot, nb, winv] {
            for (int b = 1; b < nb; b++) ctx->h_res[evals_slot + 2 * b] = e2_mul(ctx->h_res[evals_slot + 2 * b], (*winv)[b]);
        });
    }
    // proof map (HG_PROOF_MAP=<file>): byte offset of every protocol element, for diffing against a proof dumped by the
    // Rust reference (scripts/proof_diff.py) - each label names the convention that decides those bytes
    std::vector<std::pair<size_t, std::string>> proof_map;
    void mark(const std::string& label) {
        if (hg_proof_map_path()) push_op([this, label] { proof_map.push_back({proof.bytes.size(), label}); });
    }
    void defer_write_slots(size_t s, size_t n) {
        push_op([this, s, n] { for (size_t i = 0; i < n; i++) proof.write_e(h_res()[s + i]); });
    }

    // ---- batched bookkeeping kernels (eq tables, zkCNN DFT rows, Libra gathers) ------------------------
    std::vector<dev::EqJob> eq_queue;
    std::vector<std::function<void()>> after_eq;  // device work that reads the queued eq tables
    std::vector<dev::GatherJob> gather_queue;
    std::vector<dev::GatherSegJob> gather_seg_queue;
    std::vector<dev::GatherBJob> gatherB_queue;
    std::vector<dev::FftJob> fft_queue;

    void eq_now(E2* out, int n, size_t point_off, const E2* point_dev = nullptr) {  // single table, launched in place
        dev::EqJob J;
        memset(&J, 0, sizeof(J));
        J.out = out; J.n = n; J.point_dev = point_dev;
        J.cs.n = 1; J.cs.unit_alpha = 1; J.cs.point_off[0] = point_off;
        const bool two = eq_two_launch() && n <= 24;
        dev::EqAbGrid grid{0, 0};
        if (two) { J.ab = ctx->alloc_n<E2>(dev::eq_ab_entries(n)); grid = dev::eq_ab_plan(&J, 1); }
        dev::EqJob* d = ctx->alloc_n<dev::EqJob>(1);
        upload(d, &J, sizeof(J), "upload eq job");
        ctx->prof_begin(cls_aux, 16.0 * ((size_t)1 << n));
        if (two) dev::eq_jobs_ab(st, d, 1, grid, ctx->d_chal);
        else dev::eq_jobs(st, d, 1, n, ctx->d_chal);
        ctx->prof_end();
    }
    std::vector<std::function<void()>> eq_post;  // sums of per-claim eq tables, run right after the eq batch
    // (tables of more than 2^24 entries take the one-launch kernel, whose workgroups rebuild their own low / high factor tables)
    static bool eq_two_launch() { return true; }
    void queue_eq(E2* out, int n, const dev::ClaimSet& cs) {
        dev::EqJob J;
        memset(&J, 0, sizeof(J));
        J.n = n;
        if (eq_two_launch() && n <= 24) {  // all claims in one job: the fill kernel sums them
            J.out = out; J.cs = cs;
            J.ab = ctx->alloc_n<E2>((size_t)cs.n * dev::eq_ab_entries(n));
            eq_queue.push_back(J);
            return;
        }
        if (cs.n == 1) { J.out = out; J.cs = cs; eq_queue.push_back(J); return; }
        // several claims: one job per claim (all of them run in parallel in the batch), then one summing pass
        const size_t N = (size_t)1 << n;
        E2* tmp = ctx->alloc_n<E2>((size_t)cs.n * N);
        for (int a = 0; a < cs.n; a++) {
            J.out = tmp + (size_t)a * N;
            memset(&J.cs, 0, sizeof(J.cs));
            J.cs.n = 1; J.cs.unit_alpha = cs.unit_alpha; J.cs.alpha_off = cs.alpha_off + a; J.cs.point_off[0] = cs.point_off[a];
            eq_queue.push_back(J);
        }
        const int nc = cs.n;
        eq_post.push_back([this, out, tmp, nc, N] { dev::sum_tables(st, out, tmp, nc, N); });
    }
    template <typename JobT, typename LaunchFn>
    void flush_jobs(std::vector<JobT>& q, int cls, double bytes, LaunchFn launch) {
        if (q.empty()) return;
        JobT* d = ctx->alloc_n<JobT>(q.size());
        upload(d, q.data(), q.size() * sizeof(JobT), "upload jobs");
        ctx->prof_begin(cls, bytes);
        launch(d, (int)q.size());
        ctx->prof_end();
        q.clear();
    }
    void flush_bookkeeping() {
        if (!eqpt_queue.empty()) {   // factor tables of the eq-factored jobs' points
            double pb = 0;
            for (auto& p : eqpt_queue) pb += 16.0 * (dev::ps_eq_lo_entries(p.nvars) + dev::ps_eq_suf_entries(p.nvars));
            flush_jobs(eqpt_queue, cls_aux, pb, [&](dev::PsEqPoint* d, int np) { dev::ps_eq_prep(st, d, np, ctx->d_chal); });
        }
        int max_n = 0; double eb = 0;
        for (auto& J : eq_queue) { max_n = std::max(max_n, J.n); eb += 16.0 * ((size_t)1 << J.n); }
        {   // jobs of the two-launch form first (queue_eq gives every job of a prove the same form)
            std::vector<dev::EqJob> ab, old;
            for (auto& J : eq_queue) (J.ab ? ab : old).push_back(J);
            if (!ab.empty() && !old.empty()) throw Error("eq tables: mixed job forms in one batch");
            if (!ab.empty()) {
                const dev::EqAbGrid grid = dev::eq_ab_plan(ab.data(), (int)ab.size());
                flush_jobs(ab, cls_aux, eb, [&](dev::EqJob* d, int nj) { dev::eq_jobs_ab(st, d, nj, grid, ctx->d_chal); });
                eq_queue.clear();
            }
        }
        flush_jobs(eq_queue, cls_aux, eb, [&](dev::EqJob* d, int nj) { dev::eq_jobs(st, d, nj, max_n, ctx->d_chal); });
        for (auto& f : eq_post) f();
        eq_post.clear();
        for (auto& f : after_eq) f();
        after_eq.clear();
        size_t max_total = 0; double gb = 0;
        for (auto& J : gather_queue) { size_t t = (size_t)1 << (J.log2_S + J.log2_R); max_total = std::max(max_total, t); gb += 24.0 * t; }
        flush_jobs(gather_queue, cls_gather, gb, [&](dev::GatherJob* d, int nj) { dev::gather_jobs(st, d, nj, max_total); });
        if (!gather_seg_queue.empty()) {
            double sb = 0;
            for (auto& J : gather_seg_queue) sb += 16.0 * (J.nseg + 1) * ((size_t)1 << (J.log2_S + J.log2_R));
            const int grid = dev::gather_seg_plan(gather_seg_queue.data(), (int)gather_seg_queue.size());
            flush_jobs(gather_seg_queue, cls_gather, sb, [&](dev::GatherSegJob* d, int nj) { dev::gather_seg_jobs(st, d, nj, grid); });
        }
        size_t maxB = 0; double bb = 0;
        for (auto& J : gatherB_queue) { size_t t = (size_t)1 << (J.log2_S + J.log2_R); maxB = std::max(maxB, t); bb += 40.0 * t; }
        flush_jobs(gatherB_queue, cls_gather, bb, [&](dev::GatherBJob* d, int nj) { dev::gather_B_jobs(st, d, nj, maxB); });
        int max_L = 0, max_claims = 1; double fb = 0;
        for (auto& J : fft_queue) { max_L = std::max(max_L, J.L); max_claims = std::max(max_claims, J.cs.n); fb += 24.0 * ((size_t)1 << J.L); }
        E2* fft_tab = fft_queue.empty() ? nullptr : ctx->alloc_n<E2>(fft_queue.size() * (size_t)max_claims * (((size_t)1 << max_L) >> 4) + 1);
        flush_jobs(fft_queue, cls_aux, fb, [&](dev::FftJob* d, int nj) { dev::fft_jobs(st, d, nj, max_L, max_claims, ctx->d_chal, fft_tab); });
    }

    // ---- Lasso node (lasso.rs:57-114) ------------------------------------------------------------
    struct GpOut { size_t point_off; std::shared_ptr<std::vector<E2>> claims; };  // claims: final per-table claims, valid after replay
    // prove_grand_product (prover.rs:183-266) over nb contiguous tables of `len` base-field values
    // `owner[n]` = rank that runs layer n (n = 0: roots + top evaluations); H may be null when no layer is owned
    int gp_deepest(int nv, const std::vector<int>& owner) const {  // highest tree level this rank needs (layer n reads level nv-1-n)
        int deepest = 0;
        for (int n = 0; n < nv; n++) if (mine(owner[n])) deepest = std::max(deepest, nv - 1 - n);
        return deepest;
    }
    // Joint classes of the read rows of the Lasso top layer (GpHashSrc::slot_of), built with the hash sources in lasso_node
    // (layer 0 = the top layer, rows = the read rows held; layer d >= 1: rows = the read and write rows held, groups of 2^(d+1) segments)
    struct SlotLayer {
        int V = 0, ng = 0, nrows = 0;
        std::vector<uint8_t> slot_of, rep;   // [row * ng + group], [slot * ng + group] (255: no such class there)
        uint8_t* d_slot_of = nullptr; uint8_t* d_rep = nullptr; E2* d_slotw = nullptr; E2* d_ratio = nullptr;
        u64* d_emit = nullptr;               // layer 0: V * ng read masks then V * ng write masks
    };
    struct GpSlots { int V = 0, NP = 0, G = 0 /* read rows held */, seg_shift = 0; std::vector<SlotLayer> layer; } gp_slots;
    // class weights W[v][g] = sum of the members' gamma^b (and W r_0 for the first round's weighted fold), and per row gamma^b / W of
    // its class: what turns a class's folded left table back into the row's (gp_slot_regroup)
    void slot_weights(const SlotLayer& sl, const dev::Powers& pw, E2 r0) {
        const int V = sl.V, ng = sl.ng, R = sl.nrows;
        std::vector<E2> W((size_t)V * ng, e2_zero());
        for (int b = 0; b < R; b++)
            for (int g = 0; g < ng; g++) { E2& w = W[(size_t)sl.slot_of[(size_t)b * ng + g] * ng + g]; w = e2_add(w, pw.v[b]); }
        // one inversion for all of them (a group with fewer classes than V keeps weight zero on the rest)
        std::vector<E2> pre(W.size()), inv(W.size(), e2_zero());
        E2 run_p = e2_one();
        for (size_t q = 0; q < W.size(); q++) { pre[q] = run_p; if (W[q].c0 | W[q].c1) run_p = e2_mul(run_p, W[q]); }
        E2 run_i = e2_inv(run_p);
        for (size_t q = W.size(); q-- > 0;) if (W[q].c0 | W[q].c1) { inv[q] = e2_mul(run_i, pre[q]); run_i = e2_mul(run_i, W[q]); }
        std::vector<E2> slotw(2 * W.size()), ratio((size_t)R * ng);
        for (size_t q = 0; q < W.size(); q++) { slotw[2 * q] = W[q]; slotw[2 * q + 1] = e2_mul(W[q], r0); }
        for (int b = 0; b < R; b++)
            for (int g = 0; g < ng; g++) {
                const size_t q = (size_t)sl.slot_of[(size_t)b * ng + g] * ng + g;
                if (!(W[q].c0 | W[q].c1)) throw Error("grand product: degenerate batching challenge");
                ratio[(size_t)b * ng + g] = e2_mul(pw.v[b], inv[q]);
            }
        upload(sl.d_slotw, slotw.data(), slotw.size() * sizeof(E2), "upload slot weights");
        upload(sl.d_ratio, ratio.data(), ratio.size() * sizeof(E2), "upload slot ratios");
    }
    // lev1 (optional): the first tree level, already produced by the hash kernel.
    // `local` (multi-GPU split by batch item): H holds only the rows of the global pairs listed in `local` (ascending);
    // with p0_only the first of them is pair 0, held only to supply p_0.
    // `hash_src` (device pointer): level 0 is not materialised, the top layer's first round recomputes it (k_gp_first_hash).
    // `emit` > 0: tree levels 1 .. emit are written by the first rounds of the top `emit` layers (their products ARE the next
    // level), which therefore run one after the other before everything else; the remaining small levels follow them.
    GpOut grand_product(const u64* H, size_t len, int nb, const std::vector<int>& owner, const u64* lev1 = nullptr,
                        const std::vector<int>* local = nullptr, bool p0_only = false, const dev::GpHashSrc* hash_src = nullptr, int emit = 0,
                        double hash_fused_bytes = 0, const u64* mirror_c = nullptr) {
        int nv = 0;
        while (((size_t)1 << nv) < len) nv++;
        const int nl = local ? (int)local->size() : nb;  // rows actually held
        std::vector<const u64*> lev(nv, nullptr);
        lev[0] = H;
        const int deepest = gp_deepest(nv, owner);
        if (emit > 0) {
            for (int n = 0; n < nv; n++) if (!mine(owner[n])) throw Error("grand product: level-emitting first rounds need every layer on this rank");
            if (emit > nv - 1 || lev1) throw Error("grand product: bad emit depth");
        } else if (hash_src) throw Error("grand product: recomputed level 0 needs emit >= 1");
        if (lev1 && nv > 1) lev[1] = lev1;
        // buffers of every level this rank needs (host-side bump allocation), then the launches that fill levels > emit
        std::vector<u64*> lev_w(nv, nullptr);
        for (int k = (lev1 && nv > 1) ? 2 : 1; k <= deepest; k++) { lev_w[k] = nl > 0 ? ctx->alloc_n<u64>((size_t)nl * (len >> k)) : nullptr; lev[k] = lev_w[k]; }
        size_t roots = slot(nb), ev0 = slot(2 * (size_t)nb);
        E2* lr = nullptr;
        E2* le = nullptr;
        const bool top_mine = mine(owner[0]) && nl > 0;
        if (top_mine && local) { lr = ctx->alloc_n<E2>(nl); le = ctx->alloc_n<E2>(2 * (size_t)nl); }
        const int first_k = std::max((lev1 && nv > 1) ? 2 : 1, emit + 1);
        auto build = [this, lev, lev_w, len, nl, nb, nv, deepest, first_k, roots, ev0, lr, le, top_mine, local]() {
            for (int k = first_k; k <= deepest; k++) {  // Layer::bottom / Layer::up: w = v_l * v_r on the MSB split
                const size_t in_len = len >> (k - 1);
                if (in_len <= (size_t)dev::PROD_TAIL_LEN && in_len >= 2) {  // the remaining (small) levels in one launch
                    dev::ProdTailOut outs;
                    memset(&outs, 0, sizeof(outs));
                    int nlev = 0;
                    double bytes = 0;
                    for (int kk = k; kk <= deepest; kk++) { outs.p[nlev++] = lev_w[kk]; bytes += (double)nl * (len >> (kk - 1)) * 8.0 * 1.5; }
                    ctx->prof_begin(cls_tree, bytes);
                    if (nl > 0) dev::prod_tail(st, lev[k - 1], (int)in_len, outs, nlev, nl);
                    ctx->prof_end();
                    break;
                }
                // three levels per launch while the third one is still above the tail's size
                if (k + 2 <= deepest && (in_len >> 2) > (size_t)dev::PROD_TAIL_LEN && in_len <= ((size_t)1 << 18)) {
                    ctx->prof_begin(cls_tree, (double)nl * in_len * 8.0 * 1.5 * 1.75);
                    if (nl > 0) dev::prod_level3(st, lev[k - 1], in_len, lev_w[k], lev_w[k + 1], lev_w[k + 2], nl);
                    ctx->prof_end();
                    k += 2;
                    continue;
                }
                ctx->prof_begin(cls_tree, (double)nl * in_len * 8.0 * 1.5);
                if (nl > 0) dev::prod_level(st, lev[k - 1], in_len, lev_w[k], nl);
                ctx->prof_end();
            }
            if (top_mine) {
                if (!local) dev::gp_top(st, lev[nv - 1], nb, d_res() + roots, d_res() + ev0);
                else dev::gp_top(st, lev[nv - 1], nl, lr, le);
            }
        };
        if (emit > 0) st_after_seq.push_back(build);
        else build();
        if (top_mine && local)
            for (int li = p0_only ? 1 : 0; li < nl; li++) {
                int b = (*local)[li];
                scatter.push_back({lr + li, roots + (size_t)b});
                scatter.push_back({le + 2 * li, ev0 + 2 * (size_t)b});
                scatter.push_back({le + 2 * li + 1, ev0 + 2 * (size_t)b + 1});
            }
        auto claims = std::make_shared<std::vector<E2>>(nb);
        mark("grand product: " + std::to_string(nb) + " root products (prover.rs:197-221)");
        push_op([this, roots, nb, claims] {  // root products (prover.rs:197-221)
            for (int b = 0; b < nb; b++) { (*claims)[b] = h_res()[roots + b]; proof.write_e((*claims)[b]); }
        });
        auto layer_down = [this, claims, nb](size_t evals_slot, E2 mu) {  // prover.rs:288-294
            push_op([this, claims, nb, evals_slot, mu] {
                const E2* ev = h_res() + evals_slot;
                for (int b = 0; b < nb; b++) (*claims)[b] = e2_add(ev[2 * b], e2_mul(mu, e2_sub(ev[2 * b + 1], ev[2 * b])));
            });
        };
        GpOut out{0, claims};
        // layer with num_vars 0
        mark("grand product layer 0: v_l, v_r evaluations per tree (prover.rs:257; no sum-check)");
        defer_write_slots(ev0, 2 * (size_t)nb);
        out.point_off = epos();
        layer_down(ev0, squeeze());
        for (int n = 1; n < nv; n++) {
            int k = nv - 1 - n;
            size_t h = (size_t)1 << n;
            E2 gamma = squeeze();  // prover.rs:238
            dev::Powers pw;
            memset(&pw, 0, sizeof(pw));
            if (nb > dev::PW_MAX) throw Error("grand product: too many batched tables");
            E2 g = e2_one();
            for (int b = 0; b < nb; b++) { pw.v[b] = g; g = e2_mul(g, gamma); }
            Cell claim = cell();
            push_op([claims, nb, pw, claim] {  // sum_check_claim (prover.rs:281-286)
                E2 c = e2_zero();
                for (int b = 0; b < nb; b++) c = e2_add(c, e2_mul((*claims)[b], pw.v[b]));
                *claim = c;
            });
            size_t evals = slot(2 * (size_t)nb);
            ScHandle sc;
            const int seq = n >= nv - emit ? nv - n : 0;                    // first round launched alone, deepest layer first
            u64* nxt = seq ? lev_w[k + 1] : nullptr;                          // ... and writes tree level k + 1
            const dev::GpHashSrc* hs = (hash_src && k == 0) ? hash_src : nullptr;
            if (hs) { pending_fused_bytes = hash_fused_bytes; pending_fused_model_extra = hash_model_extra; pending_hash_reads = hash_design_reads; }
            if (nxt) {   // rows of level k + 1 this first round writes: the next layer's slot rows in slot form, one per row held otherwise
                const bool next_slots = hash_src && !local && k + 1 < (int)gp_slots.layer.size();
                pending_level_design = (double)(next_slots ? gp_slots.layer[k + 1].V : nl) * (double)(len >> (k + 1)) * 8.0;
            }
            const bool mirrored = hs && mirror_c;
            if (mirrored) {
                // Top layer with row b + nb/2 = row b + c for every b < nb/2 (the Lasso write hashes, c = gamma^2): only the read rows
                // are stored and multiplied (StJob::mirror in kernels.hpp). The kernels sum everything but the factor 1 + kappa.
                const int G2 = nb / 2;
                std::vector<int> rows;   // global ids of the read rows held, ascending (row 0 first)
                for (int li = 0; li < nl; li++) { const int b = local ? (*local)[li] : li; if (b < G2) rows.push_back(b); }
                const int R = (int)rows.size();
                for (int li = 0; li < R; li++) if ((local ? (*local)[li] : li) != rows[li]) throw Error("grand product: read rows must come first");
                int nwr = 0;
                for (int li = R; li < nl; li++) {
                    const int b = (local ? (*local)[li] : li) - G2;
                    if (std::find(rows.begin(), rows.end(), b) == rows.end() || (p0_only && b == 0)) throw Error("grand product: write row without its read row");
                    nwr++;
                }
                if (nwr != R - (p0_only ? 1 : 0)) throw Error("grand product: mirrored rows do not pair up");
                if (R > 63 || 2 * R + 1 > dev::PW_MAX) throw Error("grand product: too many mirrored rows");
                const E2 kappa = pw.v[G2], onek = e2_add(e2_one(), kappa);
                if (onek.c0 == 0 && onek.c1 == 0) throw Error("grand product: degenerate batching challenge");
                const E2 kp = e2_mul(kappa, e2_inv(onek));
                dev::Powers pwl;
                memset(&pwl, 0, sizeof(pwl));
                E2 lam = e2_zero();
                for (int li = 0; li < R; li++) { pwl.v[li] = pw.v[rows[li]]; if (!(p0_only && li == 0)) lam = e2_add(lam, pwl.v[li]); }
                const u64 c = *mirror_c;
                MirrorSpec ms;
                ms.k1 = e2_mul_f(kp, c);
                ms.k2 = e2_mul(e2_mul_f(kp, gl_mul(c, c)), lam);
                ms.credit_ntab = 2 * nl;
                E2* fin = ctx->alloc_n<E2>(2 * (size_t)R + 1);
                const bool run = mine(owner[n]) && R > (p0_only ? 1 : 0);
                const bool slotted = !gp_slots.layer.empty();   // (decided with the hash sources, lasso_node)
                if (slotted && (!run || R != gp_slots.G)) throw Error("grand product: the slot plan does not fit the rows held");
                SlotPlan spl;
                if (slotted) {
                    const SlotLayer& sl = gp_slots.layer[0];
                    spl.tail_ntab = 2 * R + 1; spl.d_slot_of = sl.d_slot_of; spl.d_ratio = sl.d_ratio;
                    spl.nrows = R; spl.nslots = sl.V; spl.npairs = sl.ng; spl.max_rd = gp_slots.seg_shift;
                }
                sc = sc_stride(dev::SC_GRANDPROD, lev[k], true, h, slotted ? 2 * gp_slots.layer[0].V + 1 : 2 * R + 1, n, pwl, fin, run, p0_only, seq, nxt, hs, &ms, 0,
                               slotted ? &spl : nullptr);
                sc.scaled = true; sc.scale = onek;
                if (slotted) slot_weights(gp_slots.layer[0], pwl, sc.rs[0]);
                if (run)
                    for (int li = p0_only ? 1 : 0; li < R; li++) {
                        scatter.push_back({fin + 2 * li, evals + 2 * (size_t)rows[li]});
                        scatter.push_back({fin + 2 * li + 1, evals + 2 * (size_t)rows[li] + 1});
                    }
            } else if (!local && hash_src && k >= 1 && k < (int)gp_slots.layer.size()) {
                // slot form below the top layer: the input rows are this layer's slot rows (written by the layer above), the tail runs
                // on the 2 nb per-row tables again
                const SlotLayer& sl = gp_slots.layer[k];
                if (sl.nrows != nb || !mine(owner[n]) || !seq) throw Error("grand product: slot form on a partial batch");
                SlotPlan spl;
                spl.tail_ntab = 2 * nb; spl.d_slot_of = sl.d_slot_of; spl.d_ratio = sl.d_ratio;
                spl.nrows = nb; spl.nslots = sl.V; spl.npairs = sl.ng; spl.max_rd = gp_slots.seg_shift;
                spl.job_slotw = sl.d_slotw; spl.job_emit = sl.d_emit;
                sc = sc_stride(dev::SC_GRANDPROD, lev[k], true, h, 2 * sl.V, n, pw, d_res() + evals, true, false, seq, nxt, nullptr, nullptr, 2 * nb, &spl);
                slot_weights(sl, pw, sc.rs[0]);
            } else if (!local) sc = sc_stride(dev::SC_GRANDPROD, lev[k], true, h, 2 * nb, n, pw, d_res() + evals, mine(owner[n]), false, seq, nxt, hs);
            else {
                // this rank's share of the batch: local pair li is global pair b = local[li], weight gamma^b
                dev::Powers pwl;
                memset(&pwl, 0, sizeof(pwl));
                for (int li = 0; li < nl; li++) pwl.v[li] = pw.v[(*local)[li]];
                E2* fin = nl ? ctx->alloc_n<E2>(2 * (size_t)nl) : nullptr;
                if (hash_src && k >= 1 && k < (int)gp_slots.layer.size()) {   // slot form below the top layer, on the rows this rank holds
                    const SlotLayer& sl = gp_slots.layer[k];
                    if (sl.nrows != nl || !mine(owner[n]) || !seq || nl <= (p0_only ? 1 : 0)) throw Error("grand product: the slot plan does not fit the rows held");
                    SlotPlan spl;
                    spl.tail_ntab = 2 * nl; spl.d_slot_of = sl.d_slot_of; spl.d_ratio = sl.d_ratio;
                    spl.nrows = nl; spl.nslots = sl.V; spl.npairs = sl.ng; spl.max_rd = gp_slots.seg_shift;
                    spl.job_slotw = sl.d_slotw; spl.job_emit = sl.d_emit;
                    sc = sc_stride(dev::SC_GRANDPROD, lev[k], true, h, 2 * sl.V, n, pwl, fin, true, p0_only, seq, nxt, nullptr, nullptr, 2 * nl, &spl);
                    slot_weights(sl, pwl, sc.rs[0]);
                } else
                sc = sc_stride(dev::SC_GRANDPROD, lev[k], true, h, 2 * nl, n, pwl, fin, mine(owner[n]) && nl > (p0_only ? 1 : 0), p0_only, seq, nxt, hs);
                if (mine(owner[n]))
                    for (int li = p0_only ? 1 : 0; li < nl; li++) {
                        int b = (*local)[li];
                        scatter.push_back({fin + 2 * li, evals + 2 * (size_t)b});
                        scatter.push_back({fin + 2 * li + 1, evals + 2 * (size_t)b + 1});
                    }
            }
            mark("grand product layer " + std::to_string(n) + ": sum-check, " + std::to_string(n) + " rounds x 4 coefficients [C1 message format, C2 power order, C3 variable order]");
            defer_sumcheck(sc, 3, claim, nullptr);
            defer_gp_unscale(evals, nb, pw);
            if (mirrored) {   // the write rows' evaluations: folding is affine with coefficients summing to one, so row + c stays row + c
                const u64 c = *mirror_c;
                const int G2 = nb / 2;
                push_op([this, evals, G2, c] {
                    for (int b = 0; b < G2; b++)
                        for (int t = 0; t < 2; t++) ctx->h_res[evals + 2 * (size_t)(G2 + b) + t] = e2_add_f(ctx->h_res[evals + 2 * (size_t)b + t], c);
                });
            }
            mark("grand product layer " + std::to_string(n) + ": v_l, v_r evaluations per tree (prover.rs:257)");
            defer_write_slots(evals, 2 * (size_t)nb);  // prover.rs:257
            out.point_off = sc.point_off;
            layer_down(evals, squeeze());              // mu (prover.rs:259)
        }
        return out;
    }

    ClaimRef lasso_node(const u64* d_input) {
        // where the node enters the transcript: `ch.pos / 2` extension-field challenges have been squeezed before it (the argument
        // hg_lasso_prove_at takes to reproduce this section on its own)
        mark("lasso node: enters after " + std::to_string(ch.pos / 2) + " squeezed challenges (lasso.rs:57-114)");
        const LassoPlan& lp = pk->lasso;
        const dev::LassoDev& L = pk->lasso_dev;
        const int nu = lp.nu, A = lp.alpha;
        const size_t N = (size_t)1 << nu, M = 65536;
        // What this rank does of the node (single GPU: everything). Sharded over `world` GPUs the whole node is split BY MEMORY:
        // rank r owns the memories gkr_order[i] with gp1_mem_owner[i] == r and runs, for those, their E tables, their share of the
        // claimed sum, of the collation sum-check, of both grand products and of the openings. Every batched sum-check here is
        // linear in its batch items once p_0 is fixed, so every rank also keeps the one table that supplies p_0 (folded, never
        // summed: StJob::p0_only) and the ranks' round sums are partial sums that the exchange adds up.
        const int G = (int)lp.gkr_order.size();
        const bool split = world > 1;
        std::vector<int> local_pairs;  // global pair ids (reads / inits: i, writes / finals: G + i) this rank holds, ascending
        std::vector<int> local_mems;   // memory-GKR indices i it owns
        bool p0_only = false;          // pair 0 of the grand products is held only for p_0
        std::vector<char> own_mem(A, split ? 0 : 1);   // by memory index m
        if (split) {
            for (int i = 0; i < G; i++) if (mine(gp1_mem_owner[i])) { local_mems.push_back(i); own_mem[lp.gkr_order[i]] = 1; }
            p0_only = !mine(gp1_mem_owner[0]);
            if (p0_only) local_pairs.push_back(0);
            for (int i : local_mems) local_pairs.push_back(i);
            for (int i : local_mems) local_pairs.push_back(G + i);
        }
        bool any_gp1 = false;
        for (int n = 0; n < nu; n++) any_gp1 |= mine(gp1_owner[n]);
        const bool any_local = !split || !local_mems.empty();
        const bool do_col = any_local, do_open = any_local, do_gp2 = any_local;
        const bool need_counters = any_gp1 || do_gp2 || do_open;
        const bool need_split = do_col || need_counters;
        // E tables this rank materialises: its own memories, memory 0 (p_0 of the collation sum-check) and memory gkr_order[0] (pair
        // 0 of grand product #1). Row 0 is always memory 0 and the owned memories follow in ascending order, so the collation
        // sum-check's tables are rows [0, ncol) of `ep`.
        dev::EpRows ep_rows = dev::ep_rows_all(A), ep_rows_own = dev::ep_rows_all(A);
        std::vector<int> col_mems;     // memory indices of the collation tables held, table 0 first
        int ep_count = A;
        if (split) {
            for (int m = 0; m < 32; m++) ep_rows.row[m] = ep_rows_own.row[m] = -1;
            col_mems.push_back(0);
            for (int m = 1; m < A; m++) if (own_mem[m]) col_mems.push_back(m);
            ep_count = 0;
            for (int m : col_mems) ep_rows.row[m] = (signed char)ep_count++;
            if (ep_rows.row[lp.gkr_order[0]] < 0) ep_rows.row[lp.gkr_order[0]] = (signed char)ep_count++;
            for (int m = 0; m < A; m++) if (own_mem[m]) ep_rows_own.row[m] = ep_rows.row[m];
        } else for (int m = 0; m < A; m++) col_mems.push_back(m);
        const bool col_p0_only = split && !own_mem[0];
        // Lean form (wherever the hash-free first round is used): the E tables are NOT materialised.
        // E_m[j] = (row j's lookup uses m and limb < cutoff_m) ? limb : 0 is a select on a limb, so the hash round, the claimed sum and
        // the E_m(x) openings recompute it and the limb split writes E_0 (the collation sum-check's p_0 table) and C only.
        bool lean_e = false;
        {
            bool all1 = any_gp1;
            for (int n = 0; n < nu; n++) all1 = all1 && mine(gp1_owner[n]);
            const int nrows_ = split ? (int)local_pairs.size() : 2 * G;
            lean_e = all1 && nu >= 12 && nrows_ > (p0_only ? 1 : 0) && A <= 32;   // (= the condition of emit > 0 below)
        }
        const int ep_count_full = ep_count;   // (what the reference's traffic model writes)
        if (lean_e) {
            for (int m = 0; m < 32; m++) ep_rows.row[m] = ep_rows_own.row[m] = -1;
            ep_rows.row[0] = 0;
            ep_count = 1;
        }
        u32 own_mask = 0;
        for (int m = 0; m < A && m < 32; m++) if (own_mem[m]) own_mask |= 1u << m;
        // polynomialize (lasso.rs:157-250)
        u64* dims = nullptr;
        u64* ep = nullptr;
        auto epm = [&](int m) -> const u64* {
            if (ep_rows.row[m] < 0) throw Error("lasso: E table of a memory this rank does not hold");
            return ep + (size_t)ep_rows.row[m] * N;
        };
        if (need_split) {
            dims = ctx->alloc_n<u64>(4 * N);
            // one more row behind the E tables: C = sum_m M^m E_m over this rank's memories - with E_0 all the collation sum-check needs
            ep = ctx->alloc_n<u64>((size_t)(ep_count + 1) * N);
            dev::ColPow cp;
            memset(&cp, 0, sizeof(cp));
            {
                u64 mp = 1;
                for (int m = 0; m < A; m++) { if (own_mem[m]) cp.v[m] = mp; mp = gl_mul(mp, M); }
            }
            // two streams: the limbs first, in their own small launch - the counter sorts (second stream) need nothing else and start
            // while the E tables are still being written
            if (fork_recorded) {
                ctx->prof_begin(cls_aux, (double)N * 8 * (1 + 4));
                dev::lasso_dims(st, L, d_input, dims);
                ctx->prof_end();
                hip_check(hipEventRecord(ctx->ev_aux[0], st), "lasso: limbs event");
            }
            ctx->prof_begin(cls_aux, (double)N * 8 * (1 + (fork_recorded ? 0 : 4) + ep_count + 1), (double)N * 8 * (1 + (fork_recorded ? 0 : 4) + ep_count_full + 1));
            dev::lasso_split(st, L, d_input, fork_recorded ? nullptr : dims, ep, ep_rows, &cp, ep + (size_t)ep_count * N);
            stamp("limb split done");
            ctx->prof_end();
            if (fork_recorded) hip_check(hipEventRecord(ctx->ev_aux[2], st), "lasso: E tables event");
        }
        // MemoryCheckingProver::new (prover.rs:35-89)
        const int nrows = split ? (int)local_pairs.size() : 2 * G;
        // Grand product #1 without hash tables: the top layer's first round recomputes the hashes from dims / read_ts / E and
        // writes tree level 1, the next layers' first rounds write levels 2 .. emit (a fifth level-emitting layer: no gain). Small
        // tables and sharded ranks that do not run every layer take the classic path: hash rows and tree levels materialised.
        constexpr int emit_max = 4;
        bool all_gp1 = any_gp1;
        for (int n = 0; n < nu; n++) all_gp1 = all_gp1 && mine(gp1_owner[n]);
        int emit = 0;
        if (all_gp1 && nu >= 12 && nrows > (p0_only ? 1 : 0))
            for (int n = nu - 1; n >= 12 && emit < emit_max; n--) emit++;   // layers with 2^n >= 4096 entries per table
        // Off the critical path, on the second stream: counter sorts (hidden under the collation sum-check), grand product #2's
        // hashes and tree, the openings (hidden under grand product #1's rounds). Needs the hash-free grand product #1 (the
        // classic path reads read_ts on the main stream right away).
        const bool use_aux = fork_recorded && emit > 0;
        auto aux = [&](const std::function<void()>& fn) { if (use_aux) on_aux(fn); else fn(); };
        // r, claimed sum (lasso.rs:85, 264-269)
        size_t r_off = epos();
        for (int i = 0; i < nu; i++) squeeze();
        E2* eq = (do_col || do_open) ? ctx->alloc_n<E2>(N) : nullptr;
        size_t claim_slot = slot(1);
        // Stream assignment inside the node (one rank): the main stream goes from the limb split to the first hash round; counters,
        // grand product #2's tree and the opening tables on the second stream; the collation rounds on the third (col_third below).
        // A sharded rank keeps the collation rounds on the main stream ahead of its grand products. (Measured and removed in round 5:
        // the collation sum-check and the claimed sum on the second stream with the counters leading the main one, 3.8-3.9 against
        // 3.55 ms; the collation rounds behind the counters on the second stream, 2.74-2.82 against 2.64 ms; the counters on the
        // main stream, 3.87 against 3.63 ms.)
        // The claimed sum is only a result slot: with two streams it runs on the second one after grand product #2's tree (the main
        // stream goes from the limb split straight into the collation rounds, the second stream is idle at that point anyway).
        const bool claim_late = use_aux;
        auto do_claim = [&] {
            eq_now(eq, nu, r_off);
            int grid = lean_e ? dev::lasso_claim_in(st, L, eq, d_input, own_mask, partials)
                              : dev::lasso_claim(st, L, eq, ep, ep_rows_own, partials);  // sharded: this rank's memories only (partial sum)
            reduce(grid, 1, claim_slot);
        };
        if (do_col && !claim_late) do_claim();
        Cell claimed = cell();
        mark("lasso: claimed sum (lasso.rs:100-107)");
        push_op([this, claim_slot, claimed] { *claimed = h_res()[claim_slot]; proof.write_e(*claimed); });
        {   // collation sum-check (lasso.rs:271-279): g = poly(0) * sum_i M^i poly(i). Only the SUM enters the round polynomials and the
            // final evaluations are dropped (lasso.rs:97), so the sum-check runs on two tables: E_0 (supplies p_0, not summed) and
            // C = sum_i M^i E_i, written by the limb split (folding is linear: fold(C) = sum_i M^i fold(E_i)). Sharded: every rank's C
            // holds its own memories' terms, the round sums are partial sums.
            dev::Powers pw;
            memset(&pw, 0, sizeof(pw));
            if (A > dev::PW_MAX) throw Error("lasso: too many memories");
            pw.v[0] = e2_one(); pw.v[1] = e2_one();
            const bool col_run = do_col && (int)col_mems.size() > (col_p0_only ? 1 : 0);
            ScHandle sc = sc_stride(dev::SC_COLLATION, ep, true, (size_t)ep_count * N, 2, nu, pw, nullptr, col_run, true, 0, nullptr, nullptr, nullptr,
                                    (int)col_mems.size());
            mark("lasso: collation sum-check, " + std::to_string(nu) + " rounds x 3 coefficients (lasso.rs:271-279) [C1, C3; poly(0) quirk]");
            defer_sumcheck(sc, 2, claimed, nullptr);
        }
        E2 gamma_e = squeeze(), tau_e = squeeze();  // lasso.rs:99
        u64 gamma = gamma_e.c0, tau = tau_e.c0;     // prover.rs:38-39: base limb 0 only
        // counters: only the memories whose index equals a chunk (dimension) index reach the transcript
        // (lasso.rs:317-319 indexes read_ts/final_cts by chunk index)
        // a rank that only holds a few memories of grand product #1 needs the counters of their chunks only
        // (the chunk of pair 0, and the chunks whose dim / read_ts / final_cts openings it owns: those of its own memories)
        std::vector<char> need_chunk(4, split ? 0 : 1);
        if (split) {
            if (p0_only) need_chunk[lp.gkr_chunk[0]] = 1;
            for (int i : local_mems) need_chunk[lp.gkr_chunk[i]] = 1;
        }
        std::map<int, u64*> read_ts, final_cts;
        // The counters feed grand product #1's first launch; on the second stream they hide under the collation rounds.
        // The collation rounds - 0.2 ms of short launches whose results only the host reads - run on a THIRD stream forked from the main
        // one behind the E tables and joined to it at the end of the prove (one rank): the main stream goes from the limb split
        // straight to the first hash round (which waits for the counters only). 2.574 -> 2.524 ms (96 replays each).
        // (Forked from and joined to the ORIGIN stream of the capture, like the second stream: a stream forked from the second one
        // and joined back into it sent hipStreamEndCapture into an endless recursion on a sharded rank's graph, NOTEBOOK.md.)
        const bool col_third = use_aux && world == 1 && fork_recorded;
        if (col_third) on_col([&] { flush_stride(); });
        else if (use_aux) flush_stride();  // collation rounds first: see below
        bool counters_event_recorded = false;
        if (need_counters) aux([&] {
            if (use_aux) hip_check(hipStreamWaitEvent(st, ctx->ev_aux[0], 0), "lasso: wait for the limb split");
            // all requested chunks in ONE stable sort of (chunk, address) keys
            unsigned mask = 0;
            for (auto& chk : lp.chunks) {
                int c = chk.first;
                if (c < 0 || c >= 4 || !need_chunk[c]) continue;
                mask |= 1u << c;
                read_ts[c] = ctx->alloc_n<u64>(N);
                final_cts[c] = ctx->alloc_n<u64>(M);
            }
            const size_t elems = std::max<size_t>(dev::lasso_counters_all_elems(L, mask), 1);
            size_t tb = dev::lasso_counters_all_temp_bytes(elems);
            void* temp = ctx->alloc(tb);
            u32* keys = ctx->alloc_n<u32>(elems); u32* keys2 = ctx->alloc_n<u32>(elems);
            u32* rows = ctx->alloc_n<u32>(elems); u32* rows2 = ctx->alloc_n<u32>(elems);
            u32* starts = ctx->alloc_n<u32>(4 * 65536 + 1);
            ctx->prof_begin(cls_aux, (double)elems * 40);
            dev::CounterOut co;
            memset(&co, 0, sizeof(co));
            for (int c = 0; c < 4; c++) if ((mask >> c) & 1) { co.read_ts[c] = read_ts[c]; co.final_cts[c] = final_cts[c]; }
            dev::lasso_counters_all(st, L, mask, dims, co, temp, tb, keys, keys2, rows, rows2, starts);
            ctx->prof_end();
            stamp("counters done");
            if (use_aux) { hip_check(hipEventRecord(ctx->ev_aux[3], st), "lasso: counters event"); counters_event_recorded = true; counters_event_live = true; }
        });
        // the collation rounds are launched now, not with the grand products at the end of the node: the host still has the whole
        // memory-checking bookkeeping to walk (about 0.5 ms) and the main stream would sit idle meanwhile; behind the counters, so
        // that the second stream (grand product #2's tree, openings) can start while they run
        flush_stride();
        const dev::GpHashSrc* d_hash_src = nullptr;
        double hash_build_bytes = 0;
        if (emit > 0) {
            std::vector<dev::GpHashMem> hm;
            auto row_of = [&](int pair) -> int {
                if (!split) return pair;
                for (size_t q = 0; q < local_pairs.size(); q++) if (local_pairs[q] == pair) return (int)q;
                return -1;
            };
            for (int i = 0; i < G; i++) {  // memory-GKR order is chunk-major
                dev::GpHashMem m;
                m.chunk = lp.gkr_chunk[i]; m.rd_row = row_of(i); m.wr_row = row_of(G + i);
                m.mem = lp.gkr_order[i]; m.cutoff = L.mem_cutoff[m.mem];
                if (m.rd_row >= 0 || m.wr_row >= 0) { m.ep = lean_e ? nullptr : epm(lp.gkr_order[i]); hm.push_back(m); }
            }
            dev::GpHashSrc hs;
            memset(&hs, 0, sizeof(hs));
            for (int c = 0; c < 4; c++) { hs.dim[c] = dims + (size_t)c * N; hs.ts[c] = read_ts.count(c) ? read_ts[c] : nullptr; }
            for (auto& m : hm) if (!hs.ts[m.chunk]) throw Error("lasso: counters of a needed chunk were not computed");
            dev::GpHashMem* d_hm = ctx->alloc_n<dev::GpHashMem>(hm.size());
            upload(d_hm, hm.data(), hm.size() * sizeof(dev::GpHashMem), "upload hash sources");
            hs.mems = d_hm; hs.nmem = (int)hm.size(); hs.gamma = gamma; hs.gamma2 = gl_mul(gamma, gamma); hs.tau = tau;
            hs.seg_shift = L.seg_shift; hs.rows = L.rows; hs.seg_lookup = L.seg_lookup;
            memcpy(hs.lookup_uses, L.lookup_uses, sizeof(hs.lookup_uses));
            hash_recomp = lean_e;
            // Slot form (kernels.hpp, GpHashSrc::slot_of): inside a lookup's row segment the memories it does not use have, per chunk,
            // identical hash rows, and the top layer multiplies segment s with segment s + npairs (Layer::bottom splits a row into
            // halves): memories in the same class in both segments share one table pair until the tables are down to the segment pairs.
            // HG_SLOT_DEPTH = number of slot-form layers (default 4; 0 = the memory form throughout, the path small tables take anyway)
            static const int depth_max = [] { const char* e = getenv("HG_SLOT_DEPTH"); return e && *e ? atoi(e) : 4; }();
            gp_slots = GpSlots();
            const int nvars_top = nu - 1;
            // rows of the grand product as this rank holds them (all of them on one GPU; its own memories' on a sharded rank): reads first
            std::vector<int> rowmem;    // memory-GKR index of row t
            std::vector<char> roww;     // write row?
            if (!split) { for (int t = 0; t < 2 * G; t++) { rowmem.push_back(t % G); roww.push_back(t >= G); } }
            else for (int b : local_pairs) { rowmem.push_back(b % G); roww.push_back(b >= G); }
            const int NL = (int)rowmem.size();
            int R = 0;
            while (R < NL && !roww[R]) R++;
            std::vector<int> read_row_of(G, -1);
            for (int t = 0; t < R; t++) read_row_of[rowmem[t]] = t;
            bool rows_fit = (int)hm.size() == R && R >= 2 && NL <= 64;
            for (int t = 0; t < R && rows_fit; t++) rows_fit = hm[t].rd_row == t;           // (the hash kernel walks hm by read row)
            for (int t = R; t < NL && rows_fit; t++) rows_fit = roww[t] && read_row_of[rowmem[t]] >= 0;
            if (rows_fit && depth_max > 0 && G <= 32 && L.seg_shift >= 9 && nvars_top - 1 > L.seg_shift &&
                ((N / 2) >> L.seg_shift) <= 64 && nvars_top - 1 - slot_tail_h(2 * R + 1, nvars_top) <= L.seg_shift) {
                const int NP = (int)((N / 2) >> L.seg_shift);
                auto cls = [&](int i, int s) -> int {   // class of GKR position i in row segment s: itself where its memory is looked up, else its chunk
                    if (((size_t)s << L.seg_shift) < L.rows && ((L.lookup_uses[lp.seg_lookup[s]] >> lp.gkr_order[i]) & 1)) return 1000 + i;
                    return lp.gkr_chunk[i];
                };
                if (hg_debug("slots"))   // joint classes of deeper layers: layer d multiplies 2^(d+1) segments NP >> d apart
                    for (int d = 0; d < 4 && (NP >> d) >= 1; d++) {
                        const int np = NP >> d, cnt = 2 << d;
                        int vmax = 0;
                        for (int sp = 0; sp < np; sp++) {
                            std::vector<std::vector<int>> keys;
                            for (int i = 0; i < G; i++) {
                                std::vector<int> key;
                                for (int q = 0; q < cnt; q++) key.push_back(cls(i, sp + q * np));
                                if (std::find(keys.begin(), keys.end(), key) == keys.end()) keys.push_back(key);
                            }
                            vmax = std::max(vmax, (int)keys.size());
                        }
                        fprintf(stderr, "[hg slots] layer %d: %d segment groups of %d, at most %d classes of %d memories\n", d, np, cnt, vmax, G);
                    }
                GpSlots& gs = gp_slots;
                gs.NP = NP; gs.G = R; gs.seg_shift = L.seg_shift;
                // layer d multiplies 2^(d+1) segments NP >> d apart: its classes are over those; rows: reads (layer 0), reads then writes
                for (int d = 0; d < std::min(depth_max, emit); d++) {
                    SlotLayer sl;
                    sl.ng = NP >> d; sl.nrows = d == 0 ? R : NL;
                    const int nvars_d = nu - 1 - d;
                    if (sl.ng < 2 || nvars_d - 1 <= L.seg_shift || nvars_d - 1 - slot_tail_h(d == 0 ? 2 * R + 1 : 2 * NL, nvars_d) > L.seg_shift) break;
                    sl.slot_of.assign((size_t)sl.nrows * sl.ng, 0);
                    std::vector<std::vector<int>> reps(sl.ng);
                    for (int g = 0; g < sl.ng; g++) {
                        std::vector<std::vector<int>> keys;
                        for (int b = 0; b < sl.nrows; b++) {
                            std::vector<int> key;
                            if (b == 0) key.push_back(-1);   // row 0 alone: p_0
                            else {
                                key.push_back(roww[b] ? 1 : 0);
                                for (int q = 0; q < (2 << d); q++) key.push_back(cls(rowmem[b], g + q * sl.ng));
                            }
                            int v = -1;
                            for (size_t q = 0; q < keys.size(); q++) if (keys[q] == key) v = (int)q;
                            if (v < 0) { v = (int)keys.size(); keys.push_back(key); reps[g].push_back(b); }
                            sl.slot_of[(size_t)b * sl.ng + g] = (uint8_t)v;
                        }
                        sl.V = std::max(sl.V, (int)keys.size());
                    }
                    if (sl.V >= sl.nrows || sl.V > 64) break;   // (nothing to gain)
                    sl.rep.assign((size_t)sl.V * sl.ng, 255);
                    for (int g = 0; g < sl.ng; g++) for (size_t v = 0; v < reps[g].size(); v++) sl.rep[v * sl.ng + g] = (uint8_t)reps[g][v];
                    gs.layer.push_back(sl);
                }
                // where each layer's first round writes the next tree level: the next layer's slot rows, or the per-memory rows
                for (size_t d = 0; d < gs.layer.size(); d++) {
                    SlotLayer& sl = gs.layer[d];
                    const SlotLayer* nx = d + 1 < gs.layer.size() ? &gs.layer[d + 1] : nullptr;
                    const int T = nx ? nx->V : NL, ngn = sl.ng / 2;
                    std::vector<u64> em((size_t)sl.V * sl.ng * (d == 0 ? 2 : 1), 0);
                    for (int g = 0; g < sl.ng; g++)
                        for (int t = 0; t < T; t++) {
                            const int b = nx ? nx->rep[(size_t)t * ngn + (g % ngn)] : t;   // the row whose values target row t holds there
                            if (b == 255) continue;
                            if (d == 0) {   // (the top layer holds the read rows only: a write row's values come from its read row's class)
                                const int u = sl.slot_of[(size_t)(roww[b] ? read_row_of[rowmem[b]] : b) * sl.ng + g];
                                em[(roww[b] ? (size_t)sl.V * sl.ng : 0) + (size_t)u * sl.ng + g] |= (u64)1 << t;
                            } else em[(size_t)sl.slot_of[(size_t)b * sl.ng + g] * sl.ng + g] |= (u64)1 << t;
                        }
                    sl.d_slot_of = ctx->alloc_n<uint8_t>(sl.slot_of.size());
                    sl.d_rep = ctx->alloc_n<uint8_t>(sl.rep.size());
                    sl.d_slotw = ctx->alloc_n<E2>(2 * (size_t)sl.V * sl.ng);
                    sl.d_ratio = ctx->alloc_n<E2>((size_t)sl.nrows * sl.ng);
                    sl.d_emit = ctx->alloc_n<u64>(em.size());
                    upload(sl.d_slot_of, sl.slot_of.data(), sl.slot_of.size(), "upload slot map");
                    upload(sl.d_rep, sl.rep.data(), sl.rep.size(), "upload slot representatives");
                    upload(sl.d_emit, em.data(), em.size() * sizeof(u64), "upload slot emission masks");
                }
                if (hg_debug("slots")) {
                    std::string vs;
                    for (auto& sl : gs.layer) vs += " " + std::to_string(sl.V) + "/" + std::to_string(sl.nrows);
                    fprintf(stderr, "[hg slots] adopted: %d layers, classes / rows per layer:%s\n", (int)gs.layer.size(), vs.c_str());
                }
                if (!gs.layer.empty()) {
                    const SlotLayer& s0 = gs.layer[0];
                    gs.V = s0.V;
                    hs.slot_of = s0.d_slot_of; hs.rep = s0.d_rep; hs.slotw = s0.d_slotw; hs.npairs = NP; hs.nslots = s0.V;
                    hs.emit_rd = s0.d_emit; hs.emit_wr = s0.d_emit + (size_t)s0.V * s0.ng;
                }
            }
            dev::GpHashSrc* d_hs = ctx->alloc_n<dev::GpHashSrc>(1);
            upload(d_hs, &hs, sizeof(hs), "upload hash sources");
            d_hash_src = d_hs;
            // algorithmic bytes of the hash build this replaces (the accounting of lasso_hash_rw below): dim + read_ts per chunk in
            // use, E read and read / write hash rows written per memory; level 1 is credited by sc_stride (next_level)
            std::vector<char> chunk_used(4, 0);
            for (auto& m : hm) chunk_used[m.chunk] = 1;
            hash_build_bytes = 0;
            for (int c = 0; c < 4; c++) if (chunk_used[c]) hash_build_bytes += (double)N * 8 * 2;
            for (auto& m : hm) hash_build_bytes += (double)N * 8 * ((m.rd_row >= 0) + (m.wr_row >= 0) + (lean_e ? 0 : 1));
            hash_model_extra = lean_e ? (double)N * 8 * (double)hm.size() : 0.0;   // the E reads of the reference's hash build: not streamed here
            // what the kernel really reads: dim + read_ts of every chunk in use, the E tables unless they are recomputed, the row -> lookup map
            hash_design_reads = (lean_e ? 0.0 : (double)N * 8 * (double)hm.size()) + (double)N;
            for (int c = 0; c < 4; c++) if (chunk_used[c]) hash_design_reads += (double)N * 8 * 2;
        }
        u64* H1 = (any_gp1 && !emit) ? ctx->alloc_n<u64>((size_t)nrows * N) : nullptr;
        u64* L1 = (any_gp1 && !emit && gp_deepest(nu, gp1_owner) >= 1) ? ctx->alloc_n<u64>((size_t)nrows * (N / 2)) : nullptr;
        u64* H2 = do_gp2 ? ctx->alloc_n<u64>((size_t)nrows * M) : nullptr;   // sharded: the rows of local_pairs only
        // hash launches are grouped by chunk: the memories of a chunk share the dim / read_ts columns
        struct HashReq { int i; u64 *rd, *wr, *rd1, *wr1; };
        std::vector<HashReq> reqs;
        auto hash_rw = [&](int i, u64* rd, u64* wr, u64* rd1, u64* wr1) { reqs.push_back({i, rd, wr, rd1, wr1}); };
        if (emit > 0) {
            // nothing to hash here
        } else if (any_gp1 && !split) {
            for (int i = 0; i < G; i++)
                hash_rw(i, H1 + (size_t)i * N, H1 + (size_t)(G + i) * N, L1 ? L1 + (size_t)i * (N / 2) : nullptr, L1 ? L1 + (size_t)(G + i) * (N / 2) : nullptr);
        } else if (any_gp1) {
            const int base = p0_only ? 1 : 0, nlm = (int)local_mems.size();
            if (p0_only) {  // table 0 (reads of the first memory) supplies p_0; its write table is not needed here
                u64* junk = ctx->alloc_n<u64>(N + N / 2);
                hash_rw(0, H1, junk, L1 ? L1 : nullptr, L1 ? junk + N : nullptr);
            }
            for (int q = 0; q < nlm; q++)
                hash_rw(local_mems[q], H1 + (size_t)(base + q) * N, H1 + (size_t)(base + nlm + q) * N,
                        L1 ? L1 + (size_t)(base + q) * (N / 2) : nullptr, L1 ? L1 + (size_t)(base + nlm + q) * (N / 2) : nullptr);
        }
        for (int c = 0; c < 4; c++) {
            std::vector<HashReq> of_c;
            for (auto& r : reqs) if (lp.gkr_chunk[r.i] == c) of_c.push_back(r);
            for (size_t o = 0; o < of_c.size(); o += dev::HASH_RW_MAX) {
                const int cnt = (int)std::min<size_t>(dev::HASH_RW_MAX, of_c.size() - o);
                dev::HashRwArgs ha;
                memset(&ha, 0, sizeof(ha));
                for (int q = 0; q < cnt; q++) {
                    const HashReq& r = of_c[o + q];
                    ha.ep[q] = epm(lp.gkr_order[r.i]); ha.rd[q] = r.rd; ha.wr[q] = r.wr; ha.rd1[q] = r.rd1; ha.wr1[q] = r.wr1;
                }
                ctx->prof_begin(cls_hash, (double)N * 8 * (2 + cnt * (L1 ? 4 : 3)));
                dev::lasso_hash_rw(st, N, dims + (size_t)c * N, read_ts[c], ha, cnt, gamma, tau);
                ctx->prof_end();
            }
        }
        if (do_gp2) aux([&] {
            if (G > 32) throw Error("lasso: more than 32 memories");
            dev::HashIfArgs ha;
            memset(&ha, 0, sizeof(ha));
            for (int i = 0; i < G; i++) {
                ha.cutoff[i] = (u32)lp.mems[lp.gkr_order[i]].cutoff;
                ha.row_init[i] = split ? -1 : i; ha.row_fin[i] = split ? -1 : G + i;
                if (split)
                    for (size_t q = 0; q < local_pairs.size(); q++) {
                        if (local_pairs[q] == i) ha.row_init[i] = (int)q;
                        if (local_pairs[q] == G + i) ha.row_fin[i] = (int)q;
                    }
                ha.fc[i] = ha.row_fin[i] >= 0 ? final_cts[lp.gkr_chunk[i]] : nullptr;   // (init hashes do not read the counters)
                if (ha.row_fin[i] >= 0 && !ha.fc[i]) throw Error("lasso: counters of a needed chunk were not computed");
            }
            ctx->prof_begin(cls_hash, (double)nrows * M * 8 * 1.5);
            dev::lasso_hash_if(st, ha, G, gamma, tau, H2);
            ctx->prof_end();
        });
        mark("lasso: memory checking, grand product #1 over reads then writes (prover.rs:161-165)");
        // the write hash of a row is its read hash + gamma^2 (prover.rs:44: t + 1): the top layer runs on the read rows only
        constexpr bool use_mirror = true;
        const u64 gamma_sq = gl_mul(gamma, gamma);
        const u64* mirror_c = (emit > 0 && use_mirror) ? &gamma_sq : nullptr;
        GpOut g1 = split ? grand_product(H1, N, 2 * G, gp1_owner, L1, &local_pairs, p0_only, d_hash_src, emit, hash_build_bytes, mirror_c)
                         : grand_product(H1, N, 2 * G, gp1_owner, L1, nullptr, false, d_hash_src, emit, hash_build_bytes, mirror_c);  // reads then writes (prover.rs:161-165)
        mark("lasso: memory checking, grand product #2 over inits then finals (prover.rs:167-171)");
        GpOut g2{0, nullptr};
        aux([&] {   // inits then finals (prover.rs:167-171); its tree on the second stream
            g2 = split ? grand_product(H2, M, 2 * G, std::vector<int>(16, rank), nullptr, &local_pairs, p0_only)
                       : grand_product(H2, M, 2 * G, std::vector<int>(16, rank));
        });
        if (use_aux) {
            // grand product #1's first launch reads the counters, grand product #2's first rounds its tree: the main stream waits
            // for the second one only there, after the collation sum-check has been enqueued
            on_aux([&] { stamp("grand product #2's tree done"); });
            hip_check(hipEventRecord(ctx->ev_aux[1], ctx->stream2), "lasso: aux event");
            on_aux([&] { hip_check(hipStreamWaitEvent(st, ctx->ev_aux[2], 0), "lasso: wait for the E tables"); });   // claimed sum, openings
            // (the claimed sum's launches follow the grand products' below: see the openings)
            hg_ctx* c = ctx;
            // the first launches (the hash-free first round of the top layer, the next layers' level-emitting first rounds) only
            // need the counters; grand product #2's jobs join from the mixed first-round launch on and need its tree
            const bool two_waits = counters_event_recorded;
            st_before_gp = [c, two_waits] { hip_check(hipStreamWaitEvent(c->stream, c->ev_aux[two_waits ? 3 : 1], 0), "lasso: wait for the counters"); };
            if (two_waits) st_before_gp2 = [c] { hip_check(hipStreamWaitEvent(c->stream, c->ev_aux[1], 0), "lasso: wait for grand product #2's tree"); };
        }
        // openings (prover.rs:173-178, mod.rs:80-93). With two streams eq(r, .) may still be in use by the claimed-sum kernel on
        // the main stream, so the openings get their own table.
        // Their launches (and the claimed sum's) are enqueued AFTER the grand products': a replayed launch graph submits its nodes in
        // the order they were recorded, a few microseconds each, so the main stream's first grand-product kernel could not start before
        // everything recorded ahead of it was out - on a sharded rank, whose kernels are short, that left the main stream idle for
        // 100-150 us (scripts/ub/graph_order.hip shows the effect in isolation).
        E2* eqx = (use_aux && do_open) ? ctx->alloc_n<E2>(N) : eq;  // (one stream: the eq(r,.) table is dead by now)
        E2* eqy = do_open ? ctx->alloc_n<E2>(M) : nullptr;
        // every opening at x in one launch, every opening at y in another (dev::dot_eq_many): results land in their wire slots
        dev::DotTabs tx, ty;
        memset(&tx, 0, sizeof(tx)); memset(&ty, 0, sizeof(ty));
        int nx = 0, ny = 0;
        {
        auto add_x = [&](const u64* tab, size_t out_slot) {
            if (nx >= dev::DOT_MAX) throw Error("lasso: too many openings");
            tx.t[nx] = tab; tx.slot[nx] = (int)out_slot; nx++;
        };
        struct ChunkSlots { int c; size_t base, count; };
        std::vector<ChunkSlots> chunk_slots;
        for (auto& chk : lp.chunks) {
            const int c = chk.first;
            // order on the wire: dim(x), read_ts(x), final_cts(y), then E_m(x)
            // sharded: dim(x), read_ts(x), final_cts(y) of a chunk belong to the owner of its first memory, E_m(x) to the owner of m
            const bool own_chunk = do_open && own_mem[chk.second[0]];
            const size_t base_slot = slot(3 + chk.second.size());
            if (own_chunk) {
                add_x(dims + (size_t)c * N, base_slot);
                add_x(read_ts[c], base_slot + 1);
                ty.t[ny] = final_cts[c]; ty.slot[ny] = (int)(base_slot + 2); ny++;
            }
            for (size_t i = 0; i < chk.second.size(); i++)
                if (do_open && own_mem[chk.second[i]]) {
                    if (lean_e) { if (nx >= dev::DOT_MAX) throw Error("lasso: too many openings"); tx.t[nx] = nullptr; tx.emem[nx] = (signed char)chk.second[i]; tx.slot[nx] = (int)(base_slot + 3 + i); nx++; }
                    else add_x(epm(chk.second[i]), base_slot + 3 + i);
                }
            chunk_slots.push_back({c, base_slot, 3 + chk.second.size()});
        }
        for (auto& cs : chunk_slots) {
            mark("lasso: openings of chunk " + std::to_string(cs.c) + ": dim(x), read_ts(x), final_cts(y), E_m(x) (prover.rs:173-178)");
            defer_write_slots(cs.base, cs.count);
        }
        }
        flush_stride();  // collation + every grand-product layer, round-synchronised
        aux([&] { if (do_col && claim_late) do_claim(); });
        {
            // the openings (two dot-product launches over every opened table: bandwidth) by value, so that they can also run later
            const dev::DotTabs txv = tx, tyv = ty;
            const int nxv = nx, nyv = ny;
            const bool do_open_v = do_open, lean_v = lean_e;
            E2 *eqx_v = eqx, *eqy_v = eqy;
            const size_t p1 = g1.point_off, p2 = g2.point_off;
            const dev::LassoDev* Lp = &L;
            // the two eq tables of the opening points are challenges only: built right away on the second stream (beside the first hash
            // round), whatever the place of the dot products - at the end of the second stream they ran alone, 0.05 ms of its length
            auto open_tables = [this, do_open_v, eqx_v, eqy_v, p1, p2, nu] {
                if (do_open_v) {
                    eq_now(eqx_v, nu, p1);
                    eq_now(eqy_v, 16, p2);
                }
            };
            auto openings = [this, txv, tyv, nxv, nyv, lean_v, eqx_v, eqy_v, Lp, d_input, N] {
                if (nxv) {
                    int nvirt = 0;
                    for (int t = 0; t < nxv; t++) nvirt += txv.t[t] == nullptr;
                    // (a group of eight re-reads eq; recomputed E tables cost their group one 8-byte input read per entry)
                    ctx->prof_begin(cls_aux, (double)N * (16.0 * ((nxv + 7) / 8) + 8.0 * (nxv - nvirt) + (nvirt ? 8.0 * ((nxv + 7) / 8) : 0.0)),
                                    (double)N * (16.0 * ((nxv + 7) / 8) + 8.0 * nxv));   // (reference model: every opened table is read)
                    dev::DotVirt dv;
                    memset(&dv, 0, sizeof(dv));
                    if (lean_v) {
                        const dev::LassoDev& L = *Lp;
                        dv.input = d_input; dv.seg_lookup = L.seg_lookup; dv.seg_shift = L.seg_shift; dv.rows = L.rows;
                        memcpy(dv.lookup_mask, L.lookup_mask, sizeof(dv.lookup_mask)); memcpy(dv.lookup_uses, L.lookup_uses, sizeof(dv.lookup_uses));
                        memcpy(dv.mem_dim, L.mem_dim, sizeof(dv.mem_dim)); memcpy(dv.mem_cutoff, L.mem_cutoff, sizeof(dv.mem_cutoff));
                    }
                    if (!(lean_v && dev::open_x(st, eqx_v, txv, nxv, N, partials, d_res(), dv)))
                        dev::dot_eq_many(st, eqx_v, txv, nxv, N, partials, d_res(), lean_v ? &dv : nullptr);
                    ctx->prof_end();
                }
                if (nyv) dev::dot_eq_many(st, eqy_v, tyv, nyv, M, partials, d_res());
                stamp("claimed sum and openings done");
            };
            // Where the openings' dot products run (the two eq tables above are built right away in every case): behind the node
            // reductions, on THEIR stream (one rank: the third one, which waits for the opening tables of the second; a sharded rank:
            // the second). Since the eq-factored node reductions (round 5) those end 0.25 ms before the grand products do. Measured and
            // removed: right away beside the first hash round (1.846 ms), between the two waves of node reductions (1.864), at the end of
            // the main stream behind the grand products (1.87-1.91; the round-4 default) against 1.83-1.86 ms; a fourth stream for the
            // node reductions alone: 1.85 - a replayed launch graph does not run a fourth branch beside the other three (its first
            // kernel starts 0.6 ms into the prove, with or without GPU_MAX_HW_QUEUES=8).
            aux(open_tables);
            if (use_aux && world == 1) {
                hip_check(hipEventRecord(ctx->ev_aux[4], ctx->stream2), "lasso: opening tables event");
                late_col.push_back(openings);
            } else if (use_aux) late_aux.push_back(openings);
            else aux(openings);
        }
        stamp("grand products done");
        return ClaimRef{r_off, nu, claimed};  // (r, claimed_sum) for the single predecessor (lasso.rs:97,113)
    }

    // ---- GKR driver ----------------------------------------------------------------------------
    std::map<std::pair<size_t, int>, E2*> eq_shared;             // (point, variables) -> eq table of a single unit-weight claim
    std::map<std::tuple<size_t, int, int>, E2*> fft_shared;      // (point, log2 size, inverse) -> DFT-row table of a single claim
    static bool share_tables() { return true; }
    std::vector<const u64*> d_vals;               // node values in HBM
    std::vector<std::vector<ClaimRef>> claims;    // per node

    dev::ClaimSet claim_set(const std::vector<ClaimRef>& cl, std::vector<E2>* alphas_host) {
        dev::ClaimSet cs;
        memset(&cs, 0, sizeof(cs));
        if (cl.empty()) throw Error("gkr: node without claim");
        if (cl.size() > (size_t)dev::MAX_CLAIMS) throw Error("gkr: too many claims on one node");
        cs.n = (int)cl.size();
        cs.unit_alpha = cl.size() == 1;
        cs.alpha_off = epos();
        alphas_host->clear();
        if (cl.size() > 1) for (size_t a = 0; a < cl.size(); a++) alphas_host->push_back(squeeze());
        else alphas_host->push_back(e2_one());
        for (size_t a = 0; a < cl.size(); a++) cs.point_off[a] = cl[a].point_off;
        return cs;
    }
    Cell combined_value(const std::vector<ClaimRef>& cl, const std::vector<E2>& alphas) {
        Cell v = cell();
        std::vector<Cell> vals;
        for (auto& c : cl) vals.push_back(c.value);
        push_op([v, vals, alphas] {
            E2 s = e2_zero();
            for (size_t a = 0; a < vals.size(); a++) s = e2_add(s, e2_mul(*vals[a], alphas[a]));
            *v = s;
        });
        return v;
    }

    void vanilla_node(int id) {
        const HNode& n = pk->circuit.nodes[id];
        const hg_pk::NodeDev& nd = pk->node_dev[id];
        const int nin = n.log2_sub_in + n.log2_reps;
        const size_t SR = (size_t)1 << nin;
        std::vector<E2> alphas;
        dev::ClaimSet cs = claim_set(claims[id], &alphas);
        for (auto& c : claims[id]) if (c.len != n.log2_out()) throw Error("gkr: claim arity mismatch");
        Cell claim = combined_value(claims[id], alphas);
        const bool own = mine(node_owner[id]);
        std::vector<int> li, ri;
        for (int i = 0; i < n.arity; i++) { if (n.left_use[i]) li.push_back(i); if (n.right_use[i]) ri.push_back(i); }
        if (n.arity > dev::PS_MAX_PAIRS) throw Error("vanilla: arity too large");
        // phase 1's transcript side, common to both forms below
        auto phase1_steps = [&](const ScHandle& s1, size_t u_base, Cell after1) {
            mark("vanilla node " + std::to_string(id) + ": Libra phase 1 sum-check, " + std::to_string(nin) + " rounds x 3 coefficients [G1 node order, G2 alpha per claim, G3 Libra form]");
            defer_sumcheck(s1, 2, claim, after1);
            for (int i : li) {
                mark("vanilla node " + std::to_string(id) + ": input " + std::to_string(i) + " evaluation at r_x");
                defer_write_slots(u_base + i, 1);
                Cell v = cell();
                size_t sl = u_base + i;
                push_op([this, v, sl] { *v = h_res()[sl]; });
                claims[n.preds[i]].push_back(ClaimRef{s1.point_off, nin, v});
            }
        };
        // Eq-factored form (kernels.hpp PsJob::eq_n; found at setup, hg_pk::NodeDev::EqForm): the node relays aligned blocks, so with
        // ONE claim at z every phase-1 table is kappa_i eq(z', .), z' = (z_0 .. z_(w-1), bits of hib), kappa_i = sum_t coef_t
        // eq(z_(w..); block_t). Neither the node's eq table nor any bookkeeping table is built; additive constants, which are constant
        // over those blocks, leave the claim as a scalar of the challenges.
        const hg_pk::NodeDev::EqForm& ef = nd.eq_form;
        const int eq_tail = ef.ok && cs.n == 1 && ps_eq_on() ? eq_tail_rd((int)li.size(), nin) : -1;
        if (eq_tail >= 0) {
            const u64* chain = challenge_chain(2 * (cs.point_off[0] + n.log2_out()));
            auto zc = [&](int k) { return e2(chain[2 * (cs.point_off[0] + k)], chain[2 * (cs.point_off[0] + k) + 1]); };
            const int hb = n.log2_out() - ef.w;
            auto eq_hi = [&](u32 block) {
                E2 acc = e2_one();
                for (int b = 0; b < hb; b++) acc = e2_mul(acc, (block >> b) & 1 ? zc(ef.w + b) : e2_sub(e2_one(), zc(ef.w + b)));
                return acc;
            };
            if (!ef.consts.empty()) {
                E2 c = e2_zero();
                for (auto& t : ef.consts) c = e2_add(c, e2_mul_f(eq_hi(t.second), t.first));
                push_op([claim, c] { *claim = e2_sub(*claim, c); });
            }
            std::vector<const u64*> a;
            std::vector<E2> kappa;
            std::vector<E2*> fa, fb;
            const size_t u_base = slot(n.arity);
            E2* scratch = own ? ctx->alloc_n<E2>(n.arity) : nullptr;
            for (int i : li) {
                E2 k = e2_zero();
                for (auto& t : ef.terms[i]) k = e2_add(k, e2_mul_f(eq_hi(t.second), t.first));
                a.push_back(d_vals[n.preds[i]]);
                kappa.push_back(k);
                fa.push_back(d_res() + u_base + i);
                fb.push_back(scratch + i);
            }
            std::vector<E2> zp(nin);
            for (int k = 0; k < nin; k++) zp[k] = k < ef.w ? zc(k) : e2((ef.hib >> (k - ef.w)) & 1u, 0);
            ScHandle s1 = sc_prodsum_eq(a, kappa, zp, cs.point_off[0], ef.w, ef.hib, nin, eq_tail, fa, fb, own);
            phase1_steps(s1, u_base, cell());
            return;
        }
        // Nodes that received their only claim from the same sum-check share the point (every input of a Vanilla node is opened at the
        // node's r_x: the five inputs of the final sum, the eight chunk nodes behind the Lasso input ...): one eq table serves them all
        // (read-only everywhere).
        E2* eqc = nullptr;
        if (own) {
            const std::pair<size_t, int> key{cs.point_off[0], n.log2_out()};
            auto hit = cs.n == 1 && share_tables() ? eq_shared.find(key) : eq_shared.end();
            if (hit != eq_shared.end()) eqc = hit->second;
            else {
                eqc = ctx->alloc_n<E2>((size_t)1 << n.log2_out());
                queue_eq(eqc, n.log2_out(), cs);
                if (cs.n == 1) eq_shared[key] = eqc;
            }
        }
        const hg_pk::NodeDev* ndp0 = &nd;
        const HNode* np0 = &n;
        if (nd.nconst) {  // claim -= sum_g eqc[g] * w0_g
            size_t s = slot(1);
            if (own) after_eq.push_back([this, ndp0, np0, eqc, s] {
                int grid = dev::vanilla_const_sum(st, ndp0->const_gate, ndp0->const_coef, ndp0->nconst, eqc, np0->log2_sub_out, np0->log2_reps, partials);
                reduce(grid, 1, s);
            });
            push_op([this, s, claim] { *claim = e2_sub(*claim, h_res()[s]); });
        }
        // phase 1: sum_x sum_i in_i(x) T_i(x)
        dev::GatherT gt;
        memset(&gt, 0, sizeof(gt));
        for (int i = 0; i < n.arity; i++) gt.in_vals[i] = d_vals[n.preds[i]];
        std::vector<const u64*> a;
        std::vector<const E2*> b;
        std::vector<E2*> fa, fb;
        size_t u_base = slot(n.arity);
        E2* scratch = own ? ctx->alloc_n<E2>(n.arity) : nullptr;
        // HG_GATHER_CSR=1: the general (per-term) form for every table
        static const bool use_seg = !hg_env_on("HG_GATHER_CSR");
        for (int i : li) {
            const hg_pk::NodeDev::Seg& sg = nd.seg[i];
            E2* T = nullptr;
            if (own && use_seg && sg.alias) T = eqc + sg.alias_off;   // one unit relay per position: the table is a slice of eqc
            else if (own) T = ctx->alloc_n<E2>(SR);
            gt.lin = nd.lin[i];
            gt.mul = nd.mulL[i];
            if (own && use_seg && sg.alias) {
            } else if (own && use_seg && sg.d) {
                dev::GatherSegJob sj;
                memset(&sj, 0, sizeof(sj));
                sj.segs = sg.d; sj.nseg = sg.nseg; sj.eqc = eqc; sj.log2_S = n.log2_sub_in; sj.log2_G = n.log2_sub_out; sj.log2_R = n.log2_reps; sj.T = T;
                for (int q = 0; q < n.arity; q++) sj.in_vals[q] = gt.in_vals[q];
                gather_seg_queue.push_back(sj);
            } else if (own) gather_queue.push_back(dev::GatherJob{gt, eqc, n.log2_sub_in, n.log2_sub_out, n.log2_reps, T});
            a.push_back(d_vals[n.preds[i]]);
            b.push_back(T);
            fa.push_back(d_res() + u_base + i);
            fb.push_back(scratch + i);
        }
        ScHandle s1 = sc_prodsum(a, b, nin, fa, fb, own);
        Cell after1 = cell();
        phase1_steps(s1, u_base, after1);
        if (!n.mul.empty()) {
            if (!n.lin.empty()) throw Error("vanilla: nodes mixing linear and mul gates are not on this path");
            // phase 2: sum_y sum_i in_i(y) B_i(y), claim carried over from phase 1 (no linear part).
            // Transcript bookkeeping (challenges, slots) happens now, in protocol order; the device work
            // needs u = in(r_x) from phase 1 and is therefore queued for the second wave.
            std::vector<const u64*> a2;
            std::vector<E2*> Bs, fa2, fb2;
            size_t w_base = slot(n.arity);
            for (int i : ri) {
                E2* B = own ? ctx->alloc_n<E2>(SR) : nullptr;
                a2.push_back(d_vals[n.preds[i]]);
                Bs.push_back(B);
                fa2.push_back(d_res() + w_base + i);
                fb2.push_back(scratch + i);
            }
            size_t rx_off = s1.point_off;
            const hg_pk::NodeDev* ndp = &nd;
            const HNode* np = &n;
            // reserve the phase-2 sum-check (challenges + slots) now; attach its tables in the second wave
            ScHandle s2;
            s2.nv = 2; s2.nvars = nin; s2.point_off = epos(); s2.sums_slot = slot((size_t)nin * 2);
            for (int i = 0; i < nin; i++) s2.rs.push_back(squeeze());
            if (own) second_wave.push_back([this, ri, Bs, a2, fa2, fb2, rx_off, ndp, np, eqc, u_base, SR, nin, s2] {
                E2* eqx = ctx->alloc_n<E2>(SR);
                dev::ClaimSet one;
                memset(&one, 0, sizeof(one));
                one.n = 1; one.unit_alpha = 1; one.point_off[0] = rx_off;
                queue_eq(eqx, nin, one);
                for (size_t q = 0; q < ri.size(); q++)  // batched by flush_bookkeeping, after the eq tables
                    gatherB_queue.push_back(dev::GatherBJob{ndp->mulR[ri[q]], eqc, eqx, d_res() + u_base, np->log2_sub_in, np->log2_sub_out, np->log2_reps, Bs[q]});
                const int npairs = (int)ri.size();
                const size_t N = (size_t)1 << nin;
                dev::PsJob J;
                memset(&J, 0, sizeof(J));
                J.npairs = npairs; J.nvars = nin; J.r_off = s2.point_off; J.sums_slot = s2.sums_slot;
                for (int q = 0; q < 2; q++) {
                    J.bufa[q] = ctx->alloc_n<E2>((size_t)npairs * std::max<size_t>(N >> (q + 1), 1));
                    J.bufb[q] = ctx->alloc_n<E2>((size_t)npairs * std::max<size_t>(N >> (q + 1), 1));
                }
                for (int q = 0; q < npairs; q++) { J.a[q] = a2[q]; J.b[q] = Bs[q]; J.fin_a[q] = fa2[q]; J.fin_b[q] = fb2[q]; }
                ps_queue[nin].push_back(J);
            });
            mark("vanilla node " + std::to_string(id) + ": Libra phase 2 sum-check [G3]");
            defer_sumcheck(s2, 2, after1, nullptr);
            for (int i : ri) {
                mark("vanilla node " + std::to_string(id) + ": input " + std::to_string(i) + " evaluation at r_y");
                defer_write_slots(w_base + i, 1);
                Cell v = cell();
                size_t sl = w_base + i;
                push_op([this, v, sl] { *v = h_res()[sl]; });
                claims[n.preds[i]].push_back(ClaimRef{s2.point_off, nin, v});
            }
        }
    }

    void fft_node(int id) {
        const HNode& n = pk->circuit.nodes[id];
        const int L = n.log2_size;
        const size_t N = (size_t)1 << L;
        std::vector<E2> alphas;
        dev::ClaimSet cs = claim_set(claims[id], &alphas);
        Cell claim = combined_value(claims[id], alphas);
        const bool own = mine(node_owner[id]);
        // (the sixteen inverse-FFT nodes behind sai_par are all opened at sai_par's r_x: one DFT-row table)
        E2* F = nullptr;
        bool F_shared = false;
        if (own) {
            const std::tuple<size_t, int, int> key{claims[id].size() == 1 ? claims[id][0].point_off : (size_t)-1, L, n.inverse ? 1 : 0};
            auto hit = claims[id].size() == 1 && share_tables() ? fft_shared.find(key) : fft_shared.end();
            if (hit != fft_shared.end()) { F = hit->second; F_shared = true; }
            else {
                F = ctx->alloc_n<E2>(N);
                if (claims[id].size() == 1) fft_shared[key] = F;
            }
        }
        const u64* W = (n.inverse ? pk->w_inv : pk->w_fwd).at(L);
        u64 scale = n.inverse ? gl_inv(gl_from_u64(N)) : 1;
        if (own && !F_shared) fft_queue.push_back(dev::FftJob{F, W, scale, L, cs});
        size_t u = slot(1);
        E2* scratch = own ? ctx->alloc_n<E2>(1) : nullptr;
        ScHandle s = sc_prodsum({d_vals[n.preds[0]]}, {F}, L, {d_res() + u}, {scratch}, own);
        mark("fft node " + std::to_string(id) + ": zkCNN sum-check, " + std::to_string(L) + " rounds x 3 coefficients [G3 zkCNN form, G4 root of unity]");
        defer_sumcheck(s, 2, claim, nullptr);
        mark("fft node " + std::to_string(id) + ": input evaluation");
        defer_write_slots(u, 1);
        Cell v = cell();
        push_op([this, v, u] { *v = h_res()[u]; });
        claims[n.preds[0]].push_back(ClaimRef{s.point_off, L, v});
    }

    // The Vanilla / FFT node reductions run on a second stream, concurrently with the Lasso node: they are independent on the
    // device (claim points are challenges) and mostly small launches that leave CUs idle, while the Lasso node's critical
    // path has its own latency-bound stretches (counter sorts, openings, last rounds). The fork point is recorded BEFORE the
    // Lasso node is enqueued (record_fork, at the start of the walk); HG_ONE_STREAM=1 keeps everything on one stream.
    bool fork_recorded = false;
    void record_fork() {
        static const bool one_stream = hg_env_on("HG_ONE_STREAM");
        if (one_stream || ctx->one_stream) return;
        hip_check(hipEventRecord(ctx->ev_fork, ctx->stream), "fork event");   // after the result-buffer clear / ticket reset
        fork_recorded = true;
    }
    // Runs `fn` with the second stream as the enqueue target (work of the Lasso node that is off its critical path: counter
    // sorts, grand product #2's hashes and tree, the openings). One stream only: runs it in place.
    std::vector<std::function<void()>> late_aux;
    std::vector<std::function<void()>> late_col;   // the same, for whichever stream the node reductions run on; needs ev_aux[4]
    bool aux_started = false;
    template <typename Fn> void on_aux(Fn fn) {
        if (!fork_recorded) { fn(); return; }
        hipStream_t s0 = st;
        E2* p0 = partials;
        if (!aux_started) { hip_check(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0), "aux: fork wait"); aux_started = true; }
        st = ctx->stream2; partials = ctx->d_partials2; ctx->prof_stream = st;
        fn();
        st = s0; partials = p0; ctx->prof_stream = s0;
    }
    bool col_pending = false;
    template <typename Fn> void on_col(Fn fn) {   // the collation rounds on their own stream: behind the E tables, joined at the end of the prove
        hipStream_t s0 = st;
        E2* p0 = partials;
        hip_check(hipStreamWaitEvent(ctx->stream_col, ctx->ev_aux[2], 0), "collation: wait for the E tables");
        st = ctx->stream_col; partials = ctx->d_partials3; ctx->prof_stream = st;
        fn();
        hip_check(hipEventRecord(ctx->ev_col, ctx->stream_col), "collation: done event");
        st = s0; partials = p0; ctx->prof_stream = s0;
        col_pending = true;
    }
    std::function<void()> st_before_gp;  // flush_stride runs it once before the first grand-product launch (cross-stream wait)
    std::function<void()> st_before_gp2; // ... and this one once the sequenced first rounds of grand product #1's top layers are out
    // The node reductions on the THIRD stream (the collation rounds' - a few launches at the start of the prove): they then start with
    // the prove instead of behind the Lasso node's second-stream work (counters, grand product #2's tree, opening tables: 0.6 ms),
    // into the idle capacity those latency-bound launches leave.
    bool nodes_on_col = false, counters_event_live = false;   // (the second: ev_aux[3] was recorded by this prove's Lasso node)
    void fork_nodes_stream() {
        if (!fork_recorded) return;
        hip_check(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0), "fork wait");
        if (world == 1 && late_aux.empty()) {
            hip_check(hipStreamWaitEvent(ctx->stream_col, ctx->ev_fork, 0), "fork wait");
            // (behind the counters of the second stream: the first hash round waits for those, and the node reductions' first launches
            // beside them only stretch that wait - 1.79 against 1.81 ms)
            if (counters_event_live) hip_check(hipStreamWaitEvent(ctx->stream_col, ctx->ev_aux[3], 0), "node reductions: wait for the counters");
            st = ctx->stream_col; partials = ctx->d_partials3; ctx->prof_stream = st; forked = true; nodes_on_col = true;
            return;
        }
        st = ctx->stream2; partials = ctx->d_partials2; ctx->prof_stream = st; forked = true;
    }
    void join_nodes_stream() {
        if (nodes_on_col) { hip_check(hipEventRecord(ctx->ev_col, ctx->stream_col), "node reductions: done event"); col_pending = true; nodes_on_col = false; }
        if (col_pending) { hip_check(hipStreamWaitEvent(ctx->stream, ctx->ev_col, 0), "collation: join wait"); col_pending = false; }
        if (!forked) return;
        hip_check(hipEventRecord(ctx->ev_join, ctx->stream2), "join event");
        st = ctx->stream; partials = ctx->d_partials; ctx->prof_stream = st;
        hip_check(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0), "join wait");
    }
    void gkr(const ClaimRef& sum_claim) {  // prove_gkr (sk_encryption_circuit.rs:455-457)
        const HCircuit& c = pk->circuit;
        claims.assign(c.nodes.size(), {});
        claims[c.lasso_id].push_back(ClaimRef{epos(), 0, cell()});  // EvalClaim::new(vec![], E::ZERO) (:450)
        claims[c.sum_id].push_back(sum_claim);
        record_fork();
        // nodes whose claims descend from the Lasso node (the node itself and, transitively, its predecessors): their transcript steps
        // wait for the Lasso node's results; every other node's steps can be replayed as soon as the node reductions are done
        std::vector<char> lasso_cone(c.nodes.size(), 0);
        {
            std::vector<int> stack{c.lasso_id};
            lasso_cone[c.lasso_id] = 1;
            while (!stack.empty()) {
                const int u = stack.back(); stack.pop_back();
                for (int pr : c.nodes[u].preds) if (!lasso_cone[pr]) { lasso_cone[pr] = 1; stack.push_back(pr); }
            }
        }
        const bool early_ok = world == 1 && fork_recorded && early_replay_on();
        for (size_t q = c.topo.size(); q-- > 0;) {
            int id = c.topo[q];
            const HNode& n = c.nodes[id];
            cur_early = early_ok && !lasso_cone[id];
            switch (n.kind) {
                case NK_INPUT: break;
                case NK_VANILLA: vanilla_node(id); break;
                case NK_FFT: fft_node(id); break;
                case NK_LASSO: {
                    ClaimRef cr = lasso_node(d_vals[n.preds[0]]);
                    claims[n.preds[0]].push_back(cr);
                    break;
                }
            }
        }
        cur_early = false;
        // The Vanilla / FFT node reductions go to the second stream: they are independent of the Lasso node on the
        // device and consist mostly of small launches that leave CUs idle, so the two streams overlap.
        fork_nodes_stream();
        stamp("node reductions begin");
        flush_bookkeeping();                   // eq tables, constant sums, Libra gathers, DFT-row tables: batched
        stamp("node bookkeeping done");
        flush_prodsum();                       // first wave: every FFT / Libra phase-1 reduction, batched
        stamp("node phase 1 done");

        for (auto& f : second_wave) f();       // Libra phase-2 bookkeeping (needs the phase-1 scalars in HBM)
        second_wave.clear();
        flush_bookkeeping();
        flush_prodsum();
        stamp("node reductions done");
        if (early_ok) {   // every result an early transcript step reads is in the buffer: tell the host (out-of-order replay, see `ops`)
            early_slot = slot(1);
            dev::set_e2(st, d_res() + early_slot, e2(1, 0));
        }
        for (auto& f : late_aux) f();          // the Lasso node's openings (lasso_node)
        late_aux.clear();
        if (!late_col.empty()) {
            if (st != ctx->stream2) hip_check(hipStreamWaitEvent(st, ctx->ev_aux[4], 0), "openings: wait for the opening tables");
            for (auto& f : late_col) f();
            late_col.clear();
        }
        join_nodes_stream();
    }

    // copies the result buffer back (the only synchronisation) and replays the transcript
    double t_enqueued = 0, t_synced = 0, t_replayed = 0;
    // debugging aid (HG_STAMP=1): one-thread kernels that record the device clock at named points of both streams; printed after
    // the synchronisation, relative to the first one. Works inside a replayed launch graph, where events and profilers do not.
    std::vector<std::string> stamp_names;
    unsigned long long* d_stamps = nullptr;
    void stamp(const char* name) {
        static const bool on = hg_env_on("HG_STAMP");
        if (!on) return;
        if (!d_stamps) d_stamps = ctx->alloc_n<unsigned long long>(256);
        if (stamp_names.size() >= 256) return;
        dev::stamp(st, d_stamps + stamp_names.size());
        stamp_names.push_back(std::string(st == ctx->stream ? "main " : (st == ctx->stream_col ? "col  " : "aux  ")) + name);
    }
    void print_stamps() {
        if (stamp_names.empty()) return;
        std::vector<unsigned long long> h(stamp_names.size());
        if (hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
        for (size_t i = 0; i < h.size(); i++) fprintf(stderr, "stamp %8.1f us  %s\n", (double)(long long)(h[i] - h[0]) / 100.0, stamp_names[i].c_str());
    }
    // `done` (hg_prove_stream): an event already recorded behind this prove - the stream may hold the NEXT prove by now
    void sync_results(hipEvent_t done = nullptr) {
        if (res_used && ctx->d_res != ctx->h_res) hip_check(hipMemcpyAsync(ctx->h_res, ctx->d_res, res_used * sizeof(E2), hipMemcpyDeviceToHost, st), "copy results");
        t_enqueued = wall_ms();
        // The only synchronisation of a prove. Spin on an event instead of hipStreamSynchronize: a blocking wait can cost
        // up to milliseconds of wake-up latency when another runtime in the process (PyTorch in bench.py) has switched the
        // device to blocking-sync scheduling; the wait is a few milliseconds at most, so a busy core is the cheaper price.
        if (!done) { hip_check(hipEventRecord(ctx->ev_join, st), "prove: done event"); done = ctx->ev_join; }
        const double t_spin = wall_ms();
        for (;;) {
            hipError_t q = hipEventQuery(done);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) hip_check(q, "prove: event query");
            if (early_ready()) run_early();   // (the node reductions are done, the Lasso node's last launches are not)
            if (wall_ms() - t_spin > 2000.0) { hip_check(hipEventSynchronize(done), "prove: event sync"); break; }
        }
        hip_check(hipGetLastError(), "prove: kernel launch");
        t_synced = wall_ms();
        print_stamps();
    }
    void replay() {
        double t = wall_ms();
        proof.bytes.reserve((size_t)1 << 18);
        const size_t nops = ops.size();
        if (early_done) {   // the early steps are in place: the rest, each at its recorded offset
            for (size_t i = 0; i < nops; i++)
                if (!op_early[i]) { proof.pos = op_off[i]; ops[i](); }
            proof.pos = (size_t)-1;
            early_done = false;
        } else {
            const bool rec = !offsets_known;
            if (rec) op_off.assign(nops + 1, 0);
            for (size_t i = 0; i < nops; i++) {
                if (rec) op_off[i] = proof.bytes.size();
                else if (op_off[i] != proof.bytes.size()) throw Error("transcript replay: a step's output moved (its length depends on the witness?)");
                ops[i]();
            }
            if (rec) { op_off[nops] = proof.bytes.size(); offsets_known = true; }
        }
        if (const char* path = hg_proof_map_path()) {
            if (FILE* f = fopen(path, "w")) {
                for (auto& m : proof_map) fprintf(f, "%zu\t%s\n", m.first, m.second.c_str());
                fprintf(f, "%zu\tend of proof\n", proof.bytes.size());
                fclose(f);
            }
        }
        t_replayed = t_synced + (wall_ms() - t);
        ctx->prof_collect();
    }
    void finish() { sync_results(); replay(); }
};

// ------------------------------------------------------------------------------------------------
// Lays out the node tables of `pk`'s circuit in ONE allocation (plus the NTT scratch): hg_values without contents.
static hg_values* values_alloc(hg_ctx* ctx, const hg_pk* pk, const std::vector<char>* mask = nullptr, bool with_ct0is = true) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    const HCircuit& c = pk->circuit;
    const Params& p = pk->params;
    const size_t nn = c.nodes.size();
    struct ValuesDeleter { void operator()(hg_values* p) const { values_free(p); } };
    std::unique_ptr<hg_values, ValuesDeleter> v(new hg_values());
    static std::atomic<uint64_t> next_serial{1};
    v->serial = next_serial++;
    v->pk_serial = pk->serial;
    v->device = ctx->device;
    v->ctx = ctx;
    v->level.assign(nn, 0);
    for (int id : c.topo) { for (int pr : c.nodes[id].preds) v->level[id] = std::max(v->level[id], v->level[pr] + 1); v->max_level = std::max(v->max_level, v->level[id]); }
    v->d_vals.assign(nn, nullptr);
    v->sizes.assign(nn, 0);
    // layout: FFT nodes of one (level, direction) group are contiguous so that one batched NTT covers the group
    v->order.resize(nn);
    for (size_t i = 0; i < nn; i++) v->order[i] = (int)i;
    const std::vector<int>& level = v->level;
    auto key = [&](int id) { const HNode& n = c.nodes[id]; return std::make_tuple(n.kind == NK_FFT ? 1 : 0, level[id], n.inverse ? 1 : 0, id); };
    std::sort(v->order.begin(), v->order.end(), [&](int a, int b) { return key(a) < key(b); });
    v->ct0is_len = (size_t)p.k * p.SZ();
    if (mask) { v->mask = *mask; v->with_ct0is = with_ct0is; }
    auto in = [&](size_t id) { return v->mask.empty() || v->mask[id]; };
    size_t total = v->with_ct0is ? v->ct0is_len : 0;
    for (size_t id = 0; id < nn; id++) { v->sizes[id] = (size_t)1 << c.nodes[id].log2_out(); if (in(id)) total += v->sizes[id]; }
    u64* base = nullptr;
    hip_check(hipMalloc((void**)&base, std::max<size_t>(total, 1) * 8), "hipMalloc(node values)");
    v->owned.push_back(base);
    size_t off = 0;
    for (int id : v->order) if (in((size_t)id)) { v->d_vals[id] = base + off; off += v->sizes[id]; }   // (a masked FFT group stays contiguous)
    if (v->with_ct0is) v->d_ct0is = base + off;
    size_t max_fft = 0;
    {
        std::map<std::pair<int, int>, size_t> grp_sz;
        for (size_t id = 0; id < nn; id++) if (c.nodes[id].kind == NK_FFT && in(id)) grp_sz[{level[id], (int)c.nodes[id].inverse}] += v->sizes[id];
        for (auto& kv : grp_sz) max_fft = std::max(max_fft, kv.second);
    }
    if (max_fft) { hip_check(hipMalloc((void**)&v->ntt_scratch, max_fft * 8), "hipMalloc(ntt scratch)"); v->owned.push_back(v->ntt_scratch); }
    v->resident_bytes = (total + max_fft) * 8;
    return v.release();
}

static void shard_fill(hg_ctx* ctx, hg_values* v, const hg_values* full, hipStream_t st, bool sync);
static void witness_fill(hg_ctx* ctx, const hg_pk* pk, const Witness& w, hg_values* v, hipStream_t st, bool sync, double* witness_ms, double* upload_ms, u64* pinned = nullptr);
void witness_gen_into(hg_ctx* ctx, const hg_pk* pk, const Witness& w, hg_values* v, double* witness_ms, double* upload_ms) {
    witness_fill(ctx, pk, w, v, ctx->stream, true, witness_ms, upload_ms);
}
// `st`: the stream everything is enqueued on; sync == false: nothing waits (the caller orders later work behind an event on `st`)
// `pinned` (hg_prove_stream): page-locked staging for the whole witness. The caller's arrays are pageable, and an "asynchronous" copy
// from pageable memory is staged by the runtime inside the call, 37 times per witness (0.9 ms of host time at n=32768 k=16, during
// which nothing else is enqueued): the arrays are gathered into `pinned` by all host threads first, then copied by real DMAs.
static void witness_fill(hg_ctx* ctx, const hg_pk* pk, const Witness& w, hg_values* v, hipStream_t st, bool sync, double* witness_ms, double* upload_ms, u64* pinned) {
    // Circuit::evaluate on the device: inputs are uploaded, then the circuit is evaluated level by level
    // (Vanilla nodes: gate-major kernel; FFT nodes: batched NTTs, same level + direction in one batch).
    // Every table keeps its address: a launch graph recorded for `v` proves the new witness as it is (the launch sequence of a
    // prove depends on addresses only).
    if (!v || v->pk_serial != pk->serial) throw Error("witness generation: the values object was laid out for another prover key");
    if (v->device != ctx->device) throw Error("witness generation: the values object lives on another device");
    if (v->ctx != ctx) throw Error("witness generation: the values object was created on another context");
    if (v->shard_rank >= 0) {   // a rank's share: evaluate the cone its tables depend on into the object's own subset tables, copy them over
        if (!v->eval_cone) throw Error("witness generation: a rank's values object without its evaluation cone");
        // (timed around the real completion: the inner call only enqueues - upload, the cone's NTT / gate kernels - and the single
        // synchronisation is shard_fill's. The span is not split: witness_ms = all of it, upload_ms = 0. Without `sync` nothing waits
        // and both are the host's enqueue time.)
        const double t0 = wall_ms();
        witness_fill(ctx, pk, w, v->eval_cone, st, false, nullptr, nullptr, pinned);
        shard_fill(ctx, v, v->eval_cone, st, sync);
        if (witness_ms) *witness_ms = wall_ms() - t0;
        if (upload_ms) *upload_ms = 0;
        return;
    }
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    const HCircuit& c = pk->circuit;
    const Params& p = pk->params;
    if (w.ct0is.size() != v->ct0is_len) throw Error("circuit: ct0is size mismatch");
    double t0 = wall_ms();
    auto dv = [&](int id) { return const_cast<u64*>(v->d_vals[id]); };
    auto in = [&](int id) { return v->mask.empty() || v->mask[id]; };   // (a subset object evaluates its nodes only: their inputs are in it)
    {   // inputs in NodeId order: s, e, k1, ais.., r1is.., r2is (chain_par! sk_encryption_circuit.rs:408)
        const size_t SZ = p.SZ();
        size_t idx = 0;
        struct Copy { u64* dst; const u64* src; size_t len; };
        std::vector<Copy> copies;
        auto put = [&](const u64* src, size_t len) {
            int id = c.input_ids.at(idx++);
            if (len != v->sizes[id]) throw Error("circuit: input size mismatch");
            if (in(id)) copies.push_back({dv(id), src, len});
        };
        put(w.s.data(), SZ); put(w.e.data(), SZ); put(w.k1.data(), SZ);
        for (int i = 0; i < p.k; i++) put(&w.ais[i * SZ], SZ);
        for (int i = 0; i < p.k; i++) put(&w.r1is[i * SZ], SZ);
        put(w.r2is.data(), w.r2is.size());
        if (v->d_ct0is) copies.push_back({const_cast<u64*>(v->d_ct0is), w.ct0is.data(), w.ct0is.size()});
        if (pinned) {   // gather, all threads: pieces of 64 KiB
            std::vector<size_t> off(copies.size() + 1, 0);
            for (size_t q = 0; q < copies.size(); q++) off[q + 1] = off[q] + copies[q].len;
            const size_t piece = 8192, npieces = (off.back() + piece - 1) / piece;
            [[maybe_unused]] const int nt = std::max(1, std::min(hg_omp_threads(), 32));
#pragma omp parallel for schedule(static) num_threads(nt)
            for (long long pc = 0; pc < (long long)npieces; pc++) {
                size_t a = (size_t)pc * piece, b = std::min(off.back(), a + piece);
                size_t q = (size_t)(std::upper_bound(off.begin(), off.end(), a) - off.begin()) - 1;
                while (a < b) {
                    const size_t take = std::min(b, off[q + 1]) - a;
                    memcpy(pinned + a, copies[q].src + (a - off[q]), take * 8);
                    a += take; q++;
                }
            }
            for (size_t q = 0; q < copies.size(); q++) copies[q].src = pinned + off[q];
        }
        for (auto& cp : copies) hip_check(hipMemcpyAsync(cp.dst, cp.src, cp.len * 8, hipMemcpyHostToDevice, st), "upload input");
    }
    if (upload_ms && sync) hip_check(hipStreamSynchronize(st), "upload sync");   // (only to split the two timings)
    double t1 = wall_ms();
    for (int l = 1; l <= v->max_level; l++) {
        for (int inv = 0; inv < 2; inv++) {  // FFT groups
            std::vector<int> grp;
            for (int id : v->order) if (c.nodes[id].kind == NK_FFT && v->level[id] == l && (int)c.nodes[id].inverse == inv && in(id)) grp.push_back(id);
            if (grp.empty()) continue;
            const int L = c.nodes[grp[0]].log2_size;
            const size_t N = (size_t)1 << L;
            for (int id : grp) {
                if (c.nodes[id].log2_size != L) throw Error("circuit: mixed FFT sizes in one level");
                hip_check(hipMemcpyAsync(dv(id), dv(c.nodes[id].preds[0]), N * 8, hipMemcpyDeviceToDevice, st), "copy fft input");
            }
            const u64* W = (inv ? pk->w_inv : pk->w_fwd).at(L);
            dev::ntt_batch(st, dv(grp[0]), L, grp.size(), W, inv ? gl_inv(gl_from_u64(N)) : 1, v->ntt_scratch);
        }
        for (int id : c.topo) {
            const HNode& n = c.nodes[id];
            if (v->level[id] != l || !in(id)) continue;
            if (n.kind == NK_VANILLA) {
                dev::EvalNode e = pk->node_dev[id].fwd;
                for (int i = 0; i < n.arity; i++) e.in[i] = dv(n.preds[i]);
                e.out = dv(id);
                dev::gate_eval(st, e);
            } else if (n.kind == NK_LASSO) {
                hip_check(hipMemsetAsync(dv(id), 0, 8, st), "lasso output");  // LassoNode::evaluate returns [0] (lasso.rs:53-55)
            }
        }
    }
    if (sync) hip_check(hipStreamSynchronize(st), "witness generation sync");
    hip_check(hipGetLastError(), "witness generation");
    double t2 = wall_ms();
    if (upload_ms) *upload_ms = t1 - t0;
    if (witness_ms) *witness_ms = t2 - t1;
}

hg_values* witness_gen(hg_ctx* ctx, const hg_pk* pk, const Witness& w, double* witness_ms, double* upload_ms) {
    hg_values* v = values_alloc(ctx, pk);
    try { witness_gen_into(ctx, pk, w, v, witness_ms, upload_ms); } catch (...) { values_free(v); throw; }
    return v;
}

// ---- a rank's share of the node tables (BASELINE config 4: the witness is NOT replicated) ---------------------------------------------
// Which tables rank `rank` of a `world`-GPU proof reads: the Lasso node's input (every rank holds memories of the Lasso node), the
// inputs of the Vanilla / FFT node reductions it owns, ct0is if it evaluates the output claim.
static void shard_needed(const hg_pk* pk, int rank, int world, std::vector<char>* need, bool* need_ct0is) {
    const HCircuit& c = pk->circuit;
    const ShardPlan sp = shard_plan(pk, rank, world);
    need->assign(c.nodes.size(), 0);
    bool any_mem = false;
    for (int o : sp.gp1_mem_owner) any_mem |= o == rank;
    for (size_t id = 0; id < c.nodes.size(); id++) {
        const HNode& n = c.nodes[id];
        if (n.kind == NK_LASSO && any_mem) (*need)[n.preds[0]] = 1;
        if ((n.kind == NK_VANILLA || n.kind == NK_FFT) && sp.node_owner[id] == rank) for (int p : n.preds) (*need)[p] = 1;
    }
    *need_ct0is = sp.own_out_claim == rank;
}
// copies the needed tables out of a fully evaluated circuit into the compact allocation of `v` (same addresses every time)
static void shard_fill(hg_ctx* ctx, hg_values* v, const hg_values* full, hipStream_t st, bool sync) {
    for (size_t id = 0; id < v->d_vals.size(); id++)
        if (v->d_vals[id]) {
            if (!full->d_vals[id]) throw Error("shard fill: a resident table is not in the evaluated cone");
            hip_check(hipMemcpyAsync(const_cast<u64*>(v->d_vals[id]), full->d_vals[id], v->sizes[id] * 8, hipMemcpyDeviceToDevice, st), "keep node table");
        }
    if (v->d_ct0is) {
        if (!full->d_ct0is) throw Error("shard fill: ct0is is not in the evaluated cone");
        hip_check(hipMemcpyAsync(const_cast<u64*>(v->d_ct0is), full->d_ct0is, v->ct0is_len * 8, hipMemcpyDeviceToDevice, st), "keep ct0is");
    }
    if (sync) hip_check(hipStreamSynchronize(st), "shard fill");
}
hg_values* witness_gen_shard(hg_ctx* ctx, const hg_pk* pk, const Witness& w, int rank, int world, double* witness_ms, double* upload_ms) {
    if (world < 1 || rank < 0 || rank >= world) throw Error("witness_gen_shard: bad rank / world");
    if (world == 1) return witness_gen(ctx, pk, w, witness_ms, upload_ms);
    // Only the CONE of the rank's tables is evaluated: a table the rank reads, and recursively everything it is computed from - the
    // per-modulus chains (a_i -> FFT -> mul -> IFFT -> ... [REF sk_encryption_circuit.rs:122-128, 245-260]) of the moduli whose node
    // reductions it owns, the inputs behind the Lasso node's table - into a subset object the values object keeps (eval_cone), so that
    // a refill (hg_witness_gen_into) allocates nothing. Peak residency of the rank = its tables + the cone (hg_values_info).
    struct ValuesDeleter { void operator()(hg_values* p) const { values_free(p); } };
    std::vector<char> need;
    bool need_ct0is = false;
    shard_needed(pk, rank, world, &need, &need_ct0is);
    const HCircuit& c = pk->circuit;
    std::vector<char> cone = need;
    for (size_t q = c.topo.size(); q-- > 0;) {   // reverse topological order: a node's predecessors come later in this walk
        const int id = c.topo[q];
        if (cone[id]) for (int pr : c.nodes[id].preds) cone[pr] = 1;
    }
    std::unique_ptr<hg_values, ValuesDeleter> sub(values_alloc(ctx, pk, &cone, need_ct0is));
    std::unique_ptr<hg_values, ValuesDeleter> v(new hg_values());
    static std::atomic<uint64_t> next_serial{(uint64_t)1 << 40};   // (disjoint from values_alloc's serials)
    v->serial = next_serial++;
    v->pk_serial = pk->serial; v->device = ctx->device; v->ctx = ctx;
    v->sizes = sub->sizes;
    v->ct0is_len = sub->ct0is_len;
    v->d_vals.assign(sub->d_vals.size(), nullptr);
    v->shard_rank = rank; v->shard_world = world;
    size_t total = need_ct0is ? v->ct0is_len : 0, all = v->ct0is_len;
    for (size_t id = 0; id < need.size(); id++) { all += v->sizes[id]; if (need[id]) total += v->sizes[id]; }
    u64* base = nullptr;
    hip_check(hipMalloc((void**)&base, std::max<size_t>(total, 1) * 8), "hipMalloc(a rank's node tables)");
    v->owned.push_back(base);
    size_t off = 0;
    for (size_t id = 0; id < need.size(); id++) if (need[id]) { v->d_vals[id] = base + off; off += v->sizes[id]; }
    if (need_ct0is) v->d_ct0is = base + off;
    v->resident_bytes = total * 8; v->full_bytes = all * 8;
    v->cone_bytes = sub->resident_bytes;
    v->eval_cone = sub.release();
    witness_fill(ctx, pk, w, v.get(), ctx->stream, true, witness_ms, upload_ms);
    return v.release();
}

// contexts alive in this process: a values object that is freed tells its context to drop the launch graphs recorded for it
static std::mutex g_live_mu;
static std::vector<hg_ctx*> g_live_ctx;
void ctx_register(hg_ctx* ctx, bool alive) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_ctx.erase(std::remove(g_live_ctx.begin(), g_live_ctx.end(), ctx), g_live_ctx.end());
    if (alive) g_live_ctx.push_back(ctx);
}
void prove_cache_forget_values(hg_ctx* ctx, uint64_t values_serial);

void values_free(hg_values* v) {
    if (!v) return;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        for (hg_ctx* c : g_live_ctx) if (c == v->ctx) prove_cache_forget_values(c, v->serial);
    }
    for (void* p : v->owned) (void)hipFree(p);
    if (v->eval_cone) values_free(v->eval_cone);
    delete v;
}

// everything a prove puts on the streams, in protocol order (also what a graph capture records)
static void enqueue_prove(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, Prover* P, int world, bool exchange) {
    const Params& p = pk->params;
    P->stamp("start");
    if (world == 1) P->res_used = v->res_base;
    // (a sharded prove's result-buffer prefix was cleared by the Prover's first launch)
    const bool hinted = ctx->res_hint_serial == pk->serial && ctx->res_hint > 0 && ctx->res_hint <= ctx->res_cap;
    // "eval output" (sk_encryption_circuit.rs:444-448): point, ct0is MLE value
    const int ov = p.ct0is_log2();
    size_t point_off = P->epos();
    for (int i = 0; i < ov; i++) P->squeeze();
    size_t vslot = P->slot(1);
    // (on the main stream: moved behind the counter sorts on the second stream it starts the limb split 30 us earlier and changes
    // nothing at the end of the prove - and the launch graph's stream assignment is touchy about what forks first)
    if (P->mine(P->own_out_claim)) {
        E2* eq = ctx->alloc_n<E2>((size_t)1 << ov);
        P->eq_now(eq, ov, point_off);
        const u64* tabs[8] = {v->d_ct0is};
        dev::dot_eq(ctx->stream, eq, tabs, 1, (size_t)1 << ov, ctx->d_partials, P->d_res() + vslot);
    }
    Cell out_value = cell();
    P->cur_early = world == 1 && Prover::early_replay_on();   // (its slot is written ahead of the fork of the second stream)
    P->push_op([P, out_value, vslot] { *out_value = P->h_res()[vslot]; });
    P->cur_early = false;
    P->gkr(ClaimRef{point_off, ov, out_value});
    if (world > 1) {
        if (hinted && P->res_used > ctx->res_hint) throw Error("sharded prove: the result buffer grew between two proves of one key");
        ctx->res_hint = P->res_used; ctx->res_hint_serial = pk->serial;
    }
    P->stamp("end of the prove");
    if (exchange) comm_allreduce_results(ctx, P->res_used);  // the one collective of a sharded proof, on the stream
}

// ---- walk bookkeeping: which (key, values object, share) has been proven by plain launches how often -----------------------------
static hg_ctx::WalkCount* walk_find(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int share) {
    for (auto& wc : ctx->walk_counts) if (wc.pk_serial == pk->serial && wc.values_serial == v->serial && wc.share == share) return &wc;
    return nullptr;
}
static void walk_note(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int share, size_t arena_bytes, float gpu_ms) {
    hg_ctx::WalkCount* wc = walk_find(ctx, pk, v, share);
    if (!wc) {
        if (ctx->walk_counts.size() >= 64) ctx->walk_counts.erase(ctx->walk_counts.begin());   // (values objects come and go)
        ctx->walk_counts.push_back(hg_ctx::WalkCount{pk->serial, v->serial, share, 0, 0, 0.f});
        wc = &ctx->walk_counts.back();
    }
    wc->walks++;
    wc->arena_bytes = std::max(wc->arena_bytes, arena_bytes);
    wc->gpu_ms = gpu_ms;
}

// enqueue + synchronise this rank's share of one proof; leaves the (partial) result buffer in ctx->h_res
static std::unique_ptr<Prover> prove_begin(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world, double* t_start, float* gpu_ms,
                                           bool exchange = false) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    *t_start = wall_ms();
    std::unique_ptr<Prover> P(new Prover(ctx, pk, rank, world));
    P->d_vals = v->d_vals;
    hipEvent_t ev_a, ev_b;
    hip_check(hipEventCreate(&ev_a), "event"); hip_check(hipEventCreate(&ev_b), "event");
    hip_check(hipEventRecord(ev_a, ctx->stream), "event record");
    enqueue_prove(ctx, pk, v, P.get(), world, exchange);
    hip_check(hipEventRecord(ev_b, ctx->stream), "event record");
    P->sync_results();
    *gpu_ms = 0;
    (void)hipEventElapsedTime(gpu_ms, ev_a, ev_b);
    (void)hipEventDestroy(ev_a); (void)hipEventDestroy(ev_b);
    ctx->last_walk_gpu_ms = *gpu_ms;
    size_t used = 0;
    for (auto& c : ctx->chunks) used += c.high;
    walk_note(ctx, pk, v, rank * 65536 + world, used, *gpu_ms);
    return P;
}

// ---- cached launch graphs (hg_ctx::prove_cache) --------------------------------------------------------------------------------
// One entry per (key, values object, share, stream option). An entry owns the hipGraph, the Prover whose transcript steps are
// replayed on the host after each launch, and a PRIVATE arena the recorded kernels work in - so nothing else that happens on the
// context (witness generation, other keys, kernel-level entry points) touches what a replay reads, and several entries coexist.
// The tables of the values object are referenced by address: hg_witness_gen_into refills them in place and the same graph then
// proves the new witness (no launch parameter, descriptor or grid size depends on table CONTENTS; tests/test_gpu_parity.py
// test_graph_replay_across_witnesses checks it against the oracle).
struct ProveCache {
    std::unique_ptr<Prover> P;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t pk_serial = 0, values_serial = 0;
    bool one_stream = false;
    int rank = 0, world = 1;   // a sharded proof's graph holds this rank's share; the all-reduce follows the replay on the stream
    float walk_gpu_ms = 0;     // GPU time of the walked prove that preceded the capture
    int replays = 0, slow_replays = 0;
    uint64_t last_use = 0;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    char* arena = nullptr;     // private workspace of the recorded launches
    size_t arena_cap = 0;
    int device = 0;
    ~ProveCache() {
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();   // (a replay may still be in flight when a context is torn down after an error)
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (ev_a) (void)hipEventDestroy(ev_a);
        if (ev_b) (void)hipEventDestroy(ev_b);
        if (arena) (void)hipFree(arena);
    }
};
struct ProveCacheSet {
    std::vector<std::shared_ptr<ProveCache>> entries;
    uint64_t clock = 0;
};
static size_t cache_max_entries() {
    static const size_t n = [] { const char* e = getenv("HG_GRAPH_ENTRIES"); long v = e && *e ? atol(e) : 8; return (size_t)std::max(1L, std::min(64L, v)); }();
    return n;
}
void prove_cache_drop(hg_ctx* ctx) {
    delete static_cast<ProveCacheSet*>(ctx->prove_cache);
    ctx->prove_cache = nullptr;
}
static ProveCacheSet* cache_set(hg_ctx* ctx) {
    if (!ctx->prove_cache) ctx->prove_cache = new ProveCacheSet();
    return static_cast<ProveCacheSet*>(ctx->prove_cache);
}
static std::shared_ptr<ProveCache> cache_find(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world) {
    if (!ctx->prove_cache) return nullptr;
    ProveCacheSet* S = static_cast<ProveCacheSet*>(ctx->prove_cache);
    for (auto& e : S->entries)
        if (e->pk_serial == pk->serial && e->values_serial == v->serial && e->one_stream == ctx->one_stream && e->rank == rank && e->world == world) {
            e->last_use = ++S->clock;
            return e;
        }
    return nullptr;
}
static void cache_erase(hg_ctx* ctx, const ProveCache* C) {
    if (!ctx->prove_cache) return;
    auto& es = static_cast<ProveCacheSet*>(ctx->prove_cache)->entries;
    es.erase(std::remove_if(es.begin(), es.end(), [C](const std::shared_ptr<ProveCache>& e) { return e.get() == C; }), es.end());
}
void prove_cache_forget_values(hg_ctx* ctx, uint64_t values_serial) {
    if (!ctx->prove_cache) return;
    auto& es = static_cast<ProveCacheSet*>(ctx->prove_cache)->entries;
    es.erase(std::remove_if(es.begin(), es.end(), [values_serial](const std::shared_ptr<ProveCache>& e) { return e->values_serial == values_serial; }), es.end());
}
static bool graph_allowed(const hg_ctx* ctx) {
    static const bool off = hg_env_on("HG_NO_GRAPH");
    return !off && hg_proof_map_path() == nullptr && ctx->use_graph && ctx->prof_level == 0 && ctx->d_res == ctx->h_res;
}
// launches the cached graph, waits, replays the transcript
// the launch alone (nothing waits): ev_a, the graph, [the collective], ev_b on the prover stream
static void cache_launch(hg_ctx* ctx, ProveCache* C, bool exchange) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    {   // a fresh transcript; the early-replay flag of this graph's prove back to "not yet"
        Prover* P = C->P.get();
        P->proof.bytes.clear();
        P->proof_map.clear();
        P->early_done = false;
        if (P->early_slot != (size_t)-1) __atomic_store_n(&ctx->h_res[P->early_slot].c0, (u64)0, __ATOMIC_RELEASE);
    }
    hip_check(hipEventRecord(C->ev_a, ctx->stream), "event record");
    const bool time_launch = hg_debug("launch");   // (debugging aid: host time of the graph launch call)
    const double tl0 = time_launch ? wall_ms() : 0;
    hip_check(hipGraphLaunch(C->exec, ctx->stream), "hipGraphLaunch");
    if (time_launch) fprintf(stderr, "hipGraphLaunch: %.3f ms on the host\n", wall_ms() - tl0);
    if (exchange) comm_allreduce_results(ctx, C->P->res_used);   // the one collective of a sharded proof, behind the replayed graph
    hip_check(hipEventRecord(C->ev_b, ctx->stream), "event record");
}
// (`launched`: cache_launch has been called; `behind`: and other work may have been enqueued behind it - wait for THIS graph's event)
static ProveResult prove_from_cache(hg_ctx* ctx, ProveCache* C, bool exchange = false, bool replay_now = true, bool launched = false, double t_launch = 0, bool behind = false) {
    ProveResult res;
    const double t0 = launched ? t_launch : wall_ms();
    if (!launched) cache_launch(ctx, C, exchange);
    Prover* P = C->P.get();
    P->st = ctx->stream;
    P->sync_results(behind ? C->ev_b : nullptr);
    float gms = 0;
    (void)hipEventElapsedTime(&gms, C->ev_a, C->ev_b);
    // the first replays are checked against the plain launches this graph recorded: a graph that is clearly slower is given up
    // for this key (prove_through_graph) - the proof it produced is still the proof
    if (!exchange && C->replays < 4 && C->walk_gpu_ms > 0) {
        C->replays++;
        static const float factor = [] { const char* e = getenv("HG_GRAPH_GUARD_FACTOR"); return e && *e ? (float)atof(e) : 1.1f; }();   // (tests force it)
        if (gms > factor * C->walk_gpu_ms + 0.1f) C->slow_replays++;
        if (C->replays == 4 && C->slow_replays >= 3) { ctx->slow_graph_serial = C->pk_serial; ctx->slow_graph_share = C->rank * 65536 + C->world; }
    }
    res.gpu_ms = gms;
    if (!replay_now) { res.prove_ms = wall_ms() - t0; res.enqueue_ms = P->t_enqueued - t0; return res; }   // caller-side exchange first (hg_prove_shard_*)
    P->replay();
    res.prove_ms = wall_ms() - t0;
    res.enqueue_ms = P->t_enqueued - t0;
    res.sync_ms = P->t_synced - P->t_enqueued;
    res.replay_ms = P->t_replayed - P->t_synced;
    res.proof = P->proof.bytes;
    return res;
}
// records the whole enqueue into a graph (no kernel runs during the capture) whose kernels work in a private arena, instantiates it
static std::shared_ptr<ProveCache> prove_capture(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world, const hg_ctx::WalkCount& wc) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->ensure_chain(16384);  // (may allocate and copy synchronously: not allowed once the capture has begun)
    std::shared_ptr<ProveCache> C(new ProveCache());
    C->device = ctx->device;
    C->arena_cap = wc.arena_bytes + ((size_t)1 << 20);
    hip_check(hipMalloc((void**)&C->arena, C->arena_cap), "hipMalloc(private arena of a launch graph)");
    // the context's arena steps aside while the prove is recorded: every buffer the recorded kernels use comes from C->arena
    std::vector<hg_ctx::Chunk> saved_chunks;
    saved_chunks.swap(ctx->chunks);
    const size_t saved_high = ctx->arena_high, saved_total = ctx->arena_total;
    ctx->chunks.push_back(hg_ctx::Chunk{C->arena, C->arena_cap, 0, 0});
    ctx->arena_fixed = true;
    ctx->stage_used = 0;
    auto restore = [&] {
        ctx->arena_fixed = false;
        ctx->chunks.swap(saved_chunks);
        ctx->arena_high = saved_high; ctx->arena_total = saved_total;
        ctx->arena_epoch++;
    };
    bool capturing = false;
    try {
        hip_check(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
        capturing = true;
        C->P.reset(new Prover(ctx, pk, rank, world));
        C->P->d_vals = v->d_vals;
        C->P->defer_uploads = true;   // (descriptor uploads are not graph nodes: Prover::upload)
        if (hg_debug("fail_capture")) throw Error("launch-graph capture failed (forced by HG_DEBUG=fail_capture)");
        enqueue_prove(ctx, pk, v, C->P.get(), world, false);   // (the exchange is not part of the graph: prove_from_cache)
        capturing = false;
        hip_check(hipStreamEndCapture(ctx->stream, &C->graph), "hipStreamEndCapture");
    } catch (...) {
        if (capturing) {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(ctx->stream, &g);
            if (g) (void)hipGraphDestroy(g);
        }
        (void)hipGetLastError();
        restore();
        throw;
    }
    restore();
    for (auto& u : C->P->deferred_uploads)   // once, ahead of the first replay on the same stream; their targets live in the private arena
        hip_check(hipMemcpyAsync(u.dst, u.src, u.bytes, hipMemcpyHostToDevice, ctx->stream), "descriptor upload");
    hip_check(hipStreamSynchronize(ctx->stream), "descriptor uploads");   // (the pinned staging they came from is reused by the next prove)
    C->P->deferred_uploads.clear();
    // (Re-issuing the captured nodes from the library on two real streams - kernel parameters and dependencies read back from the
    // graph - was measured against hipGraphLaunch: 3.50-3.59 ms vs 3.52-3.64 ms of GPU time and 0.53 vs 0.40 ms of host time per
    // prove. No gain, not kept: the serialised look of a replay in a rocprofv3 trace is a profiling artefact.)
    hip_check(hipGraphInstantiate(&C->exec, C->graph, nullptr, nullptr, 0), "hipGraphInstantiate");
    hip_check(hipEventCreate(&C->ev_a), "event"); hip_check(hipEventCreate(&C->ev_b), "event");
    C->pk_serial = pk->serial; C->values_serial = v->serial; C->one_stream = ctx->one_stream;
    C->rank = rank; C->world = world;
    C->walk_gpu_ms = wc.gpu_ms;
    ProveCacheSet* S = cache_set(ctx);
    while (S->entries.size() >= cache_max_entries()) {   // least recently used entry out
        size_t lru = 0;
        for (size_t i = 1; i < S->entries.size(); i++) if (S->entries[i]->last_use < S->entries[lru]->last_use) lru = i;
        S->entries.erase(S->entries.begin() + lru);
    }
    C->last_use = ++S->clock;
    S->entries.push_back(C);
    return C;
}

// The cached-graph path of a (rank of a) prove: replays the graph recorded for exactly this key, values object and share; records one
// on the third prove of that triple; otherwise returns null and the caller walks the protocol. *out is filled when non-null is returned.
static std::shared_ptr<ProveCache> prove_through_graph(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world, bool exchange, ProveResult* out, bool replay_now = true) {
    if (v->pk_serial != pk->serial) throw Error("prove: the resident values were generated for another prover key");
    if (v->shard_rank >= 0 && (v->shard_rank != rank || v->shard_world != world))
        throw Error("prove: these values hold the tables of rank " + std::to_string(v->shard_rank) + " of " + std::to_string(v->shard_world) + " only (hg_witness_gen_shard)");
    if (!graph_allowed(ctx)) return nullptr;
    const int share = rank * 65536 + world;
    if (ctx->slow_graph_serial == pk->serial && ctx->slow_graph_share == share) {   // its graph replayed slower than plain launches
        prove_cache_forget_values(ctx, v->serial);
        return nullptr;
    }
    if (ctx->no_graph_serial == pk->serial && ctx->no_graph_share == share) return nullptr;   // its capture failed: plain launches
    std::shared_ptr<ProveCache> C = cache_find(ctx, pk, v, rank, world);
    if (!C) {
        const hg_ctx::WalkCount* wc = walk_find(ctx, pk, v, share);
        if (!wc || wc->walks < 2) return nullptr;
        try {
            C = prove_capture(ctx, pk, v, rank, world, *wc);
        } catch (const std::exception& e) {
            // no graph for this key and share from now on; this prove and the later ones walk the protocol
            ctx->no_graph_serial = pk->serial; ctx->no_graph_share = share;
            if (hg_debug("shard")) fprintf(stderr, "[hg] launch-graph capture failed, falling back to plain launches: %s\n", e.what());
            return nullptr;
        }
    }
    *out = prove_from_cache(ctx, C.get(), exchange, replay_now);
    return C;
}

ProveResult prove_resident(hg_ctx* ctx, const hg_pk* pk, const hg_values* v) {
    {
        ProveResult cached;
        if (prove_through_graph(ctx, pk, v, 0, 1, false, &cached)) return cached;
    }
    ProveResult res;
    double t3 = 0;
    float gms = 0;
    std::unique_ptr<Prover> P = prove_begin(ctx, pk, v, 0, 1, &t3, &gms);
    P->replay();
    double t4 = wall_ms();
    res.prove_ms = t4 - t3;
    res.gpu_ms = gms;
    res.enqueue_ms = P->t_enqueued - t3;
    res.sync_ms = P->t_synced - P->t_enqueued;
    res.replay_ms = P->t_replayed - P->t_synced;
    res.proof = std::move(P->proof.bytes);
    return res;
}

// BfvEncrypt::prove for a run of witnesses, pipelined over two sets of node tables: while witness i is proven (graph replay on the
// prover streams), witness i+1 is uploaded and evaluated on a third stream into the other set. The host launches the graph FIRST and
// does the (host-side) staging of the next upload while the device proves. Walked proves (the first two per table set) are not
// overlapped. Proofs are what hg_prove gives for each witness.
std::vector<ProveResult> prove_stream(hg_ctx* ctx, const hg_pk* pk, const std::vector<const Witness*>& ws, double* total_ms) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    const double t_all = wall_ms();
    if (!ctx->stream3) {
        hip_check(hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking), "hipStreamCreate");
        for (auto& e : ctx->ev_ready) hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    }
    if (ctx->stream_values[0] && ctx->stream_values_serial != pk->serial)
        for (auto& v : ctx->stream_values) { values_free(v); v = nullptr; }
    if (!ctx->stream_values[0]) {
        for (auto& v : ctx->stream_values) v = values_alloc(ctx, pk);
        ctx->stream_values[1]->res_base = ctx->res_cap / 2;
        ctx->stream_values_serial = pk->serial;
    }
    std::vector<ProveResult> out(ws.size());
    if (ws.empty()) return out;
    hg_values** V = ctx->stream_values;
    {   // pinned staging, one buffer per table set (kept with the context)
        size_t words = V[0]->ct0is_len;
        for (int id : pk->circuit.input_ids) words += V[0]->sizes[id];
        if (ctx->stream_pinned_words < words) {
            for (auto& p : ctx->stream_pinned) { if (p) (void)hipHostFree(p); p = nullptr; }
            for (auto& p : ctx->stream_pinned) hip_check(hipHostMalloc((void**)&p, words * 8, hipHostMallocDefault), "hipHostMalloc(witness staging)");
            ctx->stream_pinned_words = words;
        }
    }
    double wm = 0, um = 0;
    witness_fill(ctx, pk, *ws[0], V[0], ctx->stream3, false, &wm, &um, ctx->stream_pinned[0]);
    hip_check(hipEventRecord(ctx->ev_ready[0], ctx->stream3), "event record");
    // Steady state (both table sets have their launch graph): prove i is launched BEHIND prove i-1 before the host waits for i-1 and
    // replays its transcript - the two write different halves of the result buffer (hg_values::res_base) - so the GPU goes from one
    // prove straight into the next and the replay (0.1 ms) and the launch overlap a running prove; witness i+1 is staged and uploaded
    // into the table set prove i-1 has just released.
    struct Pending { std::shared_ptr<ProveCache> C; size_t idx = 0; double t0 = 0; } pend;
    auto finish = [&] {
        if (!pend.C) return;
        out[pend.idx] = prove_from_cache(ctx, pend.C.get(), false, true, true, pend.t0, true);
        pend.C = nullptr;
    };
    for (size_t i = 0; i < ws.size(); i++) {
        const int cur = (int)(i & 1), nxt = cur ^ 1;
        hip_check(hipStreamWaitEvent(ctx->stream, ctx->ev_ready[cur], 0), "wait for the witness");
        auto fill_next = [&] {
            if (i + 1 >= ws.size()) return;
            // (V[nxt] was last read by prove i-1, which has completed: its results were waited for)
            // (... and so has the DMA out of stream_pinned[nxt], which preceded that table set's evaluation)
            witness_fill(ctx, pk, *ws[i + 1], V[nxt], ctx->stream3, false, &wm, &um, ctx->stream_pinned[nxt]);
            hip_check(hipEventRecord(ctx->ev_ready[nxt], ctx->stream3), "event record");
        };
        std::shared_ptr<ProveCache> C = graph_allowed(ctx) ? cache_find(ctx, pk, V[cur], 0, 1) : nullptr;
        if (C && !(ctx->slow_graph_serial == pk->serial && ctx->slow_graph_share == 1)) {
            const double t0 = wall_ms();
            cache_launch(ctx, C.get(), false);             // (behind prove i-1 on the prover stream, if that one is still pending)
            finish();                                      // prove i-1: wait for its own event, replay - under prove i
            fill_next();                                   // host staging + the third stream's work, under prove i as well
            pend.C = C; pend.idx = i; pend.t0 = t0;
        } else {
            finish();
            fill_next();
            out[i] = prove_resident(ctx, pk, V[cur]);      // walks (and records the graph on the third prove of this table set)
        }
    }
    finish();
    hip_check(hipStreamSynchronize(ctx->stream3), "stream3");
    if (total_ms) *total_ms = wall_ms() - t_all;
    return out;
}

ProveResult prove_sharded(hg_ctx* ctx, const hg_pk* pk, const hg_values* v) {
    if (!ctx->comm) throw Error("hg_prove_sharded: no communicator on this context (hg_comm_init)");
    if (ctx->d_res != ctx->h_res) throw Error("hg_prove_sharded: needs the host-mapped result buffer");
    {
        ProveResult cached;   // this rank's share as a cached launch graph, the all-reduce enqueued behind it
        if (prove_through_graph(ctx, pk, v, ctx->comm_rank, ctx->comm_world, true, &cached)) return cached;
    }
    ProveResult res;
    double t3 = 0;
    float gms = 0;
    std::unique_ptr<Prover> P = prove_begin(ctx, pk, v, ctx->comm_rank, ctx->comm_world, &t3, &gms, true);
    P->replay();
    res.prove_ms = wall_ms() - t3;
    res.gpu_ms = gms;
    res.enqueue_ms = P->t_enqueued - t3;
    res.sync_ms = P->t_synced - P->t_enqueued;
    res.replay_ms = P->t_replayed - P->t_synced;
    res.proof = std::move(P->proof.bytes);
    return res;
}

// sharded single proof: begin (this rank's jobs) -> caller sum-all-reduces ctx->h_res[0 .. n) -> finish
// One sharded prove in flight per context (hg_ctx::pending_shard). It OWNS what finish needs: its own Prover, or a reference to
// the cache entry whose Prover recorded the replayed graph - dropping or evicting the entry in between cannot free it.
struct PendingShard { std::unique_ptr<Prover> own; std::shared_ptr<ProveCache> cached; Prover* P = nullptr; double t_start = 0; float gpu_ms = 0; };
void pending_shard_drop(hg_ctx* ctx) {
    delete static_cast<PendingShard*>(ctx->pending_shard);
    ctx->pending_shard = nullptr;
}

size_t prove_shard_begin(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) throw Error("prove_shard_begin: bad rank/world");
    pending_shard_drop(ctx);   // (a begin without its finish is superseded)
    std::unique_ptr<PendingShard> ps(new PendingShard());
    ProveResult cached;
    ps->t_start = wall_ms();
    if ((ps->cached = prove_through_graph(ctx, pk, v, rank, world, false, &cached, false))) {   // this rank's share replayed from its launch graph
        ps->P = ps->cached->P.get();
        ps->gpu_ms = (float)cached.gpu_ms;
    } else {
        ps->own = prove_begin(ctx, pk, v, rank, world, &ps->t_start, &ps->gpu_ms);
        ps->P = ps->own.get();
    }
    size_t n = ps->P->res_used;
    if (hg_debug("shard")) fprintf(stderr, "[hg] shard rank %d/%d: gpu %.3f ms, enqueue %.3f ms\n", rank, world, ps->gpu_ms, ps->P->t_enqueued - ps->t_start);
    ctx->pending_shard = ps.release();
    return n;
}
// installs the modular sum of the ranks' partial result buffers (`world` buffers of n_u64 lanes each, rank-major)
void prove_shard_combine(hg_ctx* ctx, const u64* gathered, int world, size_t n_u64) {
    if (2 * ctx->res_cap < n_u64) throw Error("prove_shard_combine: buffer larger than the result buffer");
    if (!ctx->pending_shard) throw Error("prove_shard_combine: no sharded prove in flight on this context");
    shard_combine_host(gathered, world, n_u64, reinterpret_cast<u64*>(ctx->h_res));
}
// lane-wise sum mod p of `world` buffers of canonical lanes (host only: what the caller-side exchange of a sharded proof computes)
void shard_combine_host(const u64* gathered, int world, size_t n_u64, u64* dst) {
    for (size_t i = 0; i < n_u64; i++) {
        u64 acc = 0;
        for (int r = 0; r < world; r++) {
            u64 v = gathered[(size_t)r * n_u64 + i];
            if (v >= GL_P) throw Error("prove_shard_combine: non-canonical lane");
            acc = gl_add(acc, v);
        }
        dst[i] = acc;
    }
}

ProveResult prove_shard_finish(hg_ctx* ctx) {
    if (!ctx->pending_shard) throw Error("prove_shard_finish: no sharded prove in flight on this context (or it was superseded by another prove)");
    std::unique_ptr<PendingShard> ps(static_cast<PendingShard*>(ctx->pending_shard));
    ctx->pending_shard = nullptr;
    ProveResult res;
    const double tf0 = wall_ms();
    ps->P->replay();
    if (hg_debug("shard")) fprintf(stderr, "[hg] shard finish: replay call %.3f ms\n", wall_ms() - tf0);
    res.prove_ms = wall_ms() - ps->t_start;
    res.gpu_ms = ps->gpu_ms;
    res.enqueue_ms = ps->P->t_enqueued - ps->t_start;
    res.sync_ms = ps->P->t_synced - ps->P->t_enqueued;
    res.replay_ms = ps->P->t_replayed - ps->P->t_synced;
    res.proof = std::move(ps->P->proof.bytes);
    return res;
}

std::vector<uint8_t> prove_lasso_node(hg_ctx* ctx, const hg_pk* pk, const u64* lasso_in_host, size_t chain_skip, std::vector<E2>* claim_out) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    Prover P(ctx, pk);
    ctx->ensure_chain(chain_skip + 4096);
    P.ch.pos = 2 * chain_skip;  // the node is entered with `chain_skip` E challenges already squeezed by the caller
    const size_t N = (size_t)1 << pk->lasso.nu;
    u64* d_in = ctx->alloc_n<u64>(N);
    hip_check(hipMemcpyAsync(d_in, lasso_in_host, N * 8, hipMemcpyHostToDevice, ctx->stream), "upload lasso input");
    ClaimRef cr = P.lasso_node(d_in);
    P.finish();
    if (claim_out) {
        const u64* chain = challenge_chain(2 * (cr.point_off + cr.len));
        claim_out->clear();
        for (int i = 0; i < cr.len; i++) claim_out->push_back(e2(chain[2 * (cr.point_off + i)], chain[2 * (cr.point_off + i) + 1]));
        claim_out->push_back(*cr.value);
    }
    return std::move(P.proof.bytes);
}

void sumcheck_on_tables(hg_ctx* ctx, SumcheckIO& io) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    Prover P(ctx, nullptr);
    const size_t N = (size_t)1 << io.nv;
    const int ntab = (int)io.tables.size();
    P.ch.pos = 2 * io.chain_skip;
    Cell claim = cell(io.claim), out = cell();
    ScHandle h;
    size_t evals = P.slot(ntab);
    if (io.kind == 2) {
        std::vector<const u64*> a;
        std::vector<const E2*> b;
        std::vector<E2*> fa, fb;
        for (int i = 0; i < ntab; i += 2) {
            if (!io.is_base[i] || io.is_base[i + 1]) throw Error("hg_sumcheck: prodsum expects (base, ext) table pairs");
            u64* da = ctx->alloc_n<u64>(N);
            E2* db = ctx->alloc_n<E2>(N);
            hip_check(hipMemcpyAsync(da, io.tables[i], N * 8, hipMemcpyHostToDevice, ctx->stream), "upload");
            hip_check(hipMemcpyAsync(db, io.tables[i + 1], N * 16, hipMemcpyHostToDevice, ctx->stream), "upload");
            a.push_back(da); b.push_back(db);
            fa.push_back(ctx->d_res + evals + i); fb.push_back(ctx->d_res + evals + i + 1);
        }
        h = P.sc_prodsum(a, b, (int)io.nv, fa, fb);
        P.flush_prodsum();
    } else {
        bool base = io.is_base[0] != 0;
        for (int i = 0; i < ntab; i++) if ((io.is_base[i] != 0) != base) throw Error("hg_sumcheck: mixed table fields");
        size_t el = base ? 8 : 16;
        char* d = (char*)ctx->alloc((size_t)ntab * N * el);
        for (int i = 0; i < ntab; i++) hip_check(hipMemcpyAsync(d + (size_t)i * N * el, io.tables[i], N * el, hipMemcpyHostToDevice, ctx->stream), "upload");
        dev::Powers pw;
        memset(&pw, 0, sizeof(pw));
        for (size_t i = 0; i < io.pw.size() && i < (size_t)dev::PW_MAX; i++) pw.v[i] = io.pw[i];
        h = P.sc_stride(io.kind == 1 ? dev::SC_GRANDPROD : dev::SC_COLLATION, d, base, N, ntab, (int)io.nv, pw, ctx->d_res + evals);
        P.flush_stride();
    }
    int deg = io.kind == 1 ? 3 : 2;
    P.defer_sumcheck(h, deg, claim, out);
    if (io.kind == 0) {  // collation kernels leave the final evaluation of table i multiplied by M^i (pw[i])
        std::vector<E2> w = io.pw;
        P.push_op([ctx, evals, ntab, w] {
            for (int i = 0; i < ntab && i < (int)w.size(); i++) ctx->h_res[evals + i] = e2_mul(ctx->h_res[evals + i], e2_inv(w[i]));
        });
    }
    if (io.kind == 1) {
        dev::Powers pw;
        memset(&pw, 0, sizeof(pw));
        for (size_t i = 0; i < io.pw.size() && i < (size_t)dev::PW_MAX; i++) pw.v[i] = io.pw[i];
        P.defer_gp_unscale(evals, ntab / 2, pw);
    }
    P.finish();
    io.point = h.rs;
    io.evals.assign(ctx->h_res + evals, ctx->h_res + evals + ntab);
    io.sums.assign(ctx->h_res + h.sums_slot, ctx->h_res + h.sums_slot + (size_t)h.nvars * h.nv);
    // decode the coefficient messages back from the stream
    io.msgs.clear();
    const std::vector<uint8_t>& b = P.proof.bytes;
    for (size_t o = 0; o + 16 <= b.size(); o += 16) {
        u64 c0 = 0, c1 = 0;
        for (int i = 0; i < 8; i++) { c0 = (c0 << 8) | b[o + i]; c1 = (c1 << 8) | b[o + 8 + i]; }
        io.msgs.push_back(e2(c0, c1));
    }
}

// prove_grand_product on caller tables (kernel-level parity entry point, hg_grand_product)
std::vector<uint8_t> grand_product_on_tables(hg_ctx* ctx, size_t nb, size_t len, const u64* const* tables, size_t chain_skip, std::vector<E2>* claims_out,
                                             std::vector<E2>* point_out) {
    if (nb == 0 || nb > (size_t)dev::PW_MAX || len < 2 || (len & (len - 1))) throw Error("hg_grand_product: need 1..64 tables of a power-of-two length >= 2");
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    Prover P(ctx, nullptr);
    int nv = 0;
    while (((size_t)1 << nv) < len) nv++;
    ctx->ensure_chain(chain_skip + 4096);
    P.ch.pos = 2 * chain_skip;
    u64* H = ctx->alloc_n<u64>(nb * len);
    for (size_t b = 0; b < nb; b++) {
        for (size_t i = 0; i < len; i++) if (tables[b][i] >= GL_P) throw Error("hg_grand_product: non-canonical table entry");
        hip_check(hipMemcpyAsync(H + b * len, tables[b], len * 8, hipMemcpyHostToDevice, ctx->stream), "upload table");
    }
    Prover::GpOut g = P.grand_product(H, len, (int)nb, std::vector<int>(nv, 0));
    P.flush_stride();
    P.finish();
    if (claims_out) *claims_out = *g.claims;
    if (point_out) {
        const u64* chain = challenge_chain(2 * (g.point_off + nv));
        point_out->clear();
        for (int i = 0; i < nv; i++) point_out->push_back(e2(chain[2 * (g.point_off + i)], chain[2 * (g.point_off + i) + 1]));
    }
    return std::move(P.proof.bytes);
}

// fix_var on the lowest variable of one table (hg_fold)
__global__ void k_fold_table(const void* __restrict__ in, int is_base, size_t half, E2 r, E2* __restrict__ out) {
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < half; j += (size_t)gridDim.x * blockDim.x) {
        if (is_base) {
            const u64* t = static_cast<const u64*>(in);
            const u64 x = t[2 * j], y = t[2 * j + 1];
            out[j] = e2_add_f(e2_mul_f(r, gl_sub(y, x)), x);
        } else {
            const E2* t = static_cast<const E2*>(in);
            const E2 x = t[2 * j], y = t[2 * j + 1];
            out[j] = e2_add(x, e2_mul(r, e2_sub(y, x)));
        }
    }
}
void fold_device(hg_ctx* ctx, const u64* table_host, size_t nv, bool is_base, E2 r, E2* out_host) {
    if (nv < 1 || nv > 30) throw Error("hg_fold: table size out of range");
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    const size_t N = (size_t)1 << nv, half = N >> 1, el = is_base ? 8 : 16;
    void* d = ctx->alloc(N * el);
    E2* o = ctx->alloc_n<E2>(half);
    hip_check(hipMemcpyAsync(d, table_host, N * el, hipMemcpyHostToDevice, ctx->stream), "upload");
    k_fold_table<<<(unsigned)std::min<size_t>((half + 255) / 256, 4096), 256, 0, ctx->stream>>>(d, is_base ? 1 : 0, half, r, o);
    hip_check(hipMemcpyAsync(out_host, o, half * sizeof(E2), hipMemcpyDeviceToHost, ctx->stream), "download");
    hip_check(hipStreamSynchronize(ctx->stream), "fold sync");
}

E2 mle_eval_device(hg_ctx* ctx, const u64* table_host, size_t nv, const E2* point_host) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    Prover P(ctx, nullptr);
    const size_t N = (size_t)1 << nv;
    u64* d = ctx->alloc_n<u64>(N);
    E2* pt = ctx->alloc_n<E2>(nv ? nv : 1);
    E2* eq = ctx->alloc_n<E2>(N);
    hip_check(hipMemcpyAsync(d, table_host, N * 8, hipMemcpyHostToDevice, ctx->stream), "upload");
    if (nv) hip_check(hipMemcpyAsync(pt, point_host, nv * 16, hipMemcpyHostToDevice, ctx->stream), "upload");
    P.eq_now(eq, (int)nv, 0, pt);
    const u64* tabs[8] = {d};
    size_t s = P.slot(1);
    dev::dot_eq(ctx->stream, eq, tabs, 1, N, ctx->d_partials, P.d_res() + s);
    P.finish();
    return ctx->h_res[s];
}

void ntt_device(hg_ctx* ctx, const u64* in_host, int log2n, bool inverse, size_t batch, u64* out_host) {
    hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->arena_reset();
    const size_t N = (size_t)1 << log2n;
    u64* d = ctx->alloc_n<u64>(N * batch);
    u64* scratch = ctx->alloc_n<u64>(N * batch);
    u64* W = ctx->alloc_n<u64>(N);
    u64 w = root_of_unity(log2n);
    if (inverse) w = gl_inv(w);
    hip_check(hipMemcpyAsync(d, in_host, N * batch * 8, hipMemcpyHostToDevice, ctx->stream), "upload");
    dev::powers_table(ctx->stream, W, w, N);
    dev::ntt_batch(ctx->stream, d, log2n, batch, W, inverse ? gl_inv(gl_from_u64(N)) : 1, scratch);
    hip_check(hipMemcpyAsync(out_host, d, N * batch * 8, hipMemcpyDeviceToHost, ctx->stream), "download");
    hip_check(hipStreamSynchronize(ctx->stream), "ntt sync");
}

}  // namespace hg
