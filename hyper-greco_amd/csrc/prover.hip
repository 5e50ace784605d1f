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
    if (d_partials4) (void)hipFree(d_partials4);
    if (stream_sum) { (void)hipStreamSynchronize(stream_sum); (void)hipStreamDestroy(stream_sum); }
    for (auto& e : ev_sum) if (e) (void)hipEventDestroy(e);
    if (stream_col) { (void)hipStreamSynchronize(stream_col); (void)hipStreamDestroy(stream_col); }
    if (ev_col) (void)hipEventDestroy(ev_col);
    if (comm) { try { hg::comm_destroy(this); } catch (...) {} }
    if (d_xchg) (void)hipFree(d_xchg);
    if (stream2) { (void)hipStreamSynchronize(stream2); (void)hipStreamDestroy(stream2); }
    for (auto& e : bn_ev) if (e) (void)hipEventDestroy(e);
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
    std::shared_ptr<E2> scale_cell;   // ... or by a value the replay knows when it gets there (Libra phase 2 run beside phase 1: u = in(r_x))
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
    size_t res_end = 0;   // slots are handed out below this index (0: the whole result buffer); hg_prove_stream's first table set owns the lower half only
    int cls_gp_sums, cls_gp_hash, cls_gp_base, cls_gp_ext, cls_gp_ext2, cls_col_ext2, cls_col_base, cls_col_ext, cls_ps, cls_ps2, cls_reduce, cls_aux, cls_tree, cls_hash, cls_gather, cls_tail, cls_ps_tail;

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
        cls_gp_sums = ctx->prof_class("sc_round_sums<grand_product,ext>", false);
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
        cs.p[3] = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ctx->d_partials4) + dev::PARTIALS_E2 * sizeof(E2));   // (the split rounds' sums stream)
        cs.n[3] = dev::PARTIALS_TICKETS;
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
        // checked where the slot is handed out: nothing is enqueued that would write into the other table set's half
        if (res_used + n > (res_end ? res_end : ctx->res_cap)) throw Error(res_end ? "hg_prove_stream: a prove's result slots do not fit half of the result buffer" : "result buffer exhausted");
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

#include "prover_sumcheck.inc"
#include "prover_lasso.inc"
#include "prover_nodes.inc"

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
        // (while it waits the host keeps asking for the result buffer's lines, a few per query, round and round: what the device has
        // written for good by then - the sums of the launched rounds - is in the cache when the replay reads it; a line the device
        // writes later is simply invalidated again)
        const char* warm_p = reinterpret_cast<const char*>(ctx->h_res);
        const size_t warm_n = res_used * sizeof(E2) / 64;
        size_t warm = 0;
        for (;;) {
            hipError_t q = hipEventQuery(done);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) hip_check(q, "prove: event query");
            if (warm_n) for (int k = 0; k < 8; k++) { __builtin_prefetch(warm_p + (warm % warm_n) * 64, 0, 3); warm++; }
            if (early_ready()) run_early();   // (the node reductions are done, the Lasso node's last launches are not)
            if (wall_ms() - t_spin > 2000.0) { hip_check(hipEventSynchronize(done), "prove: event sync"); break; }
        }
        hip_check(hipGetLastError(), "prove: kernel launch");
        t_synced = wall_ms();
        print_stamps();
    }
    // The transcript steps read ~9000 result values the device has just written: lines the host has never seen, one dependent miss
    // to DRAM each when the steps touch them one by one (130-150 us for the 277 steps behind the synchronisation). Asked for all
    // at once they arrive while the first steps run.
    void prefetch_results() const {
        const char* p = reinterpret_cast<const char*>(ctx->h_res);
        const size_t bytes = res_used * sizeof(E2);
        for (size_t o = 0; o < bytes; o += 64) __builtin_prefetch(p + o, 0, 3);
    }
    void replay() {
        double t = wall_ms();
        prefetch_results();
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
// page-locked staging for whole witnesses of `pk` (kept with the context): buffers 0 .. count-1 hold one witness each
static void witness_staging(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int count) {
    size_t words = v->ct0is_len;
    for (int id : pk->circuit.input_ids) words += v->sizes[id];
    if (ctx->stream_pinned_words < words) {
        for (auto& p : ctx->stream_pinned) { if (p) (void)hipHostFree(p); p = nullptr; }
        ctx->stream_pinned_words = words;
    }
    for (int q = 0; q < count; q++)
        if (!ctx->stream_pinned[q]) hip_check(hipHostMalloc((void**)&ctx->stream_pinned[q], ctx->stream_pinned_words * 8, hipHostMallocDefault), "hipHostMalloc(witness staging)");
}
// hg_prove's refill of the context-owned tables: the caller's (pageable) arrays go through the page-locked staging buffer - gathered by
// all host threads, then real DMAs - instead of 37 staged "asynchronous" copies (0.9 ms of blocked host time at n=32768 k=16)
void witness_gen_into_staged(hg_ctx* ctx, const hg_pk* pk, const Witness& w, hg_values* v, double* witness_ms, double* upload_ms) {
    witness_staging(ctx, pk, v, 1);
    witness_fill(ctx, pk, w, v, ctx->stream, true, witness_ms, upload_ms, ctx->stream_pinned[0]);
}
struct ProveCache;
static std::shared_ptr<ProveCache> cache_find(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world);
// The cold path of a drop-in caller (BfvEncrypt::setup, then ONE prove per witness: bfv-gkr/src/test.rs:31-44) made warm at set-up time:
// the context-owned node tables and the witness staging are allocated, a zero witness is proven until its launch graph is recorded
// (two walks and the capture), so the caller's FIRST hg_prove refills the tables and replays. Returns the milliseconds it took.
double prove_warmup(hg_ctx* ctx, const hg_pk* pk) {
    const double t0 = wall_ms();
    const Params& p = pk->params;
    Witness z;
    const size_t SZ = p.SZ();
    z.s.assign(SZ, 0); z.e.assign(SZ, 0); z.k1.assign(SZ, 0);
    z.ais.assign((size_t)p.k * SZ, 0); z.r1is.assign((size_t)p.k * SZ, 0);
    z.r2is.assign((size_t)p.k * p.PZ(), 0);
    z.ct0is.assign((size_t)p.k * SZ, 0);
    if (ctx->scratch_values && ctx->scratch_serial != pk->serial) { values_free(ctx->scratch_values); ctx->scratch_values = nullptr; }
    if (!ctx->scratch_values) { ctx->scratch_values = witness_gen(ctx, pk, z, nullptr, nullptr); ctx->scratch_serial = pk->serial; }
    witness_gen_into_staged(ctx, pk, z, ctx->scratch_values, nullptr, nullptr);   // (the refill path itself once: host thread pool, staging pages)
    for (int i = 0; i < 4 && !cache_find(ctx, pk, ctx->scratch_values, 0, 1); i++) (void)prove_resident(ctx, pk, ctx->scratch_values, true);
    return wall_ms() - t0;
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
        if (pinned) {   // gather by all host threads (pieces of 64 KiB) in a few groups of arrays: the DMAs of group g run while group g + 1 is gathered
            std::vector<size_t> off(copies.size() + 1, 0);
            for (size_t q = 0; q < copies.size(); q++) off[q + 1] = off[q] + copies[q].len;
            [[maybe_unused]] const int nt = std::max(1, std::min(hg_omp_threads(), 32));
            const size_t group_words = std::max<size_t>(off.back() / 6, (size_t)1 << 17);   // (a parallel region per group: ~40 us each on 32 threads)
            for (size_t q0 = 0; q0 < copies.size();) {
                size_t q1 = q0 + 1;
                while (q1 < copies.size() && off[q1] - off[q0] < group_words) q1++;
                const size_t lo = off[q0], hi = off[q1];
                const size_t piece = 8192, npieces = (hi - lo + piece - 1) / piece;
#pragma omp parallel for schedule(static) num_threads(nt)
                for (long long pc = 0; pc < (long long)npieces; pc++) {
                    size_t a = lo + (size_t)pc * piece, b = std::min(hi, a + piece);
                    size_t q = (size_t)(std::upper_bound(off.begin(), off.end(), a) - off.begin()) - 1;
                    while (a < b) {
                        const size_t take = std::min(b, off[q + 1]) - a;
                        memcpy(pinned + a, copies[q].src + (a - off[q]), take * 8);
                        a += take; q++;
                    }
                }
                for (size_t q = q0; q < q1; q++) hip_check(hipMemcpyAsync(copies[q].dst, pinned + off[q], copies[q].len * 8, hipMemcpyHostToDevice, st), "upload input");
                q0 = q1;
            }
        } else
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
    P->res_end = world == 1 ? v->res_limit : 0;
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
static ProveResult prove_from_cache(hg_ctx* ctx, ProveCache* C, bool exchange = false, bool replay_now = true, bool launched = false, double t_launch = 0, bool behind = false, bool borrow = false) {
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
    if (borrow) res.proof_ref = &P->proof.bytes;
    else res.proof = P->proof.bytes;
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
static std::shared_ptr<ProveCache> prove_through_graph(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world, bool exchange, ProveResult* out, bool replay_now = true, bool borrow = false) {
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
    *out = prove_from_cache(ctx, C.get(), exchange, replay_now, false, 0, false, borrow);
    return C;
}

ProveResult prove_resident(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, bool borrow) {
    {
        ProveResult cached;
        if (prove_through_graph(ctx, pk, v, 0, 1, false, &cached, true, borrow)) return cached;
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
        ctx->stream_values[0]->res_limit = ctx->res_cap / 2;
        ctx->stream_values[1]->res_base = ctx->res_cap / 2;
        ctx->stream_values_serial = pk->serial;
    }
    std::vector<ProveResult> out(ws.size());
    if (ws.empty()) return out;
    hg_values** V = ctx->stream_values;
    witness_staging(ctx, pk, V[0], 2);   // pinned staging, one buffer per table set (kept with the context)
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
