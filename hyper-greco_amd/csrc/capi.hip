// C ABI (include/hg.h). No exception crosses the boundary: every entry point catches, stores the
// message for hg_last_error() and returns a negative status.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstring>
#include <memory>
#include <string>
#include <tuple>
#include "prover.hpp"

// BN254 slice (bn254.hip)
namespace hg { namespace bn {
void sumcheck_bn254(hg_ctx* ctx, int kind, size_t nv, size_t ntab, const u64* const* tables, const u64* pw4, size_t npw, const u64* claim4,
                    size_t chain_skip, u64* msgs, u64* point, u64* evals, u64* sums);
void field_op_bn254(hg_ctx* ctx, int op, size_t n, const u64* a, const u64* b, u64* out);
void challenges_bn254_raw(size_t n, uint64_t* out4);
void mle_eval_bn254(hg_ctx* ctx, const u64* table4, size_t nv, const u64* point4, u64* out4);
void ntt_bn254(hg_ctx* ctx, const u64* in4, int log2n, bool inverse, size_t batch, u64* out4);
void grand_product_bn254(hg_ctx* ctx, size_t nb, size_t len, const u64* const* tables, size_t chain_skip, std::vector<uint8_t>& proof,
                         u64* claims_out, u64* point_out);
void lasso_prove_bn254(hg_ctx* ctx, const hg_pk* pk, const u64* in4, size_t chain_skip, std::vector<uint8_t>& proof, u64* claim_out);
void circuit_eval_bn254(hg_ctx* ctx, const hg_pk* pk, const Witness& w, int which, std::vector<u64>& out4);
void prove_bn254(hg_ctx* ctx, const hg_pk* pk, const Witness& w, std::vector<uint8_t>& proof, double* ms);
} }
using namespace hg;

static thread_local std::string g_last_error;

#define HG_TRY try {
#define HG_CATCH(ret)                                                           \
    }                                                                           \
    catch (const std::exception& ex) { g_last_error = ex.what(); return ret; } \
    catch (...) { g_last_error = "unknown error"; return ret; }

// (called from the worker threads of hg_setup's node loop: the current device is per thread, the key's list of allocations is shared)
template <typename T>
static const T* upload_vec(hg_pk* pk, const std::vector<T>& v) {
    if (!pk->ctx) return nullptr;  // host-only key (hg_setup(NULL, ..)): wiring without a device copy
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    hip_check(hipSetDevice(pk->ctx->device), "hipSetDevice");
    T* d = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    hip_check(hipMalloc((void**)&d, bytes), "hipMalloc(prover key)");
    if (!v.empty()) hip_check(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "upload prover key");
    pk->owned.push_back(d);
    return d;
}

// counting-sort a term list into a CSR keyed by position `key(t)` in [0, S)
template <typename Term, typename KeyFn, typename EmitFn>
static void build_csr(const std::vector<const Term*>& terms, size_t S, KeyFn key, std::vector<u32>& ptr, EmitFn emit) {
    ptr.assign(S + 1, 0);
    for (const Term* t : terms) ptr[key(*t) + 1]++;
    for (size_t i = 0; i < S; i++) ptr[i + 1] += ptr[i];
    std::vector<u32> fill(ptr.begin(), ptr.end() - 1);
    for (const Term* t : terms) emit(*t, fill[key(*t)]++);
}

// A witness handle must have been built for the prover key's parameter set: the provers and verifiers index its
// tables with sizes taken from the key (k * 2^L, k * 2^P ...), so a mismatch would read out of bounds.
static void check_witness(const hg_pk* pk, const hg_witness* w, const char* who) {
    const hg_params& a = pk->params.raw;
    const hg_params& b = w->params;
    if (a.n != b.n || a.k != b.k) throw Error(std::string(who) + ": witness was built for n=" + std::to_string(b.n) + " k=" + std::to_string(b.k) + ", the prover key for n=" + std::to_string(a.n) + " k=" + std::to_string(a.k));
    const size_t SZ = pk->params.SZ(), PZ = pk->params.PZ(), k = (size_t)pk->params.k;
    const Witness& v = w->w;
    if (v.s.size() != SZ || v.e.size() != SZ || v.k1.size() != SZ || v.ais.size() != k * SZ || v.r1is.size() != k * SZ || v.r2is.size() != k * PZ ||
        v.ct0is.size() != k * SZ)
        throw Error(std::string(who) + ": witness table sizes do not match the parameter set");
}

extern "C" {

const char* hg_last_error(void) { return g_last_error.c_str(); }

int hg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

hg_ctx* hg_create(int device_id) {
    HG_TRY
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        throw Error("no HIP device available: the prover has no CPU fallback (the CPU restatement lives in oracle/ and is test-only)");
    if (device_id < 0 || device_id >= n) throw Error("device id out of range");
    hip_check(hipSetDevice(device_id), "hipSetDevice");
    hg_ctx* c = new hg_ctx();
    c->device = device_id;
    // Plain streams. Stream priorities (main high, second low) measured no better on the streams themselves (3.00-3.02 ms either
    // way) and much worse under the cached launch graph: the second and every later graph instantiated on a context replays its side
    // branch on a stream of NORMAL priority, which competes with the high-priority main branch instead of hiding under it - 5.0 ms
    // per prove instead of 3.05 (scripts/ub/repro_graph_slow.py). Removed in round 5.
    hip_check(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipStreamCreateWithFlags(&c->stream_col, hipStreamNonBlocking), "hipStreamCreate");
    hip_check(hipEventCreateWithFlags(&c->ev_col, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming), "hipEventCreate");
    for (auto& e : c->ev_aux) hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    c->prof_stream = c->stream;
    c->res_cap = (size_t)1 << 17;
    hip_check(hipHostMalloc((void**)&c->h_res, c->res_cap * sizeof(E2), hipHostMallocDefault), "hipHostMalloc(results)");
    // The kernels write the (≈160 KB of) result slots straight into the pinned host buffer: no device-to-host copy at the end of
    // a prove (measured -0.05 ms at n=32768 k=16).
    c->d_res = c->h_res;
    hip_check(hipMalloc((void**)&c->d_partials, dev::PARTIALS_BYTES), "hipMalloc(partials)");
    hip_check(hipMalloc((void**)&c->d_partials2, dev::PARTIALS_BYTES), "hipMalloc(partials2)");
    hip_check(hipMemset(c->d_partials, 0, dev::PARTIALS_BYTES), "hipMemset(partials)");
    hip_check(hipMemset(c->d_partials2, 0, dev::PARTIALS_BYTES), "hipMemset(partials2)");
    hip_check(hipMalloc((void**)&c->d_partials3, dev::PARTIALS_BYTES), "hipMalloc(partials3)");
    hip_check(hipMemset(c->d_partials3, 0, dev::PARTIALS_BYTES), "hipMemset(partials3)");
    hip_check(hipStreamCreateWithFlags(&c->stream_sum, hipStreamNonBlocking), "hipStreamCreate");
    for (auto& e : c->ev_sum) hip_check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    hip_check(hipMalloc((void**)&c->d_partials4, dev::PARTIALS_BYTES), "hipMalloc(partials4)");
    hip_check(hipMemset(c->d_partials4, 0, dev::PARTIALS_BYTES), "hipMemset(partials4)");
    c->stage_cap = (size_t)4 << 20;
    hip_check(hipHostMalloc((void**)&c->h_stage, c->stage_cap, hipHostMallocDefault), "hipHostMalloc(staging)");
    c->ensure_chain(16384);
    ctx_register(c, true);
    return c;
    HG_CATCH(nullptr)
}

void hg_destroy(hg_ctx* ctx) { delete ctx; }

int hg_set_option(hg_ctx* ctx, const char* name, int64_t value) {
    HG_TRY
    if (!ctx || !name) throw Error("hg_set_option: null argument");
    const std::string n(name);
    if (n == "one_stream") { if (ctx->one_stream != (value != 0)) ctx->walk_counts.clear(); ctx->one_stream = value != 0; }
    else if (n == "graph") { ctx->use_graph = value != 0; if (!ctx->use_graph) prove_cache_drop(ctx); }
    else throw Error("hg_set_option: unknown option " + n);
    return 0;
    HG_CATCH(-1)
}

int hg_params_builtin(uint32_t n, uint32_t k, hg_params* out) {
    HG_TRY
    if (!params_builtin(n, k, out)) throw Error("no built-in parameter set for this (n, k)");
    return 0;
    HG_CATCH(-1)
}

static double now_ms_capi() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int hg_setup(hg_ctx* ctx, const hg_params* params, hg_pk** out) {
    HG_TRY
    if (!params || !out) throw Error("hg_setup: null argument");
    if (ctx) hip_check(hipSetDevice(ctx->device), "hipSetDevice");
    std::unique_ptr<hg_pk> pk(new hg_pk(*params));
    static std::atomic<uint64_t> next_serial{1};
    pk->serial = next_serial++;
    pk->ctx = ctx;
    const bool times = hg_times("setup");
    const double ts0 = now_ms_capi();
    auto lap = [&](const char* what) { if (times) fprintf(stderr, "[hg setup] %8.2f ms  %s\n", now_ms_capi() - ts0, what); };
    pk->lasso = lasso_preprocess(pk->params);
    lap("lasso_preprocess");
    pk->circuit = build_circuit(pk->params, pk->lasso);
    lap("build_circuit");
    const LassoPlan& lp = pk->lasso;
    if (lp.alpha > 32 || lp.lookups.size() > 32) throw Error("lasso: more than 32 memories / lookup types");
    dev::LassoDev& L = pk->lasso_dev;
    memset(&L, 0, sizeof(L));
    L.nu = lp.nu; L.alpha = lp.alpha; L.num_lookups = (int)lp.lookups.size(); L.seg_shift = lp.seg_shift; L.rows = lp.rows;
    L.seg_lookup = upload_vec(pk.get(), lp.seg_lookup);
    for (size_t l = 0; l < lp.lookups.size(); l++) {
        int tb = lp.lookups[l].total_bits;
        L.lookup_mask[l] = tb >= 64 ? ~0ULL : ((1ULL << tb) - 1);
        L.lookup_nmems[l] = (int)lp.lookups[l].mems.size();
        if (lp.lookups[l].mems.size() > 4) throw Error("lasso: lookup touches more than C memories");
        for (size_t i = 0; i < lp.lookups[l].mems.size(); i++) {
            L.lookup_uses[l] |= 1ULL << lp.lookups[l].mems[i];
            L.lookup_mems[l][i] = lp.lookups[l].mems[i];
        }
    }
    for (int m = 0; m < lp.alpha; m++) { L.mem_dim[m] = lp.mems[m].dim; L.mem_cutoff[m] = (u32)lp.mems[m].cutoff; }
    if (lp.seg_lookup.size() > 128) throw Error("lasso: more than 128 row segments");
    for (auto& chk : lp.chunks) {
        int c = chk.first;  // counter memory = memory index c (lasso.rs:317-319)
        if (c >= 4) throw Error("lasso: chunk index exceeds C");
        for (size_t sg = 0; sg < lp.seg_lookup.size(); sg++)
            if ((L.lookup_uses[lp.seg_lookup[sg]] >> c) & 1) L.cnt_segs[c][L.cnt_nsegs[c]++] = (uint8_t)sg;
    }
    L.mpow[0] = 1;
    for (int i = 1; i < 5; i++) L.mpow[i] = gl_mul(L.mpow[i - 1], 65536);
    // circuit wiring as CSR per (node, input)
    const HCircuit& c = pk->circuit;
    pk->node_dev.resize(c.nodes.size());
    for (size_t id = 0; id < c.nodes.size(); id++) {   // twiddle tables, one per transform size and direction
        const HNode& n = c.nodes[id];
        if (n.kind == NK_FFT) {
            for (int inv = 0; inv < 2; inv++) {
                auto& tab = inv ? pk->w_inv : pk->w_fwd;
                if (tab.count(n.log2_size)) continue;
                size_t N = (size_t)1 << n.log2_size;
                std::vector<u64> W(N);
                u64 w = root_of_unity(n.log2_size);
                if (inv) w = gl_inv(w);
                W[0] = 1;
                for (size_t i = 1; i < N; i++) W[i] = gl_mul(W[i - 1], w);
                tab[n.log2_size] = upload_vec(pk.get(), W);
            }
        }
    }
    lap("twiddle tables");
    // the Vanilla nodes' wiring in the forms the kernels read, node by node on all host threads (round 6: this loop was 0.8 of the
    // 1.1 s a set-up took on eight cores): a node touches its own NodeDev only, uploads go through upload_vec's lock
    std::exception_ptr node_error;
    std::vector<size_t> by_size;
    for (size_t id = 0; id < c.nodes.size(); id++) if (c.nodes[id].kind == NK_VANILLA) by_size.push_back(id);
    std::sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return c.nodes[a].lin.size() + 2 * c.nodes[a].mul.size() > c.nodes[b].lin.size() + 2 * c.nodes[b].mul.size(); });
    [[maybe_unused]] const int setup_threads = std::max(1, std::min(hg_omp_threads(), 64));
#pragma omp parallel for schedule(dynamic, 1) num_threads(setup_threads)
    for (long long qi = 0; qi < (long long)by_size.size(); qi++) {
      try {
        const size_t id = by_size[(size_t)qi];
        const HNode& n = c.nodes[id];
        hg_pk::NodeDev& nd = pk->node_dev[id];
        const size_t S = (size_t)1 << n.log2_sub_in;
        nd.lin.assign(n.arity, dev::CsrLin{nullptr, nullptr, nullptr});
        nd.mulL.assign(n.arity, dev::CsrMul{nullptr, nullptr, nullptr, nullptr, nullptr});
        nd.mulR.assign(n.arity, dev::CsrMul{nullptr, nullptr, nullptr, nullptr, nullptr});
        nd.seg.assign(n.arity, hg_pk::NodeDev::Seg{});
        std::vector<std::vector<const LinTerm*>> lin_by(n.arity);
        std::vector<std::vector<const MulTerm*>> ml_by(n.arity), mr_by(n.arity);
        for (auto& t : n.lin) lin_by[t.in].push_back(&t);
        for (auto& t : n.mul) { ml_by[t.i0].push_back(&t); mr_by[t.i1].push_back(&t); }
        std::vector<std::vector<dev::GatherSeg>> segs_of(n.arity);   // run-length form per input (empty: not affine), for eq_form below
        for (int i = 0; i < n.arity; i++) {
            if (!lin_by[i].empty()) {
                std::vector<u32> ptr, gate(lin_by[i].size());
                std::vector<u64> coef(lin_by[i].size());
                build_csr<LinTerm>(lin_by[i], S, [](const LinTerm& t) { return (size_t)t.j; }, ptr,
                                   [&](const LinTerm& t, u32 at) { gate[at] = t.gate; coef[at] = t.c; });
                nd.lin[i] = dev::CsrLin{upload_vec(pk.get(), ptr), upload_vec(pk.get(), gate), upload_vec(pk.get(), coef)};
            }
            auto mk = [&](const std::vector<const MulTerm*>& ts, bool left) {
                std::vector<u32> ptr, gate(ts.size()), oin(ts.size()), oj(ts.size());
                std::vector<u64> coef(ts.size());
                build_csr<MulTerm>(ts, S, [left](const MulTerm& t) { return (size_t)(left ? t.j0 : t.j1); }, ptr,
                                   [&](const MulTerm& t, u32 at) {
                                       gate[at] = t.gate; coef[at] = t.c;
                                       oin[at] = left ? t.i1 : t.i0;
                                       oj[at] = left ? t.j1 : t.j0;
                                   });
                return dev::CsrMul{upload_vec(pk.get(), ptr), upload_vec(pk.get(), gate), upload_vec(pk.get(), coef),
                                   upload_vec(pk.get(), oin), upload_vec(pk.get(), oj)};
            };
            if (!ml_by[i].empty()) nd.mulL[i] = mk(ml_by[i], true);
            if (!mr_by[i].empty()) nd.mulR[i] = mk(mr_by[i], false);
            {   // run-length segments of the phase-1 table of input i
                struct Ent { int other_in; long long goff, joff; u64 c; u32 x; };
                std::vector<Ent> ents;
                ents.reserve(lin_by[i].size() + ml_by[i].size());
                for (const LinTerm* t : lin_by[i]) ents.push_back(Ent{-1, (long long)t->gate - (long long)t->j, 0, t->c, t->j});
                for (const MulTerm* t : ml_by[i]) ents.push_back(Ent{(int)t->i1, (long long)t->gate - (long long)t->j0, (long long)t->j1 - (long long)t->j0, t->c, t->j0});
                std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) {
                    return std::tie(a.other_in, a.goff, a.joff, a.c, a.x) < std::tie(b.other_in, b.goff, b.joff, b.c, b.x);
                });
                std::vector<dev::GatherSeg> segs;
                bool ok = !ents.empty();
                for (size_t e = 0; e < ents.size() && ok; e++) {
                    const Ent& t = ents[e];
                    if (!segs.empty()) {
                        dev::GatherSeg& g = segs.back();
                        const bool same = g.other_in == t.other_in && g.goff == (int)t.goff && g.joff == (int)t.joff && g.coef == t.c;
                        if (same && t.x == g.hi) { g.hi++; continue; }
                        if (same && t.x < g.hi) { ok = false; break; }  // the same term twice: leave it to the general form
                    }
                    if (segs.size() >= 64 || t.goff < INT32_MIN || t.goff > INT32_MAX) { ok = false; break; }
                    segs.push_back(dev::GatherSeg{t.x, t.x + 1, (int)t.goff, t.other_in, (int)t.joff, 0, t.c});
                }
                if (ok) {
                    segs_of[i] = segs;
                    hg_pk::NodeDev::Seg& sg = nd.seg[i];
                    sg.nseg = (int)segs.size();
                    sg.d = upload_vec(pk.get(), segs);
                    const dev::GatherSeg& g = segs[0];
                    sg.alias = segs.size() == 1 && g.other_in < 0 && g.coef == 1 && g.lo == 0 && g.hi == S && g.goff >= 0 &&
                               (n.log2_reps == 0 || (n.log2_sub_out == n.log2_sub_in && g.goff == 0));
                    sg.alias_off = (size_t)g.goff;
                    sg.win_lo = segs[0].lo; sg.win_hi = segs[0].hi;
                    for (auto& q : segs) { sg.win_lo = std::min<size_t>(sg.win_lo, q.lo); sg.win_hi = std::max<size_t>(sg.win_hi, q.hi); }
                }
            }
        }
        {   // eq-factored form of the node's Libra phase 1 (kernels.hpp PsJob::eq_n): no mul gates, every used input relays the same
            // aligned window [lo, lo + 2^w) of its positions onto aligned 2^w blocks of gates (any constant coefficient per block),
            // constants constant over such blocks. Table i is then kappa_i eq(z', .), kappa_i = sum_t coef_t eq(z_(w..); block_t).
            hg_pk::NodeDev::EqForm& ef = nd.eq_form;
            const int nin = n.log2_sub_in + n.log2_reps;
            bool ok = n.mul.empty() && nin <= dev::PS_EQ_MAX_VARS;
            long long lo0 = -1; int w0 = -1;
            ef.terms.assign(n.arity, {});
            int used = 0;
            for (int i = 0; i < n.arity && ok; i++) {
                if (!n.left_use[i]) continue;
                used++;
                if (segs_of[i].empty()) { ok = false; break; }
                for (const dev::GatherSeg& g : segs_of[i]) {
                    if (g.other_in >= 0) { ok = false; break; }
                    if (n.log2_reps > 0) {   // replicated sub-circuit: only the identity relay (gate = position in every repetition)
                        if (!(n.log2_sub_out == n.log2_sub_in && g.lo == 0 && g.hi == S && g.goff == 0)) { ok = false; break; }
                        if (w0 >= 0 && (lo0 != 0 || w0 != nin)) { ok = false; break; }
                        lo0 = 0; w0 = nin;
                        ef.terms[i].push_back({g.coef, 0u});
                        continue;
                    }
                    const u64 len = (u64)g.hi - g.lo;
                    const long long g0 = (long long)g.goff + (long long)g.lo;
                    if (len == 0 || (len & (len - 1)) || (g.lo & (len - 1)) || g0 < 0 || ((u64)g0 & (len - 1))) { ok = false; break; }
                    const int w = 63 - __builtin_clzll(len);
                    if (w0 >= 0 && (lo0 != (long long)g.lo || w0 != w)) { ok = false; break; }
                    lo0 = g.lo; w0 = w;
                    ef.terms[i].push_back({g.coef, (u32)((u64)g0 >> w)});
                }
            }
            if (ok && used > 0 && !n.w0.empty()) {
                if (n.log2_reps > 0) ok = false;
                // block-uniform means: EVERY gate of the block carries the constant exactly once, with one value. Counted per DISTINCT gate
                // (a wiring with two constant terms on some gates of a block and none on others has 2^w entries too)
                std::map<u32, std::pair<u64, size_t>> blocks;   // block -> (value, distinct gates seen)
                std::vector<u32> gates;
                gates.reserve(n.w0.size());
                for (auto& t : n.w0) gates.push_back(t.gate);
                std::sort(gates.begin(), gates.end());
                if (std::adjacent_find(gates.begin(), gates.end()) != gates.end()) ok = false;   // a gate with two constant terms: not the form
                for (auto& t : n.w0) {
                    if (!ok) break;
                    auto it = blocks.find(t.gate >> w0);
                    if (it == blocks.end()) blocks[t.gate >> w0] = {t.c, 1};
                    else if (it->second.first != t.c) ok = false;
                    else it->second.second++;
                }
                for (auto& kv : blocks) { if (kv.second.second != ((size_t)1 << w0)) ok = false; ef.consts.push_back({kv.second.first, kv.first}); }
            }
            ef.ok = ok && used > 0 && w0 >= 0;
            ef.w = w0; ef.hib = ef.ok ? (unsigned)((u64)lo0 >> w0) : 0;
            if (!ef.ok) { ef.terms.clear(); ef.consts.clear(); }
        }
        {   // gate-major (forward) wiring for witness generation on the device
            dev::EvalNode& f = nd.fwd;
            memset(&f, 0, sizeof(f));
            f.log2_G = n.log2_sub_out; f.log2_S = n.log2_sub_in; f.log2_R = n.log2_reps; f.num_gates = n.num_gates;
            const size_t G = n.num_gates;
            if (!n.lin.empty()) {
                std::vector<u32> ptr(G + 1, 0), in(n.lin.size()), jj(n.lin.size());
                std::vector<u64> cf(n.lin.size());
                for (auto& t : n.lin) ptr[t.gate + 1]++;
                for (size_t i = 0; i < G; i++) ptr[i + 1] += ptr[i];
                std::vector<u32> fill(ptr.begin(), ptr.end() - 1);
                for (auto& t : n.lin) { u32 at = fill[t.gate]++; in[at] = t.in; jj[at] = t.j; cf[at] = t.c; }
                f.lptr = upload_vec(pk.get(), ptr); f.lin_in = upload_vec(pk.get(), in); f.lin_j = upload_vec(pk.get(), jj); f.lcoef = upload_vec(pk.get(), cf);
            }
            if (!n.mul.empty()) {
                std::vector<u32> ptr(G + 1, 0), i0(n.mul.size()), j0(n.mul.size()), i1(n.mul.size()), j1(n.mul.size());
                std::vector<u64> cf(n.mul.size());
                for (auto& t : n.mul) ptr[t.gate + 1]++;
                for (size_t i = 0; i < G; i++) ptr[i + 1] += ptr[i];
                std::vector<u32> fill(ptr.begin(), ptr.end() - 1);
                for (auto& t : n.mul) { u32 at = fill[t.gate]++; i0[at] = t.i0; j0[at] = t.j0; i1[at] = t.i1; j1[at] = t.j1; cf[at] = t.c; }
                f.mptr = upload_vec(pk.get(), ptr); f.mi0 = upload_vec(pk.get(), i0); f.mj0 = upload_vec(pk.get(), j0);
                f.mi1 = upload_vec(pk.get(), i1); f.mj1 = upload_vec(pk.get(), j1); f.mcoef = upload_vec(pk.get(), cf);
            }
            if (!n.w0.empty()) {
                std::vector<u64> dense(G, 0);
                for (auto& t : n.w0) dense[t.gate] = gl_add(dense[t.gate], t.c);
                f.w0 = upload_vec(pk.get(), dense);
            }
        }
        if (!n.w0.empty()) {
            std::vector<u32> g;
            std::vector<u64> cf;
            for (auto& t : n.w0) { g.push_back(t.gate); cf.push_back(t.c); }
            nd.const_gate = upload_vec(pk.get(), g);
            nd.const_coef = upload_vec(pk.get(), cf);
            nd.nconst = g.size();
        }
      } catch (...) {
#pragma omp critical(hg_setup_error)
        if (!node_error) node_error = std::current_exception();
      }
    }
    if (node_error) std::rethrow_exception(node_error);
    lap("node wiring (CSR, segments, eq forms, gate-major form) built and uploaded");
    *out = pk.release();
    return 0;
    HG_CATCH(-1)
}

void hg_pk_free(hg_pk* pk) {
    if (!pk) return;
    for (void* p : pk->owned) (void)hipFree(p);
    delete pk;
}

int hg_pk_lasso_layout(const hg_pk* pk, char* out, size_t cap) {
    HG_TRY
    if (!pk || !out) throw Error("hg_pk_lasso_layout: null argument");
    std::string s = pk->lasso.layout_text();
    if (s.size() + 1 > cap) throw Error("buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
    HG_CATCH(-1)
}

int hg_pk_info(const hg_pk* pk, uint64_t out[6]) {
    if (!pk || !out) { g_last_error = "hg_pk_info: null argument"; return -1; }
    out[4] = (uint64_t)pk->circuit.lasso_in_id;
    out[5] = (uint64_t)pk->circuit.sum_id;
    out[0] = (uint64_t)pk->lasso.nu;
    out[1] = pk->circuit.nodes.size();
    out[2] = pk->lasso.rows;
    out[3] = (uint64_t)pk->lasso.alpha;
    return 0;
}

int hg_pk_node_eq_form(const hg_pk* pk, int node, int64_t out[6]) {
    if (!pk || !out) { g_last_error = "hg_pk_node_eq_form: null argument"; return -1; }
    if (node < 0 || (size_t)node >= pk->circuit.nodes.size()) { g_last_error = "hg_pk_node_eq_form: no such node"; return -1; }
    const HNode& n = pk->circuit.nodes[node];
    out[0] = n.kind == NK_VANILLA ? 1 : (n.kind == NK_FFT ? 2 : (n.kind == NK_LASSO ? 3 : 0));
    out[1] = out[2] = out[3] = out[4] = out[5] = 0;
    if (n.kind != NK_VANILLA) return 0;
    const hg_pk::NodeDev::EqForm& ef = pk->node_dev[node].eq_form;
    out[1] = ef.ok ? 1 : 0;
    out[2] = ef.w; out[3] = ef.hib;
    for (auto& t : ef.terms) out[4] += (int64_t)t.size();
    out[5] = n.log2_sub_in + n.log2_reps;
    return 0;
}

int hg_witness_from_json(const hg_params* params, const char* path, hg_witness** w) {
    HG_TRY
    if (!params || !path || !w) throw Error("hg_witness_from_json: null argument");
    Params p(*params);
    std::unique_ptr<hg_witness> hw(new hg_witness{witness_from_json(p, path), *params});
    *w = hw.release();
    return 0;
    HG_CATCH(-1)
}

int hg_witness_synthetic(const hg_params* params, uint64_t seed, hg_witness** w) {
    HG_TRY
    if (!params || !w) throw Error("hg_witness_synthetic: null argument");
    Params p(*params);
    std::unique_ptr<hg_witness> hw(new hg_witness{witness_synthetic(p, seed), *params});
    *w = hw.release();
    return 0;
    HG_CATCH(-1)
}

int hg_witness_from_arrays(const hg_params* params, const uint64_t* s, const uint64_t* e, const uint64_t* k1, const uint64_t* ais,
                           const uint64_t* r1is, const uint64_t* r2is, const uint64_t* ct0is, hg_witness** w) {
    HG_TRY
    if (!params || !s || !e || !k1 || !ais || !r1is || !r2is || !ct0is || !w) throw Error("hg_witness_from_arrays: null argument");
    Params p(*params);
    const size_t SZ = p.SZ(), PZ = p.PZ(), k = (size_t)p.k;
    std::unique_ptr<hg_witness> hw(new hg_witness());
    hw->params = *params;
    auto cp = [](std::vector<u64>& dst, const u64* src, size_t n) {
        dst.assign(src, src + n);
        for (u64 v : dst) if (v >= GL_P) throw Error("witness: non-canonical field element");
    };
    cp(hw->w.s, s, SZ); cp(hw->w.e, e, SZ); cp(hw->w.k1, k1, SZ);
    cp(hw->w.ais, ais, k * SZ); cp(hw->w.r1is, r1is, k * SZ); cp(hw->w.r2is, r2is, k * PZ); cp(hw->w.ct0is, ct0is, k * SZ);
    *w = hw.release();
    return 0;
    HG_CATCH(-1)
}

int64_t hg_witness_get(const hg_witness* w, int which, uint64_t* out, size_t cap) {
    if (!w) { g_last_error = "hg_witness_get: null witness"; return -1; }
    const std::vector<u64>* v = nullptr;
    switch (which) {
        case 0: v = &w->w.s; break;
        case 1: v = &w->w.e; break;
        case 2: v = &w->w.k1; break;
        case 3: v = &w->w.ais; break;
        case 4: v = &w->w.r1is; break;
        case 5: v = &w->w.r2is; break;
        case 6: v = &w->w.ct0is; break;
        default: g_last_error = "hg_witness_get: bad selector"; return -1;
    }
    if (out) memcpy(out, v->data(), std::min(cap, v->size()) * 8);
    return (int64_t)v->size();
}

void hg_witness_free(hg_witness* w) { delete w; }

int hg_witness_gen(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, hg_values** out, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !w || !out || !pk->ctx) throw Error("hg_witness_gen: needs a device context and a device prover key");
    check_witness(pk, w, "hg_witness_gen");
    double wm = 0, um = 0;
    *out = witness_gen(ctx, pk, w->w, &wm, &um);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->witness_ms = wm; timings->upload_ms = um; }
    return 0;
    HG_CATCH(-1)
}

int hg_witness_gen_into(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, hg_values* v, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !w || !v || !pk->ctx) throw Error("hg_witness_gen_into: needs a device context, a device prover key and an existing values object");
    check_witness(pk, w, "hg_witness_gen_into");
    double wm = 0, um = 0;
    witness_gen_into(ctx, pk, w->w, v, &wm, &um);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->witness_ms = wm; timings->upload_ms = um; }
    return 0;
    HG_CATCH(-1)
}

int hg_witness_gen_shard(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, int rank, int world, hg_values** out, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !w || !out || !pk->ctx) throw Error("hg_witness_gen_shard: needs a device context and a device prover key");
    check_witness(pk, w, "hg_witness_gen_shard");
    double wm = 0, um = 0;
    *out = witness_gen_shard(ctx, pk, w->w, rank, world, &wm, &um);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->witness_ms = wm; timings->upload_ms = um; }
    return 0;
    HG_CATCH(-1)
}
int hg_values_info(const hg_values* v, uint64_t out[4]) {
    HG_TRY
    if (!v || !out) throw Error("hg_values_info: null argument");
    size_t n = 0;
    for (auto p : v->d_vals) n += p != nullptr;
    size_t full = v->full_bytes, res = v->resident_bytes;
    if (v->shard_rank < 0) { full = v->ct0is_len * 8; for (size_t q : v->sizes) full += q * 8; res = full; }
    out[0] = res; out[1] = full; out[2] = n; out[3] = v->d_vals.size();
    return 0;
    HG_CATCH(-1)
}

int64_t hg_values_peak_bytes(const hg_values* v) {
    if (!v) return -1;
    uint64_t info[4];
    if (hg_values_info(v, info) != 0) return -1;
    return (int64_t)(info[0] + v->cone_bytes);
}

void hg_values_free(hg_values* v) { values_free(v); }

int64_t hg_values_get(hg_ctx* ctx, const hg_values* v, int node, uint64_t* out, size_t cap) {
    HG_TRY
    if (out && !ctx) throw Error("hg_values_get: null context");
    if (!v || node < 0 || (size_t)node >= v->d_vals.size()) throw Error("hg_values_get: bad node id");
    size_t n = v->sizes[node];
    if (out && v->d_vals[node]) {
        hip_check(hipSetDevice(ctx->device), "hipSetDevice");
        hip_check(hipMemcpy(out, v->d_vals[node], std::min(cap, n) * 8, hipMemcpyDeviceToHost), "download node values");
    }
    return (int64_t)n;
    HG_CATCH(-1)
}


int hg_prove_resident(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !v || !pk->ctx) throw Error("hg_prove_resident: needs a device context, a device prover key and resident values");
    double t0 = now_ms_capi();
    ProveResult r = prove_resident(ctx, pk, v, true);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->prove_ms = r.prove_ms; timings->gpu_ms = r.gpu_ms; timings->total_ms = now_ms_capi() - t0; timings->enqueue_ms = r.enqueue_ms; timings->sync_ms = r.sync_ms; timings->replay_ms = r.replay_ms; }
    const std::vector<uint8_t>& pb = r.bytes();
    *len = pb.size();
    if (pb.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, pb.data(), pb.size());
    return 0;
    HG_CATCH(-1)
}

int hg_prove_shard_begin(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world, uint64_t** partial, size_t* n_u64) {
    HG_TRY
    if (!ctx || !pk || !v || !pk->ctx) throw Error("hg_prove_shard_begin: needs a device context, a device prover key and resident values");
    size_t n = prove_shard_begin(ctx, pk, v, rank, world);
    *partial = reinterpret_cast<uint64_t*>(ctx->h_res);
    *n_u64 = 2 * n;
    return 0;
    HG_CATCH(-1)
}

int hg_prove_shard_combine(hg_ctx* ctx, const uint64_t* gathered, int world, size_t n_u64) {
    HG_TRY
    if (!ctx || !gathered || world < 1) throw Error("hg_prove_shard_combine: bad argument");
    prove_shard_combine(ctx, gathered, world, n_u64);
    return 0;
    HG_CATCH(-1)
}

int hg_shard_combine_host(const uint64_t* gathered, int world, size_t n_u64, uint64_t* out) {
    HG_TRY
    if (!gathered || !out || world < 1) throw Error("hg_shard_combine_host: bad argument");
    shard_combine_host(gathered, world, n_u64, out);
    return 0;
    HG_CATCH(-1)
}

int hg_prove_shard_finish(hg_ctx* ctx, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings) {
    HG_TRY
    if (!ctx) throw Error("hg_prove_shard_finish: null context");
    ProveResult r = prove_shard_finish(ctx);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->prove_ms = r.prove_ms; timings->gpu_ms = r.gpu_ms; timings->total_ms = r.prove_ms; timings->enqueue_ms = r.enqueue_ms; timings->sync_ms = r.sync_ms; timings->replay_ms = r.replay_ms; }
    *len = r.proof.size();
    if (r.proof.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, r.proof.data(), r.proof.size());
    return 0;
    HG_CATCH(-1)
}

int hg_comm_unique_id(uint8_t out[128]) {
    HG_TRY
    if (!out) throw Error("hg_comm_unique_id: null argument");
    comm_unique_id(out);
    return 0;
    HG_CATCH(-1)
}
int hg_comm_init(hg_ctx* ctx, const uint8_t id[128], int rank, int world) {
    HG_TRY
    if (!ctx || !id) throw Error("hg_comm_init: null argument");
    comm_init(ctx, id, rank, world);
    return 0;
    HG_CATCH(-1)
}
int hg_comm_destroy(hg_ctx* ctx) {
    HG_TRY
    if (!ctx) throw Error("hg_comm_destroy: null context");
    comm_destroy(ctx);
    return 0;
    HG_CATCH(-1)
}
int hg_comm_count(hg_ctx* ctx, int* ranks) {
    HG_TRY
    if (!ctx || !ranks) throw Error("hg_comm_count: null argument");
    *ranks = comm_count(ctx);
    return 0;
    HG_CATCH(-1)
}
int hg_comm_selftest(hg_ctx* ctx, const uint64_t* rank_buffers, int world, size_t n_u64, uint64_t* out) {
    HG_TRY
    if (!ctx || !rank_buffers || !out) throw Error("hg_comm_selftest: null argument");
    comm_selftest(ctx, rank_buffers, world, n_u64, out);
    return 0;
    HG_CATCH(-1)
}
int hg_prove_sharded(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !v || !pk->ctx || !proof || !len) throw Error("hg_prove_sharded: needs a device context, a device prover key and resident values");
    ProveResult r = prove_sharded(ctx, pk, v);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->prove_ms = r.prove_ms; timings->gpu_ms = r.gpu_ms; timings->total_ms = r.prove_ms; timings->enqueue_ms = r.enqueue_ms; timings->sync_ms = r.sync_ms; timings->replay_ms = r.replay_ms; }
    *len = r.proof.size();
    if (r.proof.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, r.proof.data(), r.proof.size());
    return 0;
    HG_CATCH(-1)
}

int hg_prove(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !w) throw Error("hg_prove: null argument");
    if (!pk->ctx) throw Error("hg_prove: host-only prover key (created without a context)");
    if (!proof || !len) throw Error("hg_prove: null output argument");
    check_witness(pk, w, "hg_prove");
    double t0 = now_ms_capi();
    double wm = 0, um = 0;
    // the node tables live in a values object owned by the context and are refilled in place, so the launch graph recorded for
    // them (third hg_prove with one key) proves every later witness without a protocol walk
    if (ctx->scratch_values && ctx->scratch_serial != pk->serial) { values_free(ctx->scratch_values); ctx->scratch_values = nullptr; }
    if (!ctx->scratch_values) { ctx->scratch_values = witness_gen(ctx, pk, w->w, &wm, &um); ctx->scratch_serial = pk->serial; }
    else witness_gen_into_staged(ctx, pk, w->w, ctx->scratch_values, &wm, &um);
    ProveResult r = prove_resident(ctx, pk, ctx->scratch_values, true);
    if (timings) { timings->witness_ms = wm; timings->upload_ms = um; timings->prove_ms = r.prove_ms; timings->gpu_ms = r.gpu_ms; timings->total_ms = now_ms_capi() - t0; timings->enqueue_ms = r.enqueue_ms; timings->sync_ms = r.sync_ms; timings->replay_ms = r.replay_ms; }
    const std::vector<uint8_t>& pb = r.bytes();
    *len = pb.size();
    if (pb.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, pb.data(), pb.size());
    return 0;
    HG_CATCH(-1)
}

int hg_warmup(hg_ctx* ctx, const hg_pk* pk, double* ms) {
    HG_TRY
    if (!ctx || !pk) throw Error("hg_warmup: null argument");
    if (!pk->ctx) throw Error("hg_warmup: host-only prover key (created without a context)");
    const double t = prove_warmup(ctx, pk);
    if (ms) *ms = t;
    return 0;
    HG_CATCH(-1)
}

int hg_prove_stream(hg_ctx* ctx, const hg_pk* pk, const hg_witness* const* ws, size_t n, uint8_t* proofs, size_t cap_each, size_t* lens, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || (n && (!ws || !proofs || !lens))) throw Error("hg_prove_stream: null argument");
    if (!pk->ctx) throw Error("hg_prove_stream: host-only prover key (created without a context)");
    std::vector<const Witness*> W(n);
    for (size_t i = 0; i < n; i++) {
        if (!ws[i]) throw Error("hg_prove_stream: null witness");
        check_witness(pk, ws[i], "hg_prove_stream");
        W[i] = &ws[i]->w;
    }
    double total = 0;
    std::vector<ProveResult> rs = prove_stream(ctx, pk, W, &total);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->total_ms = total; }
    for (size_t i = 0; i < n; i++) {
        lens[i] = rs[i].proof.size();
        if (rs[i].proof.size() > cap_each) throw Error("proof buffer too small");
        memcpy(proofs + i * cap_each, rs[i].proof.data(), rs[i].proof.size());
        if (timings) { timings->prove_ms += rs[i].prove_ms; timings->gpu_ms += rs[i].gpu_ms; timings->replay_ms += rs[i].replay_ms; }
    }
    return 0;
    HG_CATCH(-1)
}

int hg_prove_resident_mode(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !v || !pk->ctx || !proof || !len) throw Error("hg_prove_resident_mode: needs a device context, a device prover key and resident values");
    double t0 = now_ms_capi();
    ProveResult r = prove_resident_mode(ctx, pk, v, mode);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->prove_ms = r.prove_ms; timings->gpu_ms = r.gpu_ms; timings->total_ms = now_ms_capi() - t0; timings->enqueue_ms = r.enqueue_ms; timings->sync_ms = r.sync_ms; timings->replay_ms = r.replay_ms; }
    *len = r.proof.size();
    if (r.proof.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, r.proof.data(), r.proof.size());
    return 0;
    HG_CATCH(-1)
}
// ---- ranks of a sharded round-by-round prove ------------------------------------------------------------------------------------
struct hg_group {
    int world = 1;
    // external: the caller's all-reduce
    int (*fn)(void*, uint64_t*, size_t) = nullptr;
    void* user = nullptr;
    // local: ranks are threads of this process
    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint64_t> acc, res;
    int arrived = 0;
    unsigned long long gen = 0;
    bool broken = false;
};
static int group_reduce(void* gp, uint64_t* words, size_t n) {
    hg_group* g = static_cast<hg_group*>(gp);
    if (g->fn) return g->fn(g->user, words, n);
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return -1;
    if (g->arrived == 0) g->acc.assign(words, words + n);
    else {
        if (g->acc.size() != n) { g->broken = true; g->cv.notify_all(); return -1; }   // the ranks are not in the same round
        for (size_t i = 0; i < n; i++) g->acc[i] = gl_add(g->acc[i], words[i]);
    }
    if (++g->arrived == g->world) {
        g->res = g->acc;
        g->arrived = 0;
        g->gen++;
        g->cv.notify_all();
    } else {
        const unsigned long long mine = g->gen;
        // a rank that never arrives (it failed) must not hold the others for ever
        if (!g->cv.wait_for(lk, std::chrono::seconds(20), [&] { return g->gen != mine || g->broken; }) || g->broken) { g->broken = true; g->cv.notify_all(); return -1; }
    }
    std::copy(g->res.begin(), g->res.end(), words);
    return 0;
}
hg_group* hg_group_local(int world) {
    HG_TRY
    if (world < 1 || world > 64) throw Error("hg_group_local: world out of range");
    hg_group* g = new hg_group();
    g->world = world;
    return g;
    HG_CATCH(nullptr)
}
hg_group* hg_group_external(void* reduce_fn, void* user, int world) {
    HG_TRY
    if (world < 1 || !reduce_fn) throw Error("hg_group_external: needs a reduce function and a world size");
    hg_group* g = new hg_group();
    g->world = world;
    g->fn = reinterpret_cast<int (*)(void*, uint64_t*, size_t)>(reduce_fn);
    g->user = user;
    return g;
    HG_CATCH(nullptr)
}
void hg_group_free(hg_group* g) { delete g; }
int hg_prove_resident_mode_sharded(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode, int rank, hg_group* group, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !v || !pk->ctx || !proof || !len || !group) throw Error("hg_prove_resident_mode_sharded: needs a device context, a device prover key, resident values and a group");
    double t0 = now_ms_capi();
    ProveResult r = prove_resident_mode_sharded(ctx, pk, v, mode, rank, group->world, group_reduce, group);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->prove_ms = r.prove_ms; timings->gpu_ms = r.gpu_ms; timings->total_ms = now_ms_capi() - t0; timings->enqueue_ms = r.enqueue_ms; timings->sync_ms = r.sync_ms; timings->replay_ms = r.replay_ms; }
    *len = r.proof.size();
    if (r.proof.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, r.proof.data(), r.proof.size());
    return 0;
    HG_CATCH(-1)
}
int hg_prove_mode(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, int mode, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings) {
    HG_TRY
    if (!ctx || !pk || !w || !proof || !len) throw Error("hg_prove_mode: null argument");
    if (!pk->ctx) throw Error("hg_prove_mode: host-only prover key (created without a context)");
    check_witness(pk, w, "hg_prove_mode");
    double t0 = now_ms_capi();
    double wm = 0, um = 0;
    hg_values* v = witness_gen(ctx, pk, w->w, &wm, &um);
    ProveResult r;
    try { r = prove_resident_mode(ctx, pk, v, mode); } catch (...) { values_free(v); throw; }
    values_free(v);
    if (timings) { memset(timings, 0, sizeof(*timings)); timings->witness_ms = wm; timings->upload_ms = um; timings->prove_ms = r.prove_ms; timings->gpu_ms = r.gpu_ms; timings->total_ms = now_ms_capi() - t0; timings->enqueue_ms = r.enqueue_ms; timings->sync_ms = r.sync_ms; timings->replay_ms = r.replay_ms; }
    *len = r.proof.size();
    if (r.proof.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, r.proof.data(), r.proof.size());
    return 0;
    HG_CATCH(-1)
}
int hg_verify_mode(const hg_pk* pk, const hg_witness* w, int mode, const uint8_t* proof, size_t len) {
    HG_TRY
    if (!pk || !w || !proof) throw Error("hg_verify_mode: null argument");
    if (mode < 0 || mode > 3) throw Error("hg_verify_mode: unknown mode bits");
    check_witness(pk, w, "hg_verify_mode");
    std::string why = verify_proof(pk->params, pk->lasso, pk->circuit, w->w, proof, len, mode);
    if (why.empty()) return 0;
    g_last_error = why;
    return 1;
    HG_CATCH(-1)
}

int hg_verify(const hg_pk* pk, const hg_witness* w, const uint8_t* proof, size_t len) {
    HG_TRY
    if (!pk || !w || !proof) throw Error("hg_verify: null argument");
    check_witness(pk, w, "hg_verify");
    std::string why = verify_proof(pk->params, pk->lasso, pk->circuit, w->w, proof, len);
    if (why.empty()) return 0;
    g_last_error = why;
    return 1;
    HG_CATCH(-1)
}

int hg_verify_device(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, const uint8_t* proof, size_t len) {
    HG_TRY
    if (!ctx || !pk || !w || !proof || !pk->ctx) throw Error("hg_verify_device: needs a device context and a device prover key");
    check_witness(pk, w, "hg_verify_device");
    std::string why = verify_proof_device(ctx, pk, w->w, proof, len);
    if (why.empty()) return 0;
    g_last_error = why;
    return 1;
    HG_CATCH(-1)
}

int hg_verify_bn254(const hg_pk* pk, const hg_witness* w, const uint8_t* proof, size_t len) {
    HG_TRY
    if (!pk || !w || !proof) throw Error("hg_verify_bn254: null argument");
    check_witness(pk, w, "hg_verify_bn254");
    std::string why = verify_proof_bn254(pk->params, pk->lasso, pk->circuit, w->w, proof, len);
    if (why.empty()) return 0;
    g_last_error = why;
    return 1;
    HG_CATCH(-1)
}

int hg_circuit_eval(const hg_pk* pk, const hg_witness* w, uint64_t* lasso_in, size_t lasso_cap, uint64_t* sum_out, size_t sum_cap) {
    HG_TRY
    if (!pk || !w) throw Error("hg_circuit_eval: null argument");
    check_witness(pk, w, "hg_circuit_eval");
    auto vals = circuit_evaluate(pk->circuit, pk->params, w->w);
    const auto& li = vals[pk->circuit.lasso_in_id];
    const auto& so = vals[pk->circuit.sum_id];
    if (lasso_in) { if (li.size() > lasso_cap) throw Error("lasso_in buffer too small"); memcpy(lasso_in, li.data(), li.size() * 8); }
    if (sum_out) { if (so.size() > sum_cap) throw Error("sum_out buffer too small"); memcpy(sum_out, so.data(), so.size() * 8); }
    return 0;
    HG_CATCH(-1)
}

int hg_lasso_prove(hg_ctx* ctx, const hg_pk* pk, const uint64_t* lasso_in, uint8_t* proof, size_t cap, size_t* len, uint64_t* claim_out) {
    return hg_lasso_prove_at(ctx, pk, lasso_in, 0, proof, cap, len, claim_out);
}

int hg_lasso_prove_at(hg_ctx* ctx, const hg_pk* pk, const uint64_t* lasso_in, size_t chain_skip, uint8_t* proof, size_t cap, size_t* len,
                      uint64_t* claim_out) {
    HG_TRY
    if (!ctx || !pk || !pk->ctx) throw Error("hg_lasso_prove: needs a device context and a device prover key");
    if (!lasso_in || !proof || !len) throw Error("hg_lasso_prove: null argument");
    std::vector<E2> claim;
    std::vector<uint8_t> pr = prove_lasso_node(ctx, pk, lasso_in, chain_skip, &claim);
    *len = pr.size();
    if (pr.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, pr.data(), pr.size());
    if (claim_out) for (size_t i = 0; i < claim.size(); i++) { claim_out[2 * i] = claim[i].c0; claim_out[2 * i + 1] = claim[i].c1; }
    return 0;
    HG_CATCH(-1)
}

int hg_lasso_num_challenges(const hg_pk* pk, size_t* n_e) {
    HG_TRY
    if (!pk || !n_e) throw Error("hg_lasso_num_challenges: null argument");
    // r: nu; collation rounds: nu; gamma, tau: 2; a grand product over 2^v entries: mu of layer 0, then per layer j = 1..v-1
    // one batching challenge, j round challenges and mu (lasso.rs:85-111, prover.rs:183-266)
    auto gp = [](size_t v) { size_t c = 1; for (size_t j = 1; j < v; j++) c += 2 + j; return c; };
    const size_t nu = (size_t)pk->lasso.nu;
    *n_e = nu + nu + 2 + gp(nu) + gp(16);
    return 0;
    HG_CATCH(-1)
}

int hg_sumcheck(hg_ctx* ctx, int kind, size_t nv, size_t ntab, const uint64_t* const* tables, const int* is_base, const uint64_t* pw,
                size_t npw, const uint64_t* claim2, size_t chain_skip, uint64_t* msgs, uint64_t* point, uint64_t* evals, uint64_t* sums) {
    HG_TRY
    if (!ctx || !tables || !is_base || !claim2 || (npw && !pw)) throw Error("hg_sumcheck: null argument (a HIP device is required)");
    if (ntab == 0 || nv == 0 || nv > 30) throw Error("hg_sumcheck: bad shape");
    SumcheckIO io;
    io.kind = kind; io.nv = nv; io.chain_skip = chain_skip;
    io.tables.assign(tables, tables + ntab);
    io.is_base.assign(is_base, is_base + ntab);
    for (size_t i = 0; i < npw; i++) io.pw.push_back(e2(pw[2 * i], pw[2 * i + 1]));
    io.claim = e2(claim2[0], claim2[1]);
    sumcheck_on_tables(ctx, io);
    auto put = [](uint64_t* dst, const std::vector<E2>& v) { if (dst) for (size_t i = 0; i < v.size(); i++) { dst[2 * i] = v[i].c0; dst[2 * i + 1] = v[i].c1; } };
    put(msgs, io.msgs); put(point, io.point); put(evals, io.evals); put(sums, io.sums);
    return 0;
    HG_CATCH(-1)
}

int hg_grand_product(hg_ctx* ctx, size_t nb, size_t len, const uint64_t* const* tables, size_t chain_skip, uint8_t* proof, size_t cap,
                     size_t* proof_len, uint64_t* claims2, uint64_t* point2) {
    HG_TRY
    if (!ctx || !tables || !proof || !proof_len) throw Error("hg_grand_product: null argument (a HIP device is required)");
    std::vector<E2> claims, point;
    std::vector<uint8_t> bytes = grand_product_on_tables(ctx, nb, len, tables, chain_skip, &claims, &point);
    *proof_len = bytes.size();
    if (bytes.size() > cap) throw Error("proof buffer too small");
    memcpy(proof, bytes.data(), bytes.size());
    if (claims2) for (size_t i = 0; i < claims.size(); i++) { claims2[2 * i] = claims[i].c0; claims2[2 * i + 1] = claims[i].c1; }
    if (point2) for (size_t i = 0; i < point.size(); i++) { point2[2 * i] = point[i].c0; point2[2 * i + 1] = point[i].c1; }
    return 0;
    HG_CATCH(-1)
}
int hg_fold(hg_ctx* ctx, const uint64_t* table, size_t nv, int is_base, const uint64_t r2[2], uint64_t* out) {
    HG_TRY
    if (!ctx || !table || !r2 || !out) throw Error("hg_fold: null argument (a HIP device is required)");
    fold_device(ctx, table, nv, is_base != 0, e2(r2[0], r2[1]), reinterpret_cast<E2*>(out));
    return 0;
    HG_CATCH(-1)
}
int hg_params_derive(uint32_t n, uint32_t k, const uint64_t* qis, uint64_t t, hg_params* out) {
    HG_TRY
    if (!qis || !out) throw Error("hg_params_derive: null argument");
    params_derive(n, k, qis, t, out);
    return 0;
    HG_CATCH(-1)
}

int hg_mle_eval(hg_ctx* ctx, const uint64_t* table, size_t nv, const uint64_t* point, uint64_t out2[2]) {
    HG_TRY
    if (!ctx || !table || (nv && !point) || !out2) throw Error("hg_mle_eval: null argument (a HIP device is required)");
    std::vector<E2> pt(nv);
    for (size_t i = 0; i < nv; i++) pt[i] = e2(point[2 * i], point[2 * i + 1]);
    E2 v = mle_eval_device(ctx, table, nv, pt.data());
    out2[0] = v.c0; out2[1] = v.c1;
    return 0;
    HG_CATCH(-1)
}

int hg_ntt(hg_ctx* ctx, const uint64_t* in, size_t log2n, int inverse, size_t batch, uint64_t* out) {
    HG_TRY
    if (!ctx || !in || !out) throw Error("hg_ntt: null argument (a HIP device is required)");
    if (log2n < 1 || log2n > 32) throw Error("hg_ntt: size out of range");
    ntt_device(ctx, in, (int)log2n, inverse != 0, batch, out);
    return 0;
    HG_CATCH(-1)
}

int hg_challenges(size_t n, uint64_t* out) {
    HG_TRY
    if (!out) throw Error("hg_challenges: null argument");
    const u64* c = challenge_chain(n);
    memcpy(out, c, n * 8);
    return 0;
    HG_CATCH(-1)
}

int hg_challenges_bn254(size_t n, uint64_t* out4) {
    HG_TRY
    hg::bn::challenges_bn254_raw(n, out4);
    return 0;
    HG_CATCH(-1)
}
int hg_bn254_field_op(hg_ctx* ctx, int op, size_t n, const uint64_t* a4, const uint64_t* b4, uint64_t* out4) {
    HG_TRY
    if (!ctx) throw hg::Error("hg_bn254_field_op: no context (a HIP device is required)");
    hg::bn::field_op_bn254(ctx, op, n, a4, b4, out4);
    return 0;
    HG_CATCH(-1)
}
int hg_sumcheck_bn254(hg_ctx* ctx, int kind, size_t nv, size_t ntab, const uint64_t* const* tables, const uint64_t* pw4, size_t npw,
                      const uint64_t* claim4, size_t chain_skip, uint64_t* msgs, uint64_t* point, uint64_t* evals, uint64_t* sums) {
    HG_TRY
    if (!ctx) throw hg::Error("hg_sumcheck_bn254: no context (a HIP device is required)");
    hg::bn::sumcheck_bn254(ctx, kind, nv, ntab, tables, pw4, npw, claim4, chain_skip, msgs, point, evals, sums);
    return 0;
    HG_CATCH(-1)
}

int hg_grand_product_bn254(hg_ctx* ctx, size_t nb, size_t len, const uint64_t* const* tables, size_t chain_skip, uint8_t* proof, size_t cap,
                           size_t* proof_len, uint64_t* claims4, uint64_t* point4) {
    HG_TRY
    if (!ctx) throw hg::Error("hg_grand_product_bn254: no context (a HIP device is required)");
    std::vector<uint8_t> bytes;
    ctx->arena_reset();
    hg::bn::grand_product_bn254(ctx, nb, len, tables, chain_skip, bytes, claims4, point4);
    *proof_len = bytes.size();
    if (bytes.size() > cap) throw hg::Error("proof buffer too small");
    memcpy(proof, bytes.data(), bytes.size());
    return 0;
    HG_CATCH(-1)
}
int hg_lasso_prove_bn254(hg_ctx* ctx, const hg_pk* pk, const uint64_t* lasso_in4, size_t chain_skip, uint8_t* proof, size_t cap, size_t* len,
                         uint64_t* claim_out4) {
    HG_TRY
    if (!ctx || !pk) throw hg::Error("hg_lasso_prove_bn254: null argument (a HIP device is required)");
    std::vector<uint8_t> bytes;
    ctx->arena_reset();
    hg::bn::lasso_prove_bn254(ctx, pk, lasso_in4, chain_skip, bytes, claim_out4);
    *len = bytes.size();
    if (bytes.size() > cap) throw hg::Error("proof buffer too small");
    memcpy(proof, bytes.data(), bytes.size());
    return 0;
    HG_CATCH(-1)
}
int hg_witness_from_json_bn254(const hg_params* params, const char* path, hg_witness** out) {
    HG_TRY
    if (!params || !path || !out) throw hg::Error("hg_witness_from_json_bn254: null argument");
    std::unique_ptr<hg_witness> w(new hg_witness());
    w->params = *params;
    w->w = hg::witness_from_json_bn254(hg::Params(*params), path);
    *out = w.release();
    return 0;
    HG_CATCH(-1)
}
int hg_circuit_eval_bn254(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, int which, uint64_t* out4, size_t cap_elems, size_t* n_elems) {
    HG_TRY
    if (!ctx || !pk || !w || !n_elems) throw hg::Error("hg_circuit_eval_bn254: null argument (a HIP device is required)");
    check_witness(pk, w, "hg_circuit_eval_bn254");
    std::vector<uint64_t> v;
    hg::bn::circuit_eval_bn254(ctx, pk, w->w, which, v);
    *n_elems = v.size() / 4;
    if (v.size() / 4 > cap_elems) throw hg::Error("hg_circuit_eval_bn254: output buffer too small");
    memcpy(out4, v.data(), v.size() * 8);
    return 0;
    HG_CATCH(-1)
}
int hg_prove_bn254(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, uint8_t* proof, size_t cap, size_t* len, double* ms2) {
    HG_TRY
    if (!ctx || !pk || !w || !len) throw hg::Error("hg_prove_bn254: null argument (a HIP device is required)");
    check_witness(pk, w, "hg_prove_bn254");
    std::vector<uint8_t> bytes;
    hg::bn::prove_bn254(ctx, pk, w->w, bytes, ms2);
    *len = bytes.size();
    if (bytes.size() > cap) throw hg::Error("proof buffer too small");
    memcpy(proof, bytes.data(), bytes.size());
    return 0;
    HG_CATCH(-1)
}
int hg_mle_eval_bn254(hg_ctx* ctx, const uint64_t* table4, size_t nv, const uint64_t* point4, uint64_t out4[4]) {
    HG_TRY
    if (!ctx) throw hg::Error("hg_mle_eval_bn254: no context (a HIP device is required)");
    hg::bn::mle_eval_bn254(ctx, table4, nv, point4, out4);
    return 0;
    HG_CATCH(-1)
}
int hg_ntt_bn254(hg_ctx* ctx, const uint64_t* in4, size_t log2n, int inverse, size_t batch, uint64_t* out4) {
    HG_TRY
    if (!ctx) throw hg::Error("hg_ntt_bn254: no context (a HIP device is required)");
    hg::bn::ntt_bn254(ctx, in4, (int)log2n, inverse != 0, batch, out4);
    return 0;
    HG_CATCH(-1)
}

int hg_profile(hg_ctx* ctx, int level) {
    if (!ctx) { g_last_error = "hg_profile: null context"; return -1; }
    ctx->prof_level = level;
    return 0;
}
int hg_profile_select(hg_ctx* ctx, const char* name) {
    if (!ctx || !name) { g_last_error = "hg_profile_select: null argument"; return -1; }
    bool found = false;
    for (auto& s : ctx->prof_stats) found = found || s.name == name;
    if (!found) { g_last_error = std::string("hg_profile_select: no kernel class named ") + name; return -1; }
    for (auto& s : ctx->prof_stats) s.dominant = s.name == name;
    return 0;
}
int hg_profile_reset(hg_ctx* ctx) {
    if (!ctx) { g_last_error = "hg_profile_reset: null context"; return -1; }
    for (auto& s : ctx->prof_stats) { s.launches = 0; s.ms = 0; s.bytes = 0; s.model = 0; s.design = 0; }
    return 0;
}
int hg_profile_get(hg_ctx* ctx, hg_kernel_stat* out, int cap) {
    if (!ctx || (!out && cap > 0)) { g_last_error = "hg_profile_get: null argument"; return -1; }
    int n = 0;
    for (auto& s : ctx->prof_stats) {
        if (n >= cap) break;
        memset(&out[n], 0, sizeof(out[n]));
        strncpy(out[n].name, s.name.c_str(), sizeof(out[n].name) - 1);
        out[n].launches = s.launches; out[n].total_ms = s.ms; out[n].algo_bytes = s.bytes; out[n].model_bytes = s.model; out[n].hbm_bytes = s.design;
        n++;
    }
    return n;
}

}  // extern "C"
