// Device-side proving engine: context (stream, arena, result buffer, challenge chain in HBM),
// prover key (circuit wiring as CSR in HBM) and the enqueue-everything / assemble-later prover.
//
// Because the reference transcript never absorbs prover messages (host.hpp), every challenge is
// known before the first launch. The prover therefore (1) walks the protocol once on the host,
// squeezing challenges in protocol order and enqueueing kernels whose scalar outputs (round sums,
// final evaluations) land in one result buffer in HBM, (2) copies that buffer back with a single
// synchronisation, (3) replays the recorded transcript steps on the host to interpolate round
// polynomials, chain the claims and emit the proof bytes. No device->host round trip per round.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "host.hpp"
#include "kernels.hpp"

struct hg_ctx {
    typedef hg::u64 u64;
    int device = 0;
    hipStream_t stream = nullptr;   // Lasso node + everything sequential
    hipStream_t bn_stream_hi = nullptr, bn_stream_lo = nullptr;   // BN254 prove only (bn254_gkr.inc: BnStreams), created on first use
    hipEvent_t bn_ev[2] = {nullptr, nullptr};                     // BN254 Lasso node: its claim / collation rounds on the second stream (bn254.hip)
    hipStream_t stream2 = nullptr;  // Vanilla / FFT node reductions (independent of the Lasso node on the device)
    hipStream_t stream_col = nullptr;   // one rank: the collation sum-check's short launches, then the node reductions, off the main stream (forked from and joined to it)
    hipEvent_t ev_col = nullptr;
    hg::E2* d_partials3 = nullptr;      // scratch of stream_col
    // the sums of the grand products' split rounds (prover_sumcheck.inc: flush_stride) run beside the rounds that follow them: forked from
    // the main stream behind the last fold-only launch (ev_sum[0]), joined to it at the end of the prove (ev_sum[1], sum_pending)
    hipStream_t stream_sum = nullptr;
    hipEvent_t ev_sum[2] = {nullptr, nullptr};
    hg::E2* d_partials4 = nullptr;      // scratch of stream_sum
    bool sum_pending = false;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_aux[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // Lasso node: limbs done (stream -> stream2), grand product #2 levels done (stream2 -> stream), E tables done (stream -> stream2), counters done (stream2 -> stream), opening tables done (stream2 -> stream)
    hipStream_t prof_stream = nullptr;  // stream the profiling events are recorded on (the one being enqueued to)
    // multi-GPU (comm.hip): RCCL communicator of this rank (ncclComm_t, type-erased: RCCL is loaded at run time), exchange buffer
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    u64* d_xchg = nullptr;
    size_t xchg_cap = 0;
    // cached launch graph of the resident prove (prover.hip: ProveCache). The launch sequence of a proof depends on the prover key
    // and on ADDRESSES only - every challenge is known up front, no kernel argument depends on the witness - so after two ordinary
    // proves of the same (key, values) pair the third is captured into a hipGraph and later ones replay it: the host's ~0.7 ms
    // protocol walk and ~200 launch calls collapse into one hipGraphLaunch. Any other use of the arena invalidates it.
    // Round 3: the cache holds SEVERAL graphs (one per (key, values object, share), least recently used evicted), each with a
    // PRIVATE arena, so a replay neither depends on nor is invalidated by anything else that uses the context's arena, and a
    // values object that is refilled in place (hg_witness_gen_into) keeps its graph: a new witness every prove replays.
    void* prove_cache = nullptr;    // prover.hip: ProveCacheSet
    void* pending_shard = nullptr;  // prover.hip: PendingShard of hg_prove_shard_begin .. _finish (owned; dropped with the context)
    uint64_t arena_epoch = 0;       // bumped by arena_reset()
    bool arena_fixed = false;       // alloc() must not grow the arena (a launch graph is being recorded into a private arena)
    // walked proves seen per (key serial, values serial, share): the third one of a values object is recorded into a graph
    struct WalkCount { uint64_t pk_serial, values_serial; int share; int walks; size_t arena_bytes; float gpu_ms; };
    std::vector<WalkCount> walk_counts;
    // hg_prove_stream: the next witness is uploaded and evaluated on a third stream into the OTHER of two table sets while the
    // current one is being proven
    hipStream_t stream3 = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr};
    hg_values* stream_values[2] = {nullptr, nullptr};
    uint64_t stream_values_serial = 0;     // key serial they were laid out for
    hg::u64* stream_pinned[2] = {nullptr, nullptr};   // pinned staging of one witness each (hg_prove_stream): host arrays are pageable
    size_t stream_pinned_words = 0;
    hg_values* scratch_values = nullptr;   // hg_prove's resident tables, refilled in place per call (so its launch graph survives)
    uint64_t scratch_serial = 0;           // key serial they were laid out for
    uint64_t no_graph_serial = 0;          // key whose graph capture failed: its proves walk (no retry)
    int no_graph_share = -1;
    // guard against a launch graph that replays slower than the plain launches it recorded (the runtime's stream assignment of a
    // graph's branches is not under the library's control): GPU time of the last walked prove, and the key a slow graph was seen for
    float last_walk_gpu_ms = 0;
    uint64_t slow_graph_serial = 0;
    int slow_graph_share = -1;
    bool use_graph = true;          // hg_set_option("graph", 0) / HG_NO_GRAPH=1 turn it off
    // options (hg_set_option)
    bool one_stream = false;  // keep every launch on `stream` (per-kernel timings without cross-stream interference)
    int mode = 0;             // protocol mode of the next proves: bit 0 absorbing transcript, bit 1 extension-field memory checking
    // bump arena: chunks are kept across proves, offsets reset per prove
    struct Chunk { char* p; size_t cap, used; size_t high = 0; };  // high: largest `used` since the last reset
    std::vector<Chunk> chunks;
    size_t arena_total = 0;
    void* alloc(size_t bytes);
    template <typename T> T* alloc_n(size_t n) { return static_cast<T*>(alloc(n * sizeof(T))); }
    void arena_reset();
    // nested scopes inside one prove (BN254 path): snapshot of the per-chunk offsets, restored when the scope's buffers are dead
    // (single stream: later users of the recycled bytes are ordered after the earlier ones)
    std::vector<size_t> arena_mark() const;
    void arena_rewind(const std::vector<size_t>& mark);
    // moves every chunk's offset to its high-water mark: what is allocated next aliases nothing handed out since the last reset,
    // released or not (needed when a second stream may still be working on released scopes)
    void arena_skip_to_high();
    size_t arena_high = 0;  // high-water mark since the last coalesce
    // challenge chain in HBM (as E2 pairs)
    hg::E2* d_chal = nullptr;
    size_t chal_e = 0;
    void ensure_chain(size_t n_e);
    // scalar results
    hg::E2* d_res = nullptr;
    hg::E2* h_res = nullptr;  // pinned
    size_t res_cap = 0;
    size_t res_hint = 0;            // result slots a sharded prove of key `res_hint_serial` uses (enqueue_prove)
    uint64_t res_hint_serial = 0;
    size_t bn_res_used = 0;  // bytes of the result buffer handed out to the BN254 path since the last arena_reset
    hg::E2* d_partials = nullptr;
    hg::E2* d_partials2 = nullptr;  // scratch of stream2
    // pinned staging for small host->device descriptor copies; bump-allocated, reset per prove
    char* h_stage = nullptr;
    void* h_mailbox = nullptr;   // pinned: the sequential prover's transcript mailbox + host-tail buffer (prover_seq.hip), allocated on first use
    size_t mailbox_bytes = 0;
    size_t stage_cap = 0, stage_used = 0;
    // BN254 path: device mirror of the staging buffer - descriptors are staged at the same offset on both sides and copied over in
    // ONE transfer per launch phase (bn254.hip: bn_stage / bn_flush) instead of one small copy per descriptor array
    char* bn_dstage = nullptr;
    size_t bn_flushed = 0;
    // profiling
    int prof_level = 0;
    struct ProfEvent { int cls; hipEvent_t a, b; };
    std::vector<ProfEvent> prof_events;
    std::vector<hipEvent_t> event_pool;
    struct ProfStat { std::string name; uint64_t launches = 0; double ms = 0, bytes = 0, model = 0, design = 0; bool dominant = false; };
    std::vector<ProfStat> prof_stats;
    int prof_class(const char* name, bool dominant);
    // bytes: what the launch streams by this implementation's algorithm; model_bytes (< 0: the same): the same work in the traffic
    // model of the reference's algorithm - they differ where an algebraic shortcut avoids tables
    // design_bytes (< 0: the same as bytes): what the launch moves to or from HBM by design (hg_kernel_stat::hbm_bytes)
    void prof_begin(int cls, double bytes, double model_bytes = -1.0, double design_bytes = -1.0);
    void prof_end();
    void prof_collect();
    int cur_cls = -1;
    hipEvent_t cur_a = nullptr;
    ~hg_ctx();
};

struct hg_witness {
    hg::Witness w;
    hg_params params;
};

struct hg_pk {
    uint64_t serial = 0;  // unique per hg_setup: identifies the key even if its address is reused after hg_pk_free
    hg_ctx* ctx = nullptr;
    hg::Params params;
    hg::LassoPlan lasso;
    hg::HCircuit circuit;
    hg::dev::LassoDev lasso_dev;
    // per vanilla node, per input: device CSRs
    struct NodeDev {
        std::vector<hg::dev::CsrLin> lin;    // [arity]
        std::vector<hg::dev::CsrMul> mulL;   // [arity] keyed by left operand
        std::vector<hg::dev::CsrMul> mulR;   // [arity] keyed by right operand
        // run-length form of lin + mulL per input (kernels.hpp GatherSeg), when the wiring is affine in the input position;
        // `alias`: the table IS a slice of the node's eq table (one unit-coefficient relay per position): nothing to build
        struct Seg { const hg::dev::GatherSeg* d = nullptr; int nseg = 0; bool alias = false; size_t alias_off = 0;
                     size_t win_lo = 0, win_hi = 0; };   // positions outside [win_lo, win_hi) have no term: the table is zero there
        std::vector<Seg> seg;                // [arity]
        // eq-factored form of Libra phase 1 (kernels.hpp PsJob::eq_n), found at setup (capi.hip): the table of every used input is a
        // constant times eq(z', .), z' = the first w coordinates of the claim point and the bits of hib above them
        struct EqForm { bool ok = false; int w = 0; unsigned hib = 0;
                        std::vector<std::vector<std::pair<hg::u64, hg::u32>>> terms;   // per input: (coefficient, gate block)
                        std::vector<std::pair<hg::u64, hg::u32>> consts; };            // additive constants per gate block: (value, block)
        EqForm eq_form;
        const hg::u32* const_gate = nullptr;
        const hg::u64* const_coef = nullptr;
        size_t nconst = 0;
        hg::dev::EvalNode fwd{};  // gate-major wiring for circuit evaluation (in/out pointers filled per run)
    };
    std::vector<NodeDev> node_dev;
    std::map<int, const hg::u64*> w_fwd, w_inv;  // log2n -> w^i table (N entries)
    std::vector<void*> owned;                    // device allocations released in hg_pk_free
    explicit hg_pk(const hg_params& p) : params(p) {}
};

struct hg_values {
    uint64_t serial = 0;                 // unique per object, kept across in-place refills (hg_witness_gen_into): launch-graph cache key
    uint64_t pk_serial = 0;              // the key whose circuit the tables are laid out for
    int device = 0;
    hg_ctx* ctx = nullptr;               // the context it was created on (told to drop its launch graphs when the object is freed)
    std::vector<const hg::u64*> d_vals;  // per node
    std::vector<size_t> sizes;
    const hg::u64* d_ct0is = nullptr;
    hg::u64* ntt_scratch = nullptr;      // kept: a refill allocates nothing
    size_t ct0is_len = 0;
    // evaluation plan (witness_fill): levels, FFT groups
    std::vector<int> level;
    int max_level = 0;
    std::vector<int> order;
    std::vector<void*> owned;
    // a rank's share of a sharded proof (witness_gen_shard): the tables of nodes it does not read are not resident (d_vals[id] == nullptr)
    int shard_rank = -1, shard_world = 0;   // -1: every table is resident
    size_t resident_bytes = 0, full_bytes = 0;
    // evaluation of a SUBSET of the circuit (witness_fill): mask[id] != 0 for the nodes that are laid out and evaluated (empty: all).
    // A rank's values object owns such a subset object for the cone of nodes its resident tables depend on (eval_cone): a refill
    // evaluates the cone into it and copies the resident tables over - nothing is allocated, the rest of the circuit is never touched.
    std::vector<char> mask;
    bool with_ct0is = true;
    hg_values* eval_cone = nullptr;
    size_t cone_bytes = 0;                 // bytes of eval_cone (tables + NTT scratch): resident_bytes + cone_bytes = the rank's peak
    // first result slot of this object's one-rank proves (hg_prove_stream: its second table set writes the upper half of the
    // result buffer, so that a proof can be replayed on the host while the next prove - of the other set - already runs)
    size_t res_base = 0, res_limit = 0;   // (res_limit != 0: the slots of a prove must end below it - the other table set's half)
};

namespace hg {

struct ProveResult {
    std::vector<uint8_t> proof;
    // prove_resident(.., borrow = true) from a cached launch graph: the bytes stay in the cached prover's buffer (valid until the next
    // prove of the context) and `proof` is empty - no 150 KB allocation (an mmap, its page faults and an munmap) and copy per proof
    const std::vector<uint8_t>* proof_ref = nullptr;
    const std::vector<uint8_t>& bytes() const { return proof_ref ? *proof_ref : proof; }
    double witness_ms = 0, upload_ms = 0, prove_ms = 0, gpu_ms = 0, enqueue_ms = 0, sync_ms = 0, replay_ms = 0;
};

// who owns what when ONE proof is sharded over `world` GPUs (prover.hip): Vanilla / FFT node reductions dealt whole, the Lasso
// node split by memory (memory-GKR index -> rank), the output claim's evaluation
struct ShardPlan { std::vector<int> node_owner; std::vector<int> gp1_mem_owner; int own_out_claim = 0; };
ShardPlan shard_plan(const hg_pk* pk, int rank, int world);
hg_values* witness_gen(hg_ctx* ctx, const hg_pk* pk, const Witness& w, double* witness_ms, double* upload_ms);
// Circuit::evaluate for ONE rank of a sharded proof: only the node tables the rank's share reads stay resident in HBM (the Lasso
// node's input, the inputs of the node reductions it owns, ct0is for the owner of the output claim); the others are released
hg_values* witness_gen_shard(hg_ctx* ctx, const hg_pk* pk, const Witness& w, int rank, int world, double* witness_ms, double* upload_ms);
// Circuit::evaluate into the tables of an existing values object (same addresses: its cached launch graph stays valid)
void witness_gen_into(hg_ctx* ctx, const hg_pk* pk, const Witness& w, hg_values* v, double* witness_ms, double* upload_ms);
void witness_gen_into_staged(hg_ctx* ctx, const hg_pk* pk, const Witness& w, hg_values* v, double* witness_ms, double* upload_ms);   // through the context's page-locked staging
double prove_warmup(hg_ctx* ctx, const hg_pk* pk);   // hg_warmup
// BfvEncrypt::prove for a run of witnesses, pipelined: upload + circuit.evaluate of witness i+1 overlap the GKR prove of witness i
std::vector<ProveResult> prove_stream(hg_ctx* ctx, const hg_pk* pk, const std::vector<const Witness*>& ws, double* total_ms);
void values_free(hg_values* v);
void pending_shard_drop(hg_ctx* ctx);
void ctx_register(hg_ctx* ctx, bool alive);
ProveResult prove_resident(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, bool borrow = false);
// the same in a protocol mode of SURVEY.md 8(f) f-4 (bit 0 absorbing transcript, bit 1 extension-field memory checking):
// round-by-round prover (prover_seq.hip); mode 0 = prove_resident
ProveResult prove_resident_mode(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode);
ProveResult prove_resident_mode_sharded(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode, int rank, int world,
                                        int (*reduce)(void*, uint64_t*, size_t), void* user);
size_t prove_shard_begin(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world);  // -> #E2 slots in ctx->h_res
// the whole sharded proof with the exchange inside the library (comm.hip): begin -> RCCL all-reduce on the stream -> replay
ProveResult prove_sharded(hg_ctx* ctx, const hg_pk* pk, const hg_values* v);
void comm_unique_id(uint8_t out[128]);
void comm_init(hg_ctx* ctx, const uint8_t id[128], int rank, int world);
void comm_destroy(hg_ctx* ctx);
int comm_count(hg_ctx* ctx);   // ncclCommCount of the context's communicator (0: none)
void comm_selftest(hg_ctx* ctx, const u64* bufs, int world, size_t n, u64* out);
void comm_allreduce_results(hg_ctx* ctx, size_t n_e2);
void prove_shard_combine(hg_ctx* ctx, const u64* gathered, int world, size_t n_u64);
ProveResult prove_shard_finish(hg_ctx* ctx);
void shard_combine_host(const u64* gathered, int world, size_t n_u64, u64* dst);
// Lasso node alone on a fresh transcript; claim_out = nu point coordinates then the value
std::vector<uint8_t> prove_lasso_node(hg_ctx* ctx, const hg_pk* pk, const u64* lasso_in_host, size_t chain_skip, std::vector<E2>* claim_out);
// one sum-check on caller tables (kernel-level parity entry point)
struct SumcheckIO {
    int kind; size_t nv; std::vector<const u64*> tables; std::vector<int> is_base; std::vector<E2> pw; E2 claim; size_t chain_skip;
    std::vector<E2> msgs, point, evals, sums;
};
void sumcheck_on_tables(hg_ctx* ctx, SumcheckIO& io);
E2 mle_eval_device(hg_ctx* ctx, const u64* table_host, size_t nv, const E2* point_host);
std::vector<uint8_t> grand_product_on_tables(hg_ctx* ctx, size_t nb, size_t len, const u64* const* tables, size_t chain_skip, std::vector<E2>* claims_out,
                                             std::vector<E2>* point_out);
void fold_device(hg_ctx* ctx, const u64* table_host, size_t nv, bool is_base, E2 r, E2* out_host);
void ntt_device(hg_ctx* ctx, const u64* in_host, int log2n, bool inverse, size_t batch, u64* out_host);

void hip_check(hipError_t e, const char* what);
// BfvEncrypt::verify with the table-sized work on the device (verifier_dev.hip); "" = accepted, else the rejection reason
std::string verify_proof_device(hg_ctx* ctx, const hg_pk* pk, const Witness& w, const uint8_t* proof, size_t len);
void prove_cache_drop(hg_ctx* ctx);

}  // namespace hg
