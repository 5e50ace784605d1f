/* hyper-greco-amd: C ABI of the MI355X-native GKR prover for the BFV secret-key-encryption circuit.
 *
 * This is the drop-in boundary: plain C, caller-owned buffers, `int` status
 * (0 = OK, negative = error, text via hg_last_error()). Field elements cross the boundary as
 * canonical little-endian u64 limbs (Goldilocks: 1 limb; GoldilocksExt2: 2 limbs [c0, c1]); proof
 * bytes use the reference wire format (canonical repr, big-endian, ext = bases in order)
 * [REF bfv-gkr/src/transcript.rs:183-195].
 *
 * Each entry point names the reference interface it replaces. The Rust-side binding a maintainer
 * would add is shown in INTEGRATION.md.
 */
#ifndef HG_H
#define HG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_MAX_K 16

typedef struct hg_ctx hg_ctx;         /* one per GPU: stream, workspace arena, cached challenge chain */
typedef struct hg_pk hg_pk;           /* prover key = LassoPreprocessing + circuit wiring, device resident */
typedef struct hg_witness hg_witness; /* BfvSkEncryptArgs after get_inputs(): laid-out field tables (host) */
typedef struct hg_values hg_values;   /* circuit.evaluate() result: every node's table, resident in HBM */
typedef struct hg_group hg_group;     /* the ranks of a sharded round-by-round prove: who adds the partial round sums */

/* Per-parameter-set constants [REF bfv-gkr/src/constants/mod.rs:16-35, constants/sk_enc_constants_*.rs] */
typedef struct hg_params {
    uint32_t n;        /* ring degree N */
    uint32_t k;        /* number of CRT moduli (const generic K) */
    uint64_t s_bound, e_bound, k1_bound;
    uint64_t r1_bounds[HG_MAX_K], r2_bounds[HG_MAX_K];
    uint64_t qis[HG_MAX_K], k0is[HG_MAX_K];
} hg_params;

typedef struct hg_timings {
    double witness_ms; /* "wintess gen" span: circuit.evaluate + output claim [REF sk_encryption_circuit.rs:439-453] */
    double upload_ms;  /* host -> HBM copy of node values (not part of any reference span) */
    double prove_ms;   /* "GKR prove" span [REF sk_encryption_circuit.rs:455-457]: HIP events on the prover stream,
                          node values resident in HBM at start; wall clock from the
                          first launch to the assembled proof bytes (includes host launch + transcript replay) */
    double gpu_ms;     /* the same span measured with HIP events on the prover stream (device time only) */
    double total_ms;   /* wall clock of the whole hg_prove call */
    double enqueue_ms; /* host time spent walking the protocol and launching (part of prove_ms) */
    double sync_ms;    /* host wait for the stream after the last launch (part of prove_ms) */
    double replay_ms;  /* host transcript replay: interpolation, claim chaining, proof bytes (part of prove_ms) */
} hg_timings;

/* Per-kernel-class profile (HIP events recorded on the prover stream around every launch of the class). */
typedef struct hg_kernel_stat {
    char name[48];
    uint64_t launches;
    double total_ms;
    double algo_bytes; /* algorithmic bytes over all launches: tables read once + folded tables written once */
    double model_bytes; /* the same launches in the traffic model of the REFERENCE's algorithm: larger than algo_bytes
                           where an algebraic shortcut avoids tables (mirrored grand product, two-table collation sum-check) */
    double hbm_bytes;   /* what the launches move to or from HBM BY DESIGN: every table a launch reads and every table it writes, once -
                           nothing for tables that are recomputed and never stored (the hash rows of the first grand-product round), for
                           the intermediate folds of a two-round launch, or for rounds that run inside LDS */
} hg_kernel_stat;

const char* hg_last_error(void);
int hg_device_count(void);

/* = nothing in the reference (single-process rayon); one ctx per device, calls on a ctx are serialised by the caller */
hg_ctx* hg_create(int device_id);
void hg_destroy(hg_ctx* ctx);

/* Context options (nothing in the reference: its only knob is the rayon pool). name:
 *   "one_stream"  value != 0: every launch on one stream (the default overlaps the Vanilla / FFT node reductions, the counter
 *                 sorts and the openings with the Lasso node's critical path on a second stream); used to time kernels in isolation
 *   "graph"       value == 0: never replay a cached launch graph (default: after two ordinary resident proves of the same key and
 *                 values object the third is captured into a hipGraph and later ones replay it - the launch sequence depends on
 *                 addresses only, because every challenge is known up front; a values object refilled by hg_witness_gen_into
 *                 keeps its graph; up to HG_GRAPH_ENTRIES (8) graphs per context, each with a private workspace)
 * Returns 0, or -1 for an unknown name. */
int hg_set_option(hg_ctx* ctx, const char* name, int64_t value);

/* = `type Params = constants::SkEnc{N}_{K}x{bits}_65537` [REF bfv-gkr/src/test.rs:8] */
int hg_params_builtin(uint32_t n, uint32_t k, hg_params* out);

/* = the constants emitter of scripts/circuit_sk.py [REF scripts/circuit_sk.py:80, 249, 296-297, 334-337, 422-439]: a parameter set
 *   for ring degree n, k CRT moduli qis[] and plaintext modulus t, so that (n, k) can be swept beyond the six shipped sets.
 *   S_BOUND = 1, E_BOUND = 19, K1_BOUND = (t-1)/2, K0_i = (-t)^-1 mod q_i, R2_BOUND_i = int((q_i - 1) / 2) and
 *   R1_BOUND_i = int((int((q_i-1)/2) (n+2) + 19 + int((t-1)/2) K0_i) / q_i), with the script's float semantics of "/" reproduced
 *   (that is why shipped R2 bounds are doubles rounded to 53 bits). n must be a power of two, k one of 1, 2, 4, 8, 16. */
int hg_params_derive(uint32_t n, uint32_t k, const uint64_t* qis, uint64_t t, hg_params* out);

/* = BfvEncrypt::setup -> LassoPreprocessing::preprocess::<4, 65536> [REF sk_encryption_circuit.rs:319-349, lasso.rs:527-627]
 *   plus BfvEncrypt::configure (circuit wiring) [REF sk_encryption_circuit.rs:351-363, 86-293], done once. */
int hg_setup(hg_ctx* ctx, const hg_params* params, hg_pk** pk); /* ctx == NULL: host-only key (layout / circuit_eval) */
void hg_pk_free(hg_pk* pk);
/* Lasso memory map as text "subtable@dim,...|lookup:bits:m/m;..." (for tests) */
int hg_pk_lasso_layout(const hg_pk* pk, char* out, size_t cap);
/* [nu, num_nodes, rows, alpha, NodeId of lasso_inputs_batched, NodeId of sum] */
int hg_pk_info(const hg_pk* pk, uint64_t out[6]);
/* How hg_setup classified one node of the circuit (for tests; host-only keys too): [kind (0 input, 1 Vanilla, 2 FFT, 3 Lasso),
 * eq-factored form found (every Libra phase-1 table of the node is a constant times an eq table: VanillaNode wirings that relay aligned
 * blocks, sk_encryption_circuit.rs:97-285), log2 of the relayed block, index of the input window, number of (coefficient, gate block)
 * terms, log2 of the input size]. Whether a prove uses the form also depends on the node having ONE claim and on its size. */
int hg_pk_node_eq_form(const hg_pk* pk, int node, int64_t out[6]);

/* = serde_json::from_str::<BfvSkEncryptArgs> + BfvEncrypt::get_inputs / Poly::{new,new_padded,new_shifted}
 *   [REF bfv-gkr/src/test.rs:21-33, sk_encryption_circuit.rs:365-415, poly.rs:12-44] */
int hg_witness_from_json(const hg_params* params, const char* path, hg_witness** w);
/* Replaces scripts/circuit_sk.py (offline witness generator) with a seeded synthetic BFV sk-encryption
 * [REF scripts/circuit_sk.py:18-140, scripts/utils.py:4-18]; needed because the n=32768 fixture is a missing blob. */
int hg_witness_synthetic(const hg_params* params, uint64_t seed, hg_witness** w);
/* Already laid-out tables: s,e,k1: 2^L; ais,r1is: k*2^L; r2is: k*2^P; ct0is: k*2^L (L = log2 n + 1, P = log2 n) */
int hg_witness_from_arrays(const hg_params* params, const uint64_t* s, const uint64_t* e, const uint64_t* k1,
                           const uint64_t* ais, const uint64_t* r1is, const uint64_t* r2is, const uint64_t* ct0is,
                           hg_witness** w);
/* which: 0 s, 1 e, 2 k1, 3 ais, 4 r1is, 5 r2is, 6 ct0is. Returns the element count (copies min(count, cap)). */
int64_t hg_witness_get(const hg_witness* w, int which, uint64_t* out, size_t cap);
void hg_witness_free(hg_witness* w);

/* = BfvEncrypt::prove [REF sk_encryption_circuit.rs:417-460]. Fails (never falls back to a CPU path)
 *   when no HIP device is available. */
int hg_prove(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, uint8_t* proof, size_t cap, size_t* len,
             hg_timings* timings);

/* Makes the FIRST hg_prove of a key a warm one - what a drop-in BfvEncrypt::setup calls right after hg_setup [REF the caller's sequence
 *   bfv-gkr/src/test.rs:31-44: setup, then ONE prove per witness]: allocates the context-owned node tables and the page-locked witness
 *   staging, proves an all-zero witness until the launch graph of those tables is recorded (two protocol walks and the capture; nothing
 *   of it depends on table contents), so that the first real hg_prove refills the tables in place and replays the graph. *ms (may be
 *   NULL): the time it took. Optional: without it the first two hg_prove calls of a key walk the protocol and the third records. */
int hg_warmup(hg_ctx* ctx, const hg_pk* pk, double* ms);

/* hg_prove for a run of `n` witnesses under one key, pipelined [REF: the loop a caller of BfvEncrypt::prove writes; the only
 *   in-tree caller is the test macro bfv-gkr/src/test.rs:31-44, one witness per call - the reference has no batch entry]: upload + circuit.evaluate of witness i+1 run on a third stream into a second set of node
 *   tables while witness i is proven. Proof i is written at proofs + i*cap_each, its length to lens[i]; each proof is byte-identical
 *   to hg_prove's for that witness. timings (may be NULL): total_ms = wall clock of the whole run, prove_ms / gpu_ms = sums over
 *   the proofs. The first proofs of a context + key walk the protocol (the launch graph of a table set is recorded on its third
 *   prove); after that a proof costs about max(prove, upload + evaluate). */
int hg_prove_stream(hg_ctx* ctx, const hg_pk* pk, const hg_witness* const* ws, size_t n, uint8_t* proofs, size_t cap_each,
                    size_t* lens, hg_timings* timings);

/* = BfvEncrypt::verify [REF sk_encryption_circuit.rs:462-517] (host-side, like the reference's): the witness handle
 *   supplies the public inputs and ct0is. Returns 0 = accepted, 1 = rejected (reason via hg_last_error), < 0 = error.
 *   Works with a host-only key (hg_setup(NULL, ..)).
 *   NOTE: "accepted" means what the reference's verifier means, which is NOT soundness: the challenges are a fixed Keccak chain
 *   independent of the proof bytes, gamma / tau are truncated to one base limb, the collation sum-check's final evaluation and
 *   the multiset relation init * write == read * final between the two grand products are never checked, trailing bytes are
 *   ignored. hg_verify_mode(.., 3, ..) closes the first two. */
int hg_verify(const hg_pk* pk, const hg_witness* w, const uint8_t* proof, size_t len);
/* The same check with the table-sized work on the device [REF sk_encryption_circuit.rs:462-517; lasso/src/memory_checking/verifier.rs:
 * 130-176]: the host parses the proof and checks the round polynomials and the Lasso scalars; the eq tables, the wiring-predicate
 * sums of the Vanilla nodes, the DFT rows of the FFT nodes and the MLE evaluations of the public inputs run as kernels (one stream,
 * one synchronisation). Same return values and the same accept / reject decisions as hg_verify; Goldilocks, mode 0. */
int hg_verify_device(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, const uint8_t* proof, size_t len);

/* The same pair in a protocol mode that FIXES the reference's two known soundness gaps. mode bits:
 *   1  absorbing transcript: write_felt / read_felt also hash the element - the rule of the in-tree plonkish-trait writer of
 *      the same struct [REF bfv-gkr/src/transcript.rs:205-208, 224-233]; the gkr-trait writer the prover uses does not
 *      [REF :146-157, 180-196], which leaves every challenge independent of the proof;
 *   2  extension-field memory checking: gamma, tau are used as E elements, not truncated to base limb 0
 *      [REF lasso/src/memory_checking/prover.rs:36-39; README.md:108 "Known issues"].
 * mode 0 = hg_prove / hg_verify (the reference as it is, bit-exact). Other modes are Goldilocks only and run the round-by-round
 * prover: the device hands every round's sums to the host transcript through a pinned mailbox and spins on the challenge (no
 * stream synchronisation inside a sum-check); timings->sync_ms then holds the NUMBER of stream synchronisations and
 * timings->enqueue_ms the number of mailbox round trips. HG_SEQ_NO_MAIL=1 restores one synchronisation per round. */
int hg_prove_mode(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, int mode, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings);
int hg_prove_resident_mode(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings);
int hg_verify_mode(const hg_pk* pk, const hg_witness* w, int mode, const uint8_t* proof, size_t len);

/* The round-by-round prover (modes 1-3) on several ranks, with ONE all-reduce per sum-check round: the exchange pattern an absorbing
 * transcript leaves [REF bfv-gkr/src/transcript.rs:205-208, 224-233: every round's message is hashed before the next challenge is
 * squeezed], SURVEY 8(e)'s conservative form of the north_star partition. Every rank holds the whole witness (hg_witness_gen) and runs
 * the whole protocol; inside every round kernel rank r evaluates the hypercube sums of the tiles t = r (mod world) only - the
 * prove_sum_check work of [REF lasso/src/lasso.rs:278-279; memory_checking/prover.rs:242-252] split along the hypercube - and folds
 * everything; the group adds the ranks' partial sums (at most six canonical Goldilocks words, lane-wise mod p), after which every
 * rank's transcript absorbs the same message and squeezes the same challenge. Rounds finished on the host and the scalar steps between
 * sum-checks are replicated. Every rank returns the same proof, byte for byte the one hg_prove_resident_mode gives.
 *   hg_group_local(world)    ranks are threads of this process, one context each (any devices): barrier + modular sum in memory;
 *   hg_group_external(fn, user, world)   fn = int (*)(void* user, uint64_t* words, size_t n): adds `words` over the ranks in place
 *                            (e.g. an all-gather over torch.distributed / MPI followed by the modular sum) and returns 0.
 * In this form a round kernel is launched only after the challenge of the round before has been posted (the single-rank prover
 * launches ahead and lets the kernel wait on the device): no kernel ever waits for the host, so ranks may share a device or a
 * hardware queue (ranks as threads of one process in the tests) without waiting for each other's waiting kernels.
 * timings->replay_ms holds the number of all-reduces of the proof. Mode 0 shards through hg_prove_sharded (one all-reduce per proof).
 * WITHOUT replicating the witness (round 6): when `v` is a rank's share (hg_witness_gen_shard: the Lasso node's input and the inputs of
 * the node reductions the rank owns, chains dealt by CRT modulus as in mode 0), a Vanilla / FFT node's reduction runs on its owner
 * alone - every tile of its round kernels, the rounds finished on the host - and the other ranks launch nothing for it: they join the
 * same all-reduces with zeros (one per device round; one of up to 144 words for the rounds the owner's host finished; one for the final
 * evaluations), absorb the same messages and squeeze the same challenges. The Lasso node keeps the tile-split form. Same proof bytes. */
hg_group* hg_group_local(int world);
hg_group* hg_group_external(void* reduce_fn, void* user, int world);
void hg_group_free(hg_group* g);
int hg_prove_resident_mode_sharded(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int mode, int rank, hg_group* group, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings);

/* The two halves of hg_prove, split where the reference splits its spans:
 *   hg_witness_gen   = "wintess gen": circuit.evaluate(inputs) [REF sk_encryption_circuit.rs:439-442] ON THE DEVICE: the
 *                      3+2k+1 input tables are uploaded, the 2k+1 size-2^L NTTs (FFT -> pointwise mul -> IFFT) and
 *                      the Vanilla gate maps run as HIP kernels; every node table stays resident in HBM;
 *   hg_prove_resident = "eval output" + "GKR prove" [REF sk_encryption_circuit.rs:444-457] on resident tables.
 * bench.py times hg_prove_resident (inputs already in HBM when the timed region starts). */
int hg_witness_gen(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, hg_values** out, hg_timings* timings);
/* The same into an EXISTING values object (of the same key and context): no allocation, every table keeps its address. This is the
 * steady state of a prover that receives a new witness per proof, as the reference's caller does [REF bfv-gkr/src/test.rs:37-38]: the
 * launch graph the library recorded for `v` (hg_set_option "graph") stays valid, because the launch sequence of a prove depends on
 * addresses only - the next hg_prove_resident(v) replays it on the new witness. hg_prove does this internally with a values
 * object owned by the context. */
int hg_witness_gen_into(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, hg_values* v, hg_timings* timings);
/* The same for ONE rank of a proof sharded over `world` GPUs (BASELINE config 4): only the node tables the rank's share reads stay
 * resident - the Lasso node's input, the inputs of the Vanilla / FFT node reductions the planner deals to it, ct0is on the rank that
 * evaluates the output claim; the per-modulus objects a rank does not own [REF sk_encryption_circuit.rs:122-128, 245-260] are released.
 * Only the cone of those tables is ever evaluated (the per-modulus chains of the moduli the rank owns, not the whole circuit), into
 * subset tables the object keeps, so hg_witness_gen_into refills it without allocating (hg_values_peak_bytes: tables + cone).
 * The result proves through hg_prove_sharded / hg_prove_shard_begin with the same (rank, world) only. */
int hg_witness_gen_shard(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, int rank, int world, hg_values** out, hg_timings* timings);
/* [resident bytes, bytes of the full set of node tables, resident tables, tables] */
int hg_values_info(const hg_values* v, uint64_t out[4]);
/* Device bytes the object holds in all: its resident tables plus, for a rank's share (hg_witness_gen_shard), the subset tables its
 * refills evaluate into - the cone of nodes the resident tables are computed from, never the whole circuit. -1 for a null handle. */
int64_t hg_values_peak_bytes(const hg_values* v);
void hg_values_free(hg_values* v);
/* copies node `node`'s table (NodeId order of configure) back to the host; returns its element count */
int64_t hg_values_get(hg_ctx* ctx, const hg_values* v, int node, uint64_t* out, size_t cap);
int hg_prove_resident(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, uint8_t* proof, size_t cap, size_t* len,
                      hg_timings* timings);

/* ONE proof sharded over `world` GPUs (one process per GPU, same witness resident on each). Every rank walks the whole
 * protocol but runs only the device jobs it owns; the largest part, grand product #1 of the Lasso node, is split by
 * memory (batch item), so its round sums are PARTIAL sums on every rank. `*partial` (pinned host memory owned by the
 * context, *n_u64 lanes of canonical field elements) holds this rank's share of the scalar results, zeros elsewhere.
 * The caller all-gathers the ranks' buffers (RCCL) and hands them to hg_prove_shard_combine, which installs their
 * lane-wise sum mod p; hg_prove_shard_finish then replays the transcript — every rank obtains the identical bytes.
 * One all-gather per proof is the only exchange. world == 1 degenerates to hg_prove_resident. */
int hg_prove_shard_begin(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, int rank, int world, uint64_t** partial,
                         size_t* n_u64);
int hg_prove_shard_combine(hg_ctx* ctx, const uint64_t* gathered /* world x n_u64, rank-major */, int world, size_t n_u64);
int hg_prove_shard_finish(hg_ctx* ctx, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings);
/* the arithmetic of hg_prove_shard_combine alone, host only (no context, no device): out[i] = sum over ranks of gathered[r][i] mod p;
 * non-canonical lanes are an error. What a caller-side exchange (any byte transport) computes between _begin and _finish. */
int hg_shard_combine_host(const uint64_t* gathered /* world x n_u64, rank-major */, int world, size_t n_u64, uint64_t* out);

/* The same with the exchange INSIDE the library: one process per GPU, each with its own context and a copy of the resident
 * witness; one RCCL all-reduce per proof (each 64-bit lane as two 32-bit halves in 64-bit lanes, ncclSum, folded back mod p on
 * the device; no host staging), enqueued on the prover stream behind the rank's last kernel. Nothing in the reference
 * corresponds (single-process rayon); BASELINE config 4.
 *   hg_comm_unique_id: rank 0 obtains the 128-byte RCCL id and hands it to the other ranks out of band (any byte channel);
 *   hg_comm_init:      collective over all `world` ranks (ncclCommInitRank); world == 1 is allowed (single-rank communicator);
 *   hg_prove_sharded:  collective; every rank returns the identical proof bytes. */
int hg_comm_unique_id(uint8_t out[128]);
int hg_comm_init(hg_ctx* ctx, const uint8_t id[128], int rank, int world);
int hg_comm_destroy(hg_ctx* ctx);
/* number of ranks of the context's communicator as RCCL reports it (ncclCommCount); 0 without a communicator */
int hg_comm_count(hg_ctx* ctx, int* ranks);
/* The arithmetic of the exchange without a communicator (a one-GPU box cannot form one of more than one rank): `world` rank
 * buffers of n_u64 canonical lanes (host, rank-major) are split into 32-bit halves on the device, added lane-wise as plain 64-bit
 * integers - what ncclAllReduce(ncclUint64, ncclSum) does - and folded back mod p; out[n_u64] = lane-wise sum mod p. */
int hg_comm_selftest(hg_ctx* ctx, const uint64_t* rank_buffers, int world, size_t n_u64, uint64_t* out);
int hg_prove_sharded(hg_ctx* ctx, const hg_pk* pk, const hg_values* v, uint8_t* proof, size_t cap, size_t* len, hg_timings* timings);

/* = circuit.evaluate (host part of witness generation) [REF sk_encryption_circuit.rs:442]:
 *   copies out the Lasso node's input table (2^nu) and the `sum` node output (k*2^L). Host only. */
int hg_circuit_eval(const hg_pk* pk, const hg_witness* w, uint64_t* lasso_in, size_t lasso_cap, uint64_t* sum_out,
                    size_t sum_cap);

/* = <LassoNode as gkr::circuit::node::Node>::prove_claim_reduction [REF lasso/src/lasso.rs:57-114] on a fresh
 *   transcript. lasso_in: host table of 2^nu field elements. claim_out: nu E coordinates then the value. */
int hg_lasso_prove(hg_ctx* ctx, const hg_pk* pk, const uint64_t* lasso_in, uint8_t* proof, size_t cap, size_t* len,
                   uint64_t* claim_out);
/* The same, entered inside a larger transcript: `chain_skip` E challenges have already been squeezed by the caller
 * (what a Rust `impl Node` shim passes: the position of its `&mut dyn TranscriptWrite` in the challenge chain). */
int hg_lasso_prove_at(hg_ctx* ctx, const hg_pk* pk, const uint64_t* lasso_in, size_t chain_skip, uint8_t* proof, size_t cap,
                      size_t* len, uint64_t* claim_out);

/* Number of E challenges the Lasso node squeezes (nu for r, nu collation rounds, gamma / tau, both grand products:
 * SURVEY.md appendix C). A caller that replays the node's bytes into its own transcript (the Rust `impl Node` shim)
 * advances its challenge position by this much. */
int hg_lasso_num_challenges(const hg_pk* pk, size_t* n_e);

/* = gkr::sum_check::prove_sum_check [REF call sites lasso.rs:278-279, prover.rs:242-252] on caller tables.
 *   kind: 0 collation g = p0*sum M^i p_i, 1 grand product g = p0*sum gam^i p_2i p_2i+1, 2 sum of pair products.
 *   tables[i]: host pointer, 2^nv u64 (is_base) or 2^nv (c0,c1) pairs. The challenge chain starts after
 *   `chain_skip` E challenges. Outputs: msgs nv*(d+1) E, point nv E, evals ntab E, sums nv*d E (raw per-round sums). */
int hg_sumcheck(hg_ctx* ctx, int kind, size_t nv, size_t ntab, const uint64_t* const* tables, const int* is_base,
                const uint64_t* pw, size_t npw, const uint64_t* claim2, size_t chain_skip, uint64_t* msgs,
                uint64_t* point, uint64_t* evals, uint64_t* sums);

/* = prove_grand_product [REF lasso/src/memory_checking/prover.rs:183-266] on nb host tables of len = 2^nv base-field values: product
 *   tree on the MSB split, root products, per layer the batched degree-3 sum-check, 2 nb evaluations and the mu fold. The transcript
 *   starts after `chain_skip` E challenges. claims2: nb final claims (E), point2: nv coordinates (E). (Goldilocks counterpart of
 *   hg_grand_product_bn254) */
int hg_grand_product(hg_ctx* ctx, size_t nb, size_t len, const uint64_t* const* tables, size_t chain_skip, uint8_t* proof, size_t cap,
                     size_t* proof_len, uint64_t* claims2, uint64_t* point2);
/* = BoxMultilinearPoly::fix_var on the lowest variable (inside gkr::sum_check::prove_sum_check; SURVEY.md 8(c) convention C3):
 *   out[j] = T[2j] + r (T[2j+1] - T[2j]), j < 2^(nv-1). table: 2^nv base values (is_base) or (c0, c1) pairs; out: 2^(nv-1) pairs. */
int hg_fold(hg_ctx* ctx, const uint64_t* table, size_t nv, int is_base, const uint64_t r2[2], uint64_t* out);

/* = BoxMultilinearPoly::evaluate [REF call sites memory_checking/mod.rs:80-93, sk_encryption_circuit.rs:446]
 *   on a host table of 2^nv base-field values at an E point. */
int hg_mle_eval(hg_ctx* ctx, const uint64_t* table, size_t nv, const uint64_t* point, uint64_t out2[2]);

/* = FftNode evaluate (size-2^log2n NTT, natural order in/out) [REF sk_encryption_circuit.rs:224,249,251]. Device. */
int hg_ntt(hg_ctx* ctx, const uint64_t* in, size_t log2n, int inverse, size_t batch, uint64_t* out);

/* Fiat-Shamir challenge chain [REF bfv-gkr/src/transcript.rs:146-157,198-203]: first n base-field challenges. */
int hg_challenges(size_t n, uint64_t* out);

/* ---- BN254 (BASELINE config 5): the same path over halo2curves bn256::Fr (F = E = Fr) ---------------------------------
 * Elements cross the boundary as 4 canonical little-endian u64 limbs (non-Montgomery). The extension field of the
 * reference's bn254 tests is the field itself [REF sk_encryption_circuit.rs:614-626: (Fr, Fr)], so a challenge is one
 * element. hg_prove_bn254 is the whole BfvEncrypt::prove over Fr; the entry points before it expose its parts for parity tests. */
/* = Keccak256Transcript::squeeze_challenge over Fr: c_j = LE(Keccak^j("")) mod r [REF transcript.rs:146-157,198-203]; n x 4 limbs */
int hg_challenges_bn254(size_t n, uint64_t* out4);
/* device field arithmetic on n element pairs: op 0 add, 1 sub, 2 mul, 3 mul through the column accumulators, 4 a b + a a + b b
 * through one deferred reduction (known-answer tests of the Montgomery kernels; operands and results canonical, converted inside).
 * ops 5 .. 9: the branch-free loose forms the hot kernels use (bn254_lazy.hpp) on RAW 256-bit operands, results as canonical
 * residues: 5 a b R^-1 (any operands), 6 a + r b R^-1 with r = 2^200 + 12345 (a < 2p), 7 a + b, 8 a - b (a, b < 2p),
 * 9 (a b + a a + b b + (a - b + 2p) b) R^-1 through one reduction (a, b < 2p); R = 2^256. */
int hg_bn254_field_op(hg_ctx* ctx, int op, size_t n, const uint64_t* a4, const uint64_t* b4, uint64_t* out4);
/* = gkr::sum_check::prove_sum_check over Fr on caller tables, same shapes and conventions as hg_sumcheck
 *   [REF call sites lasso.rs:278-279, prover.rs:242-252]. tables[i]: host pointer, 2^nv elements (4 limbs each).
 *   Outputs: msgs nv*(d+1), point nv, evals ntab, sums nv*d elements. */
int hg_sumcheck_bn254(hg_ctx* ctx, int kind, size_t nv, size_t ntab, const uint64_t* const* tables, const uint64_t* pw4, size_t npw,
                      const uint64_t* claim4, size_t chain_skip, uint64_t* msgs, uint64_t* point, uint64_t* evals, uint64_t* sums);

/* = prove_grand_product over Fr [REF lasso/src/memory_checking/prover.rs:183-266]: nb tables of len = 2^nv elements; product tree on
 *   the MSB split, root products, per layer a degree-3 sum-check with g = poly(0) * sum_b gamma^b v_l,b v_r,b, 2 nb evaluations and the
 *   mu fold. proof: 32-byte big-endian canonical elements [REF transcript.rs:183-189]. claims4: nb final claims, point4: nv coordinates. */
int hg_grand_product_bn254(hg_ctx* ctx, size_t nb, size_t len, const uint64_t* const* tables, size_t chain_skip, uint8_t* proof, size_t cap,
                           size_t* proof_len, uint64_t* claims4, uint64_t* point4);
/* = <LassoNode as Node>::prove_claim_reduction over Fr [REF lasso/src/lasso.rs:57-114]: the same node as hg_lasso_prove_at, for the
 *   bn254 test family. lasso_in4: 2^nu elements (4 limbs each; range-shifted values, i.e. below 2^64). The limb split and the
 *   counters are integer kernels shared with the Goldilocks path; claimed sum, collation sum-check, multiset hashes, both grand
 *   products and the openings run over Fr. proof: 32-byte big-endian elements. claim_out4: nu coordinates of r, then the claimed sum. */
int hg_lasso_prove_bn254(hg_ctx* ctx, const hg_pk* pk, const uint64_t* lasso_in4, size_t chain_skip, uint8_t* proof, size_t cap, size_t* len,
                         uint64_t* claim_out4);
/* = BoxMultilinearPoly::evaluate over Fr [REF memory_checking/mod.rs:80-93]: table of 2^nv elements at a point of nv elements */
int hg_mle_eval_bn254(hg_ctx* ctx, const uint64_t* table4, size_t nv, const uint64_t* point4, uint64_t out4[4]);
/* = FftNode evaluate over Fr [REF sk_encryption_circuit.rs:224,249,251]: size-2^log2n NTT with the root of unity
 *   7^((r-1)/2^log2n) (halo2curves ROOT_OF_UNITY, two-adicity 28), natural order in / out; inverse scales by 1/n */
int hg_ntt_bn254(hg_ctx* ctx, const uint64_t* in4, size_t log2n, int inverse, size_t batch, uint64_t* out4);

/* = BfvSkEncryptArgs from one of the reference's bn254 fixtures [REF bfv-gkr/src/data/bn254/ *.json, sk_encryption_circuit.rs:365-415]:
 *   coefficients are bn256::Fr elements (negatives as r - |z|). Every coefficient of a valid witness is a small signed integer, which
 *   is what the handle stores (an element that is not an integer below 2^62 in magnitude is rejected); the same handle type as
 *   hg_witness_from_json / hg_witness_synthetic, so those witnesses can be proven over Fr as well. */
int hg_witness_from_json_bn254(const hg_params* params, const char* path, hg_witness** out);
/* = Circuit::evaluate over Fr on the device [REF sk_encryption_circuit.rs:434-442]; returns one table as canonical limbs:
 *   which = 0 the `sum` node (must equal ct0is), 1 the Lasso input node, 2 the ct0is table as laid out by get_inputs */
int hg_circuit_eval_bn254(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, int which, uint64_t* out4, size_t cap_elems, size_t* n_elems);
/* = BfvEncrypt::<_, 1>::prove::<Fr, Fr> [REF sk_encryption_circuit.rs:417-460, 614-626]: witness generation, output claim, the
 *   prove_gkr walk (Libra / zkCNN / Lasso node reductions) over bn256::Fr. proof: 32-byte big-endian elements
 *   [REF transcript.rs:183-189]. ms2 (may be null): witness generation and proving wall time in ms. */
int hg_prove_bn254(hg_ctx* ctx, const hg_pk* pk, const hg_witness* w, uint8_t* proof, size_t cap, size_t* len, double* ms2);
/* = BfvEncrypt::verify::<Fr, Fr> [REF sk_encryption_circuit.rs:462-517]: host side (no device needed; pk may come from
 *   hg_setup(NULL, ..)). Returns 0 accept, 1 reject (reason in hg_last_error), -1 error. */
int hg_verify_bn254(const hg_pk* pk, const hg_witness* w, const uint8_t* proof, size_t len);

/* profiling: level 0 off, 1 = events around the selected kernel class only, 2 = every class */
int hg_profile(hg_ctx* ctx, int level);
/* selects the class that level 1 times (a name hg_profile_get reported); returns 0, or -1 if there is no such class */
int hg_profile_select(hg_ctx* ctx, const char* name);
int hg_profile_reset(hg_ctx* ctx);
int hg_profile_get(hg_ctx* ctx, hg_kernel_stat* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
