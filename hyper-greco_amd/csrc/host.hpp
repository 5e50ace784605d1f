// Host side of the prover: Fiat-Shamir chain + proof stream, parameter tables, witness loading /
// synthesis, Lasso preprocessing and the circuit wiring of the BFV sk-encryption circuit.
// Mirrors the reference's host-level API for this path (names and argument meaning):
//   Keccak256Transcript         [REF bfv-gkr/src/transcript.rs:117-203]
//   BfvSkEncryptConstans        [REF bfv-gkr/src/constants/mod.rs:16-35]
//   Poly::{new,new_padded,new_shifted}, BfvEncrypt::get_inputs  [REF poly.rs:12-44, sk_encryption_circuit.rs:365-415]
//   LassoPreprocessing::preprocess, RangeLookup, {FullLimb,Bound}Subtable  [REF lasso.rs:527-627, table/range.rs]
//   BfvEncrypt::configure       [REF sk_encryption_circuit.rs:86-293, 351-363]
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include "gl_field.hpp"
#include "../../include/hg.h"

namespace hg {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------------------------------------
// Fiat-Shamir. The reference transcript never absorbs prover messages (write_felt only appends to
// the stream, common_felt is a no-op, transcript.rs:156,183-189), so the challenges are the fixed
// chain c_j = LE(H_j) mod p, H_1 = Keccak256(""), H_{j+1} = Keccak256(H_j). The chain is cached
// process-wide; `ChallengeSource` is the seam where an absorbing transcript would plug in.
void keccak256(const uint8_t* data, size_t len, uint8_t out[32]);
int cgroup_cpu_quota();  // CPUs the cgroup grants (0 = unlimited / unknown)
int hg_omp_threads();    // threads one of the library's OpenMP regions may use (runtime default capped by the quota)
const u64* challenge_chain(size_t n_base);  // pointer to >= n_base cached base-field challenges
// Diagnostics on stderr: HG_DEBUG = comma-separated tokens, read at every call (tests set and unset it inside one process).
//   shard: per-rank times of a sharded prove, the shard plan, failed graph captures    slots: the slot-form layers a prove adopted
//   eq: how many queued node reductions run eq-factored    launch: host time of every launch-graph replay    fail_capture: makes the next launch-graph capture fail (fallback test)
bool hg_debug(const char* token);
// Timing breakdowns on stderr: HG_TIMES = comma-separated tokens, read at every call.
//   seq: the round-by-round prover (waits, host steps, drains)    bn: the bn254 prove and its witness generation
//   verify: hg_verify / hg_verify_device    json: the JSON witness loader    setup: the phases of hg_setup
bool hg_times(const char* token);
const char* hg_proof_map_path();   // HG_PROOF_MAP=<file>: byte offset of every protocol element of a proof (scripts/proof_diff.py); null: off
bool hg_env_on(const char* name);  // "<name>=1" in the environment (read once per call site through a static)
u64 felt_from_hash(const uint8_t h[32]);    // fe_mod_from_le_bytes (transcript.rs:202): 256-bit little-endian integer mod p

// The transcript with its hash state kept explicitly: the bytes absorbed since the last squeeze (H::update appends;
// squeeze_challenge = finalize_fixed_reset, then update(hash), transcript.rs:198-203). With `absorb` off nothing but the
// previous hash is ever in the state and the challenges are the fixed chain above. With it on, write_felt also hashes the
// element the way the in-tree plonkish-trait writer of the same struct does (common_field_element -> update(to_repr),
// transcript.rs:205-208, 224-233): SURVEY.md 8(f) f-4.
struct FsTranscript {
    bool absorb = false;
    std::vector<uint8_t> pending;
    std::vector<uint8_t> bytes;  // the proof stream
    u64 squeeze_f() {
        uint8_t h[32];
        keccak256(pending.data(), pending.size(), h);
        pending.assign(h, h + 32);
        return felt_from_hash(h);
    }
    E2 squeeze() { u64 a = squeeze_f(); u64 b = squeeze_f(); return e2(a, b); }
    void write_f(u64 a) {
        if (absorb) { uint8_t le[8]; memcpy(le, &a, 8); pending.insert(pending.end(), le, le + 8); }
        u64 be = __builtin_bswap64(a);
        size_t at = bytes.size();
        bytes.resize(at + 8);
        memcpy(bytes.data() + at, &be, 8);
    }
    void write_e(E2 a) { write_f(a.c0); write_f(a.c1); }
    // verifier side: read_felt absorbs what it read (transcript.rs:210-222)
    void absorb_read(u64 a) { if (absorb) { uint8_t le[8]; memcpy(le, &a, 8); pending.insert(pending.end(), le, le + 8); } }
};

struct ChallengeSource {
    size_t pos = 0;  // base-field challenges consumed so far
    E2 squeeze() {   // E challenge = from_bases of DEGREE = 2 consecutive base challenges (transcript.rs:149-154)
        const u64* c = challenge_chain(pos + 2);
        E2 r = e2(c[pos], c[pos + 1]);
        pos += 2;
        return r;
    }
    std::vector<E2> squeeze_n(size_t n) { std::vector<E2> v(n); for (auto& x : v) x = squeeze(); return v; }
};

struct ProofStream {
    std::vector<uint8_t> bytes;
    size_t pos = (size_t)-1;   // != -1: write at this offset of the (pre-sized) buffer instead of appending (out-of-order transcript replay)
    void write_f(u64 a) {  // canonical repr, byte-reversed to big-endian (transcript.rs:183-189)
        u64 be = __builtin_bswap64(a);
        if (pos != (size_t)-1) { memcpy(bytes.data() + pos, &be, 8); pos += 8; return; }
        size_t at = bytes.size();
        bytes.resize(at + 8);
        memcpy(bytes.data() + at, &be, 8);
    }
    void write_e(E2 a) { write_f(a.c0); write_f(a.c1); }                                           // bases in order (:191-195)
};

// ---------------------------------------------------------------------------------------------
struct Params {
    hg_params raw;
    int n_log2, L;  // L = log2_size = N_LOG2 + 1 (sk_encryption_circuit.rs:81-83)
    int k, log2k;
    explicit Params(const hg_params& p);
    size_t SZ() const { return (size_t)1 << L; }
    size_t PZ() const { return (size_t)1 << n_log2; }
    int ct0is_log2() const { return L + log2k; }  // sk_encryption_circuit.rs:519-522
    int num_chunks() const { return k / 2 > 1 ? k / 2 : 1; }
};
bool params_builtin(uint32_t n, uint32_t k, hg_params* out);
void params_derive(uint32_t n, uint32_t k, const u64* qis, u64 t, hg_params* out);  // scripts/circuit_sk.py:422-439

struct Witness {  // tables exactly as get_inputs lays them out
    std::vector<u64> s, e, k1;  // 2^L
    std::vector<u64> ais, r1is; // k * 2^L
    std::vector<u64> r2is;      // k * 2^P
    std::vector<u64> ct0is;     // k * 2^L
};
Witness witness_from_json(const Params& p, const std::string& path);
Witness witness_from_json_bn254(const Params& p, const std::string& path);  // bn256::Fr fixture -> signed integers in Goldilocks form
Witness witness_synthetic(const Params& p, u64 seed);

// ---------------------------------------------------------------------------------------------
// Lasso preprocessing (host description; the device copy lives in the prover key)
struct LassoLookup {
    u64 bound;                // RangeLookup bound (2b+1)
    std::string id;           // "range_{bound}"
    int total_bits;           // sum(chunk_bits)
    std::vector<int> mems;    // lookup_to_memory_indices
};
struct LassoMemory {
    int subtable, dim;
    u64 cutoff;               // T[a] = a < cutoff ? a : 0   (65536 for the full-limb table)
    std::string subtable_id;
};
struct LassoPlan {
    static constexpr int C = 4, LOGM = 16;
    std::vector<LassoLookup> lookups;  // BTreeMap<String,_> order
    std::vector<LassoMemory> mems;
    std::vector<std::string> subtable_ids;
    std::vector<u64> subtable_bound;   // 0 = full
    int alpha = 0;
    // node
    int nu = 0;
    size_t rows = 0;
    int seg_shift = 0;                 // rows come in segments of 2^seg_shift with one lookup type each
    std::vector<uint8_t> seg_lookup;
    // memory checking order: chunks by dimension, memories ascending (lasso.rs:303-336)
    std::vector<std::pair<int, std::vector<int>>> chunks;
    std::vector<int> gkr_order, gkr_chunk;
    int lookup_index(u64 bound) const;
    std::string layout_text() const;
};
LassoPlan lasso_preprocess(const Params& p);

// ---------------------------------------------------------------------------------------------
// Circuit wiring
enum NodeKind { NK_INPUT = 0, NK_VANILLA = 1, NK_FFT = 2, NK_LASSO = 3 };
struct LinTerm { u32 gate, in, j; u64 c; };
struct MulTerm { u32 gate, i0, j0, i1, j1; u64 c; };
struct ConstTerm { u32 gate; u64 c; };
struct HNode {
    NodeKind kind = NK_INPUT;
    int log2_size = 0;  // input / fft: log2 of the whole table
    int arity = 0, log2_sub_in = 0, log2_sub_out = 0, log2_reps = 0;
    u32 num_gates = 0;
    bool inverse = false;
    std::vector<ConstTerm> w0;
    std::vector<LinTerm> lin;
    std::vector<MulTerm> mul;
    std::vector<int> preds, succs;
    std::vector<char> left_use, right_use;  // per input: appears as linear/left operand, as right operand
    int log2_out() const { return kind == NK_VANILLA ? log2_sub_out + log2_reps : (kind == NK_LASSO ? 0 : log2_size); }
};
struct HCircuit {
    std::vector<HNode> nodes;
    std::vector<int> topo;  // Kahn order, smallest id first
    std::vector<int> input_ids;
    int lasso_id = -1, lasso_in_id = -1, sum_id = -1;
};
HCircuit build_circuit(const Params& p, const LassoPlan& lp);

// host witness generation = Circuit::evaluate (sk_encryption_circuit.rs:442); returns one table per node
std::vector<std::vector<u64>> circuit_evaluate(const HCircuit& c, const Params& p, const Witness& w);
// BfvEncrypt::verify on the host; "" = accept, otherwise the rejection reason (verifier.cpp)
std::string verify_proof(const Params& p, const LassoPlan& lp, const HCircuit& c, const Witness& w, const uint8_t* proof, size_t len, int mode = 0);
// The verifier's table-sized work done elsewhere (verifier_dev.hip: on the device). Goldilocks, mode 0 only: every evaluation point
// is a run of the fixed challenge chain, so a point is an offset into it. Every method DEFERS: it enqueues work and returns a
// ticket; value(ticket) is valid after finish(). The walk itself (proof parsing, sum-check round checks, the Lasso scalar checks)
// stays on the host and never waits for a ticket - checks that need one are evaluated after finish().
struct VerifyBackend {
    struct ClaimOffs { std::vector<size_t> point_off; size_t alpha_off = 0; bool unit = true; };   // alpha_off: chain offset of alpha_0 (claims > 1)
    virtual ~VerifyBackend() {}
    virtual void begin_node(int node, const ClaimOffs& cl) = 0;     // the node's combined eq table over its outputs
    virtual int const_sum() = 0;                                     // sum over reps and constant gates of eqc c
    virtual void set_x(size_t x_off) = 0;                            // eq table of the phase-1 / FFT point
    virtual std::vector<int> lin_terms() = 0;                        // per input i: sum over its linear gates of c eqc[g] eqx[j] (-1: none)
    virtual void set_y(size_t y_off, const std::vector<E2>& u) = 0;  // eq table of the phase-2 point; the phase-1 evaluations
    virtual std::vector<int> mul_terms() = 0;                        // per input i1: sum over mul gates (.., i1) of u[i0] c eqc eqx[j0] eqy[j1] (-1: none)
    virtual int fft_term() = 0;                                      // sum_x F_c(x) eqx(x), alphas and the inverse scale included
    virtual void end_node() = 0;
    virtual int mle_input(size_t k, size_t point_off, int nvars) = 0;   // input table k (chain_par! order) at a point
    virtual int mle_ct0is(size_t point_off, int nvars) = 0;
    virtual void finish() = 0;
    virtual E2 value(int ticket) const = 0;
};
std::string verify_proof_with(VerifyBackend& dev, const Params& p, const LassoPlan& lp, const HCircuit& c, const uint8_t* proof, size_t len);
// the same over bn256::Fr (F = E = Fr, 32-byte proof elements): the bn254 test family
std::string verify_proof_bn254(const Params& p, const LassoPlan& lp, const HCircuit& c, const Witness& w, const uint8_t* proof, size_t len);
void ntt_host(u64* a, int log2n, bool inverse);  // in place, natural order
u64 root_of_unity(int log2n);                    // 2^log2n-th root from ROOT_OF_UNITY = 7^((p-1)/2^32)

}  // namespace hg
