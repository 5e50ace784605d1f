// Launch wrappers of the gfx950 kernels (definitions in kernels.hip). All pointers are device
// pointers; every launch goes to the given stream; nothing here synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include "gl_field.hpp"

namespace hg {
namespace dev {

constexpr int SC_MAX_BLOCKS = 1024;  // fixed partial-sum fan-in of the round kernels
// A `partials` buffer holds PARTIALS_E2 per-workgroup partial sums followed by PARTIALS_TICKETS arrival counters
// (zero between launches): the workgroup that arrives last sums the partials of its job, so a round needs no
// second launch. One buffer per stream.
constexpr size_t PARTIALS_E2 = (size_t)SC_MAX_BLOCKS * 6 * 64;
constexpr int PARTIALS_TICKETS = 64 * 32;  // one 128-byte line per job of a launch
constexpr size_t PARTIALS_BYTES = PARTIALS_E2 * sizeof(E2) + PARTIALS_TICKETS * sizeof(unsigned);
constexpr int PS_MAX_PAIRS = 32;     // max (input, bookkeeping) table pairs of one PRODSUM sum-check
constexpr int PW_MAX = 64;           // max batched products / memories carried in kernarg powers

struct Powers { E2 v[PW_MAX]; };     // by-value kernarg: gamma^i (grand product) or M^i (collation)
// by-value kernarg: the output claims of one node. Points and alphas are runs of the (device-resident)
// challenge chain, so a claim set is just offsets into it.
constexpr int MAX_CLAIMS = 32;
struct ClaimSet {
    int n;                       // number of claims
    int unit_alpha;              // n == 1: alpha = 1 (nothing squeezed)
    size_t alpha_off;            // chain index (E units) of alpha_0
    size_t point_off[MAX_CLAIMS];  // chain index of each point's coordinate 0
};

enum ScKind { SC_COLLATION = 0, SC_GRANDPROD = 1 };

// Stride-layout sum-check instance (collation / grand-product shapes), device-visible descriptor.
// Table t of round 0 lives at in + t*in_stride (u64 if base else E2). All instances of a prover run are
// scheduled round-synchronously: launch k runs, for every instance, its next round(s), whatever the sizes - the
// items of a launch share a 1-D grid (StItem::blk0 / nblk); instances are independent on the device, only the
// transcript orders them. Because the challenges do not
// depend on the prover's messages (transcript.rs:146-157), two consecutive rounds of an instance can also run
// in ONE launch (st_step2): the intermediate folded tables then never touch HBM.
// Grand product #1 without materialised hash tables: the first round of the top layer recomputes the multiset hashes
// h = dim + E*gamma + ts*gamma^2 - tau (prover.rs:44; write hash = read hash + gamma^2) from the integer tables. The read pair
// and the write pair of one memory share every input, and the memories of a chunk share dim / ts.
struct GpHashMem {
    const u64* ep;       // E_m (2^nu integers)
    int chunk;           // dimension index: which dim / read_ts columns (the counters are indexed by chunk, lasso.rs:317-319)
    int rd_row, wr_row;  // this job's pair indices of the memory's read / write tables (weight pw[row], tables 2 row, 2 row + 1); -1: not held
    u32 cutoff;          // E_m[j] = (row j's lookup uses memory `mem` and limb < cutoff) ? limb : 0 (k_lasso_split): ep == nullptr means
    int mem;             // "recompute it from the chunk's limb" (GpHashSrc::seg_lookup ...), the E tables are then never written
};
struct GpHashSrc {
    const u64* dim[4];
    const u64* ts[4];
    const GpHashMem* mems;  // chunk-major (memory-GKR order)
    int nmem;
    u64 gamma, gamma2, tau;
    // for recomputed E values (GpHashMem::ep == nullptr): LassoDev's row -> lookup map and the lookups' memory masks
    int seg_shift;
    size_t rows;
    const uint8_t* seg_lookup;
    u64 lookup_uses[32];
    // Slot form of the mirrored top layer (prover.hip: grand_product): inside a lookup segment the memories the lookup does not use
    // have, per chunk position, identical hash rows, and the layer pairs segment s with s + npairs - memories in the same class in
    // both segments ("joint class") contribute w_b l r with the same l r. slot_of[row * npairs + sp] numbers the joint classes of
    // segment pair sp (slot 0 = row 0 alone), rep[slot * npairs + sp] names the read row that represents a slot there, slotw[(slot *
    // npairs + sp) * 2] = the class weight sum_b gamma^b and [.. + 1] = that times r_0. The job then holds 2 V + 1 tables (the
    // slots' weighted left / right halves and S); rep = 255: no such class in that segment pair (its tables are zero there).
    // null: memory form (k_gp_first_hash<.., SLOT = false>).
    const uint8_t* slot_of;
    const uint8_t* rep;
    const E2* slotw;
    int npairs, nslots;
    // slot form: product-tree level 1 goes to the rows named by these bit masks instead of one row per memory - bit t of
    // emit_rd[slot * npairs + sp] (emit_wr: for the write row, read + gamma^2) = "row t of J.next_level equals this class's product
    // in segment pair sp". The rows are the next layer's own slot rows (StJob::slotw) or, below the last slot-form layer, the
    // per-memory rows. Memories that represent no class are skipped altogether.
    const u64* emit_rd;
    const u64* emit_wr;
};
// the per-memory tables of a slot-form job, gathered once its tables are down to 2^len_log2 >= npairs entries (every entry still
// inside one segment pair): in / out hold tables of that length (DE-INTERLEAVED like every folded table); left table of read row b
// = ratio[b * npairs + sp] * left table of its slot, right table as it is, S (the last table of both) copied
// has_s = false: a layer below the top one (2 nrows / 2 nslots tables, no S).
void gp_slot_regroup(hipStream_t st, const E2* in, E2* out, const uint8_t* slot_of, const E2* ratio, int nrows, int nslots, int npairs, int len_log2, bool has_s = true);
struct SlotRegroupJob { const E2* in; E2* out; const uint8_t* slot_of; const E2* ratio; int nrows, nslots, npairs, len_log2, sh, has_s; };
SlotRegroupJob gp_slot_regroup_job(const E2* in, E2* out, const uint8_t* slot_of, const E2* ratio, int nrows, int nslots, int npairs, int len_log2, bool has_s);
void gp_slot_regroup_jobs(hipStream_t st, const SlotRegroupJob* jobs, int njobs, size_t max_entries);   // every job in one launch
struct StJob {
    const void* in;
    size_t in_stride;
    E2* buf[2];        // ping-pong storage for folded tables (ntab * len/2, ntab * len/4)
    E2* final_out;     // ntab folded scalars
    int kind, ntab, nvars, base;
    int p0_only;       // grand product on a subset of the batch (multi-GPU): pair 0 only supplies p_0, its product is not summed
    // Slot form of a layer below the top one (see GpHashSrc::slot_of; prover.hip: grand_product): the job's table pairs are joint
    // classes of rows, its input rows are slot rows written by the layer above. The first round takes the weight of pair v at table
    // position 2j from slotw[(v * slot_ng + (2j >> slot_shift)) * 2] (and times r_0 from [.. + 1]) instead of pw / pwr, and writes the
    // next tree level through emit_mask[v * slot_ng + group] as the top layer does (GpHashSrc::emit_rd).
    const E2* slotw;
    const u64* emit_mask;
    int slot_ng, slot_shift;
    // Mirrored grand product (top layer of the Lasso read / write product): every WRITE row is its READ row plus the constant
    // c = gamma^2 and carries kappa times its weight, so the write pairs are never stored or multiplied. With l, r the halves of a
    // read row and S = sum_i w_i (l_i + r_i):  sum_i w_i [l_i r_i + kappa (l_i + c)(r_i + c)] = (1 + kappa) [ sum_i w_i l_i r_i +
    // K1 S + K2 ],  K1 = kappa c / (1 + kappa), K2 = kappa c^2 sum_i w_i / (1 + kappa). The job holds the R read pairs (tables
    // 0 .. 2R-1) and ONE more table, S (index 2R = ntab - 1; S folds like any table because folding is linear and c is constant);
    // every round adds K1 S(t) + K2 to P0 and P1 (Pinf is unchanged: constants cancel in differences) and the host multiplies the
    // round sums by 1 + kappa and derives the write rows' final evaluations (read + c).
    int mirror;        // 1: ntab = 2R + 1 as above
    E2 mk1, mk2;       // K1, K2
    size_t r_off;      // chain index of round 0's challenge
    size_t sums_slot;  // result slots: nv per round
    const GpHashSrc* hash_src;  // first round reads these instead of `in` (st_first_hash); null otherwise
    u64* next_level;   // grand product, first round on base-field rows: also emits the next product-tree level
                       // (v_l[2j] v_r[2j], v_l[2j+1] v_r[2j+1] ARE the first-round products), row stride = in_stride; null: not emitted
    E2 pw[PW_MAX];     // gamma^i (grand product) or M^i (collation)
    E2 pwr[PW_MAX];    // gamma^i * r_0 (grand product: first-round fold of the left tables)
};
// one (job, round) of a step launch: where the round reads and writes (host-planned ping-pong)
struct StItem {
    int job;
    int h_log2;        // this item's half length (log2): a launch may mix sizes
    int jb_log2;       // threads along the pair index (2^jb_log2 of the 256; the rest split the tables)
    int blk0, nblk;    // this item's workgroups are [blk0, blk0 + nblk) of the 1-D grid (planned by st_plan_blocks)
    const void* in;
    size_t in_stride;
    E2* out;           // folded tables of the (last) round, stride = its half length
    // tail launches only (st_tail): rounds [rd, rd + nrounds) = all that remain run inside one workgroup
    int rd, nrounds;
    int ntab;          // tail launches: > 0 = the job's table count from here on (a slot-form job regrouped into its per-memory tables)
};
// Fills jb_log2 / blk0 / nblk of the items of one launch (host side, before upload); returns the grid size.
// `rounds2`: fused two-round launch (one pair index per thread).
int st_plan_blocks(StItem* items, int nitems, bool rounds2);
// step: every item runs its job's round with half = 2^item.h_log2; `grid` from st_plan_blocks
// `mode`: 0 the round, 1 its folds only, 2 its sums only (later rounds of grand-product jobs; kernels.hip: sc_round_body)
void st_step(hipStream_t st, int kind, bool base, const StJob* jobs, const StItem* items, int nitems, int grid, const E2* chal,
             E2* partials, E2* res, bool slot = false, int mode = 0);
// first round of ONE grand-product job whose level-0 rows are recomputed from the Lasso integer tables (StJob::hash_src)
// (`mirror` = the host copy of job->mirror: selects the kernel variant)
// (`recomp`: the memories' E values are recomputed from the limbs, GpHashMem::ep is not read)
void st_first_hash(hipStream_t st, const StJob* job, const StItem* item, int grid, bool mirror, bool recomp, const E2* chal, E2* partials, E2* res, bool slot = false);
// fused step (folded Ext2 inputs; grand-product or collation shape): every item runs its job's rounds with half = 2^h_log2 and 2^(h_log2-1)
void st_step2(hipStream_t st, int kind, const StJob* jobs, const StItem* items, int nitems, int grid, const E2* chal, E2* partials, E2* res);
// LDS-resident tail (st_tail): every item runs ALL rounds from its `rd` on in one workgroup; st_tail_h(ntab, nvars) = log2 of the
// largest half length whose tables fit, i.e. the rounds with half <= 2^st_tail_h belong to the tail; table_bytes = max over the
// items of ntab * (2^h0 + 2^(h0-1)) * sizeof(E2), h0 = nvars - 1 - rd
int st_tail_h(int ntab, int nvars);
void st_tail(hipStream_t st, int kind, const StJob* jobs, const StItem* items, int nitems, size_t table_bytes, const E2* chal, E2* res);
constexpr int ST_STEP2_MIN_H = 9;  // fused steps need 2^h_log2 >= 2 * 256 (one pair index per thread, lane pairs share the second round)

// PRODSUM sum-check instance (g = sum_i a_i b_i), device-visible descriptor; instances of equal nvars are
// batched over grid.y so that the 2k+1 independent FFT-node reductions etc. advance in lock step.
struct PsJob {
    const void* a[PS_MAX_PAIRS];  // u64* inputs (round 0)
    const E2* b[PS_MAX_PAIRS];    // bookkeeping tables (round 0)
    E2* fin_a[PS_MAX_PAIRS];      // where the fully folded scalars go
    E2* fin_b[PS_MAX_PAIRS];
    E2* bufa[2];                  // ping-pong storage, table i at buf + i * (current length)
    E2* bufb[2];
    int npairs, nvars;
    int tail_rd;                  // first round of the single-workgroup tail (host-planned)
    int tail_buf;                 // where that round reads: -1 = a[] / b[], else bufa/bufb[tail_buf]
    size_t r_off;                 // chain index of round 0's challenge
    size_t sums_slot;             // result slots: 2 per round
    // Eq-factored job (eq_n = npairs > 0): EVERY bookkeeping table is b_i(x) = kappa_i eq(z', x) for one point z'
    // (a Libra phase-1 table whose wiring relays aligned blocks: a slice of the node's eq table times a constant). No b table exists
    // before the tail: with rho the challenges so far, A = sum_i kappa_i a_i and P_rd = prod_(j<rd) eq(z'_j, rho_j), round rd's
    // polynomial is P_rd eq(z'_rd; X) [(1 - X) S_0 + X S_1], S_t = sum_x' eq(z'_(rd+1..); x') A(rho, t, x'): inner products of
    // the folded A with a suffix eq table, whatever npairs is; the a_i are only folded (kernels.hip: ps_eq_step2_body). The suffix
    // table is never stored at full length either: it is a per-thread factor lo[tid] times SUF_k[tile] (PsEqPoint), the second
    // uniform over a workgroup's tile. eq_single (npairs == 1): A is a_0 itself, kappa_0 is folded into the prefactors.
    // eq_scal: [2 rd], [2 rd + 1] = P_rd (1 - z'_rd), P_rd (3 z'_rd - 1) (times kappa_0 if eq_single); [2 nvars] = P_tail_rd;
    // [2 nvars + 1 + i] = kappa_i; [2 nvars + 1 + npairs + k] = z'_k; then four constants per pass (rounds 2s, 2s+1): (1-ra)(1-rb),
    // ra (1-rb), (1-ra) rb, ra rb. The tail materialises b_i = kappa_i P_tail_rd SUF_tail_rd in bufb[tail_buf] and runs as ever.
    int eq_n, eq_single;
    const E2* eq_scal;
    const E2* eq_lo;              // PsEqPoint::lo of the job's point
    const E2* eq_suf;             // PsEqPoint::suf
    E2* bufA[2];                  // folded A, ping-pong like bufa (one table); unused if eq_single
    E2* eqA0;                     // A itself (2^nvars entries, 4-way de-interleaved), built by ps_eq_A ahead of round 0; unused if eq_single
};
// One point z' of eq-factored jobs and the tables ps_eq_prep builds for it (shared by the jobs opened at the same point).
// z'_k = chain[point_off + k] for k < w, bit (k - w) of hib above (a wiring that relays ONE aligned block of the input: the table
// is zero outside it, i.e. an eq table whose top coordinates are Boolean).
struct PsEqPoint {
    size_t point_off;
    int w, nvars;
    unsigned hib;
    int kmin;                     // SUF_k is stored for k in [kmin, nvars]
    E2* lo;                       // [nvars - 8][384]: round rd -> eq(z'_(rd+1..rd+8); t), t < 256, then eq(z'_(rd+2..rd+8); t), t < 128
    E2* suf;                      // SUF_k = eq(z'_(k..nvars-1); .), 2^(nvars-k) entries at suf + 2^(nvars-k) - 1
};
inline int ps_eq_kmin(int nvars) { int k = nvars - 13 < 9 ? nvars - 13 : 9; return k < 0 ? 0 : k; }
inline size_t ps_eq_lo_entries(int nvars) { return nvars > 8 ? (size_t)(nvars - 8) * 384 : 0; }
inline size_t ps_eq_suf_entries(int nvars) { return ((size_t)2 << (nvars - ps_eq_kmin(nvars))) - 1; }
constexpr int PS_EQ_MAX_VARS = 26;   // (the prep kernel's LDS factor tables)
void ps_eq_prep(hipStream_t st, const PsEqPoint* pts, int npts, const E2* chal);
// A = sum_i kappa_i a_i of jobs[ids[.]] (eq-factored, several tables); max_quads = the largest job's 2^nvars / 4
void ps_eq_A(hipStream_t st, const PsJob* jobs, const int* ids, int nids, size_t max_quads);
// one (job, round) of a PRODSUM launch; the items of a launch share a 1-D grid like StItem
struct PsItem {
    int job, jb_log2, blk0, nblk;
    int rd;                       // (first) round this item runs
    int in_buf, out_buf;          // -1 = the job's a[] / b[] (in) or fin_a / fin_b (out), else bufa/bufb[.]; host-planned ping-pong
    int pad;
};
// fills jb_log2 / blk0 / nblk of the items of one launch (host side); returns the grid size. rounds2: fused launch.
int ps_plan_blocks(PsItem* items, int nitems, const PsJob* host_jobs, bool rounds2);
// one round (rounds2: two consecutive rounds, lane pairs share the second) of every item's job; jobs of different sizes share the launch
// eq: every item is an eq-factored job (a launch holds one kind)
void ps_round(hipStream_t st, bool rounds2, const PsJob* jobs, const PsItem* items, int nitems, int grid, const E2* chal, E2* partials, E2* res, bool eq = false);
// rounds [tail_rd, nvars) of every job, one workgroup per job
void ps_tail(hipStream_t st, const PsJob* jobs, int njobs, const E2* chal, E2* res);
size_t ps_tail_items_max();   // (pair, j) items a job may bring to its first tail round (the folds live in LDS from there on)

// dst_base[ent[i].dst] = ent[i].src[0]: moves locally produced scalars to their global result slots
struct ScatterEnt { const E2* src; size_t dst; };
void scatter_e2(hipStream_t st, const ScatterEnt* ents, size_t n, E2* dst_base);
void set_e2(hipStream_t st, E2* dst, E2 v);   // *dst = v (one thread): a progress mark in the host-mapped result buffer
void stamp(hipStream_t st, unsigned long long* slot);   // debugging aid: device wall clock at this point of the stream
struct ClearSet { unsigned* p[4]; size_t n[4]; };   // up to four regions of 32-bit words
void clear_words(hipStream_t st, const ClearSet& c);

// out[v] = sum_b partials[b*nv + v], v < nv
void reduce_partials(hipStream_t st, const E2* partials, int nblocks, int nv, E2* out);

// eq tables: out = sum_a alpha_a * eq(point_a, .) over n variables (little-endian), batched over grid.y.
// Points are runs of the challenge chain (cs.point_off) or, for the kernel-level entry points, of point_dev.
struct EqJob {
    E2* out;
    int n;
    const E2* point_dev;  // nullptr: points index the chain
    ClaimSet cs;
    // two-launch form (eq_jobs_ab): per claim a low table A (256 entries) and a high table B (2^(n - 8) entries, alpha folded in)
    E2* ab;               // cs.n * eq_ab_entries(n) entries of scratch
    int blk0, rows;       // first workgroup of the fill launch, table rows (of 2^min(n, 8) outputs) per workgroup
    int pblk0;            // first workgroup of the prep launch (one per claim and 256 entries of B)
};
void eq_jobs(hipStream_t st, const EqJob* jobs, int njobs, int max_n, const E2* chal);
// The same tables in two launches: k_eq_prep builds A and B of every (job, claim) - a few hundred workgroups, ~14 dependent
// products - and k_eq_fill streams out[idx] = sum_a A_a[idx & 255] * B_a[idx >> 8]: ONE product per claim and output and no
// per-workgroup setup (k_eq_jobs rebuilds A and a slice of B in every workgroup: 21 dependent products ahead of 16-32 outputs
// per thread). eq_ab_plan fills ab-independent launch fields of the HOST copies (blk0, rows) and returns the fill grid.
__host__ __device__ inline size_t eq_ab_entries(int n) { return 256 + ((size_t)1 << (n > 8 ? n - 8 : 0)); }
struct EqAbGrid { int prep, fill; };
EqAbGrid eq_ab_plan(EqJob* host_jobs, int njobs);
void eq_jobs_ab(hipStream_t st, const EqJob* jobs, int njobs, EqAbGrid grid, const E2* chal);
void sum_tables(hipStream_t st, E2* out, const E2* tabs, int ntabs, size_t n);  // out[i] = sum_t tabs[t*n + i]

// ---- Lasso ------------------------------------------------------------------------------------
struct LassoDev {
    int nu, alpha, num_lookups, seg_shift;
    size_t rows;
    const uint8_t* seg_lookup;   // device
    u64 lookup_mask[32];         // per lookup: (1 << total_bits) - 1
    u64 lookup_uses[32];         // per lookup: bitmask of memories
    int mem_dim[32];
    u32 mem_cutoff[32];
    u64 mpow[5];                 // M^i
    int lookup_nmems[32];
    int lookup_mems[32][4];
    // row segments that touch counter memory c (c = chunk index < 4): only those rows are ranked
    int cnt_nsegs[4];
    uint8_t cnt_segs[4][128];
};
// dims[c][j] (4 x 2^nu) and E[m][j] (alpha x 2^nu), zero beyond `rows`.
// EpRows (multi-GPU: a rank only materialises the E tables of its own memories): row[m] = row of memory m in e_polys, -1 = not held.
struct EpRows { signed char row[32]; };
EpRows ep_rows_all(int alpha);
// col (optional, with colpow): col[j] = sum_m colpow.v[m] E_m[j] over the memories held - the ONE table the collation sum-check
// needs besides E_0: its round polynomial is E_0(t) * sum_m M^m E_m(t), and the sum folds as a single table (folding is linear).
// (dims may be null when lasso_dims has already written the limbs)
void lasso_dims(hipStream_t st, const LassoDev& L, const u64* input, u64* dims);
struct ColPow { u64 v[32]; };   // M^m for the memories to add up, 0 for the others
void lasso_split(hipStream_t st, const LassoDev& L, const u64* input, u64* dims, u64* e_polys, const EpRows& rows, const ColPow* colpow = nullptr,
                 u64* col = nullptr);
// read/final counters of memory m (sequential-scan semantics of lasso.rs:181-196) via a stable sort
size_t lasso_counter_temp_bytes(size_t n);
void lasso_counters(hipStream_t st, const LassoDev& L, int m, const u64* dims, u64* read_ts, u64* final_cts,
                    void* temp, size_t temp_bytes, u32* keys, u32* keys_sorted, u32* rows_in, u32* rows_sorted, u32* starts);
// The same for SEVERAL counter memories (chunks) in one pass: one stable sort of (chunk << 16 | address) keys over the rows of all
// requested chunks (13 launches instead of 12 per chunk). chunk_mask: bit c = compute chunk c; read_ts[c] / final_cts[c] as above.
struct CounterOut { u64* read_ts[4]; u64* final_cts[4]; };
size_t lasso_counters_all_elems(const LassoDev& L, unsigned chunk_mask);             // number of (row, chunk) pairs to sort
size_t lasso_counters_all_temp_bytes(size_t n_elems);
void lasso_counters_all(hipStream_t st, const LassoDev& L, unsigned chunk_mask, const u64* dims, const CounterOut& out, void* temp, size_t temp_bytes,
                        u32* keys, u32* keys_sorted, u32* vals, u32* vals_sorted, u32* starts /* 4 * 65536 + 1 */);
// sum_k eq[k] * sum_i M^i E_{mems(lookup(k))[i]}[k]  -> partials (nv = 1)
// (memories not held - EpRows::row < 0 - contribute nothing: the ranks' partial claimed sums add up)
int lasso_claim(hipStream_t st, const LassoDev& L, const E2* eq, const u64* e_polys, const EpRows& rows, E2* partials);
// ... with the E values recomputed from the node input (no E tables); own = bit mask of the memories whose terms are summed
int lasso_claim_in(hipStream_t st, const LassoDev& L, const E2* eq, const u64* input, u32 own, E2* partials);
// multiset hashes h = a + v*gamma + t*gamma^2 - tau
// for up to HASH_RW_MAX memories of one chunk (shared dim / read_ts columns); n >= 4, a multiple of 4
// rd1/wr1 (may be null): first product-tree level rd[j]*rd[j+n/2], wr[j]*wr[j+n/2], n/2 entries each
constexpr int HASH_RW_MAX = 8;
struct HashRwArgs { const u64* ep[HASH_RW_MAX]; u64* rd[HASH_RW_MAX]; u64* wr[HASH_RW_MAX]; u64* rd1[HASH_RW_MAX]; u64* wr1[HASH_RW_MAX]; };
void lasso_hash_rw(hipStream_t st, size_t n, const u64* dim, const u64* read_ts, const HashRwArgs& args, int nmem, u64 gamma, u64 tau);
// init / final hashes of all G memories in one launch: H2[i] = init_i, H2[G + i] = final_i (2^16 entries each)
struct HashIfArgs { u32 cutoff[32]; const u64* fc[32]; int row_init[32], row_fin[32]; };  // rows of H2 (2^16 entries each); -1 = skip
void lasso_hash_if(hipStream_t st, const HashIfArgs& args, int G, u64 gamma, u64 tau, u64* H2);
// product tree level: out[b][i] = in[b][i] * in[b][i + h], b < nb, i < h
void prod_level(hipStream_t st, const u64* in, size_t in_len, u64* out, int nb);
// three levels in one launch (in_len a multiple of 8): o1, o2, o3 = the levels of in_len/2, in_len/4, in_len/8 entries per row
void prod_level3(hipStream_t st, const u64* in, size_t in_len, u64* o1, u64* o2, u64* o3, int nb);
// all levels above a level of in_len <= PROD_TAIL_LEN entries in one launch (one workgroup per row)
constexpr int PROD_TAIL_LEN = 2048;
struct ProdTailOut { u64* p[12]; };
void prod_tail(hipStream_t st, const u64* in, int in_len, const ProdTailOut& outs, int nlevels, int nb);
// gathers: roots[b] = top[b][0]*top[b][1] (as E2) ; evals[2b+s] = top[b][s]
void gp_top(hipStream_t st, const u64* top, int nb, E2* roots, E2* evals);
// dot products with an eq table: out[t] = sum_j eq[j] * tabs[t][j] for ntab <= 8 base tables; `partials` is a partials buffer
void dot_eq(hipStream_t st, const E2* eq, const u64* const tabs[8], int ntab, size_t n, E2* partials, E2* out);
// the same for up to DOT_MAX tables in one launch (+ one reduction launch): result t goes to out[tabs.slot[t]]
constexpr int DOT_MAX = 64;
struct DotTabs { const u64* t[DOT_MAX]; int slot[DOT_MAX]; signed char emem[DOT_MAX]; };   // t == nullptr: the E table of memory emem (DotVirt)
// what recomputing E_m[j] = (row j's lookup uses m and limb < cutoff_m) ? limb : 0 takes (k_lasso_split): the node input and LassoDev's maps
struct DotVirt {
    const u64* input; const uint8_t* seg_lookup; int seg_shift; size_t rows;
    u64 lookup_mask[32]; u64 lookup_uses[32]; int mem_dim[32]; u32 mem_cutoff[32];
};
void dot_eq_many(hipStream_t st, const E2* eq, const DotTabs& tabs, int ntab, size_t n, E2* partials, E2* out, const DotVirt* virt = nullptr);
// The Lasso node's openings at x in their own launch shape: a workgroup walks a contiguous run of rows inside ONE lookup segment, so
// of the alpha recomputed E tables only the few memories that lookup uses (one per chunk position) can be non-zero there - one
// multiply-add per row and USED memory instead of one per row and memory, eq and the node input read once for all of them; the
// materialised tables (chunk values, read counters) in groups of eight beside it. Returns false (nothing launched) when the shape
// does not fit (rows per workgroup not a power of two inside a segment, a lookup with more than eight memories): use dot_eq_many.
bool open_x(hipStream_t st, const E2* eq, const DotTabs& tabs, int ntab, size_t n, E2* partials, E2* out, const DotVirt& virt);

// ---- Vanilla / FFT nodes ----------------------------------------------------------------------
struct CsrLin { const u32* ptr; const u32* gate; const u64* coef; };           // per input position -> (gate, c)
struct CsrMul { const u32* ptr; const u32* gate; const u64* coef; const u32* other_in; const u32* other_j; };
// T[rep*S + x] = sum_lin eqc[rep*G+gate]*c + sum_mul eqc[rep*G+gate]*c*in_other[rep*S + j1]
struct GatherT {
    CsrLin lin; CsrMul mul;     // ptr == nullptr when absent
    const u64* in_vals[PS_MAX_PAIRS];  // node input tables (for the mul part)
};
struct GatherJob { GatherT g; const E2* eqc; int log2_S, log2_G, log2_R; E2* T; };
void gather_jobs(hipStream_t st, const GatherJob* jobs, int njobs, size_t max_total);
// Run-length form of the same table for wiring that is affine in the input position (every node of the BFV circuit:
// relays, scaled relays, sums and element-wise products connect position x to gate x + const): a short list of segments
//   T[rep*S + x] += c * eqc[rep*G + x + goff] (* in_other[rep*S + x + joff])   for lo <= x < hi
// read through scalar loads; every table access is coalesced and no per-term metadata is streamed.
struct GatherSeg { u32 lo, hi; int goff; int other_in; int joff; int pad; u64 coef; };  // other_in < 0: linear term
struct GatherSegJob { const GatherSeg* segs; int nseg; const E2* eqc; int log2_S, log2_G, log2_R; E2* T; const u64* in_vals[PS_MAX_PAIRS]; int blk0; };
void gather_seg_jobs(hipStream_t st, const GatherSegJob* jobs, int njobs, int grid);
int gather_seg_plan(GatherSegJob* host_jobs, int njobs);   // fills blk0, returns the grid
// B[rep*S + y] = sum_mulR eqc[rep*G+gate]*c*eqx[rep*S + j0]*u[i0]
void vanilla_gather_B(hipStream_t st, const CsrMul& mulR, const E2* eqc, const E2* eqx, const E2* u, int log2_S, int log2_G, int log2_R, E2* B);
struct GatherBJob { CsrMul m; const E2* eqc; const E2* eqx; const E2* u; int log2_S, log2_G, log2_R; E2* B; };   // u: the left inputs' values at r_x (result slots), or null = ones
void gather_B_jobs(hipStream_t st, const GatherBJob* jobs, int njobs, size_t max_total);
// sum over reps and constant gates of eqc[rep*G+gate]*c -> partials (nv = 1)
int vanilla_const_sum(hipStream_t st, const u32* gate, const u64* coef, size_t nterms, const E2* eqc, int log2_G, int log2_R, E2* partials);
// F_c(x) = sum_a alpha_a * scale * prod_b (1 + r_{a,b} (W[(x<<b) & (N-1)] - 1))
struct FftJob { E2* out; const u64* W; u64 scale; int L; ClaimSet cs; };
// `tab`: scratch of njobs * max_claims * 2^(max_L - 4) E2 (the high-bit factor tables)
void fft_jobs(hipStream_t st, const FftJob* jobs, int njobs, int max_L, int max_claims, const E2* chal, E2* tab);
void powers_table(hipStream_t st, u64* W, u64 w, size_t n);  // W[i] = w^i

// ---- circuit evaluation (witness generation): one Vanilla node, gate-major wiring -------------------------
struct EvalNode {
    const u32* lptr; const u32* lin_in; const u32* lin_j; const u64* lcoef;   // linear terms per gate
    const u32* mptr; const u32* mi0; const u32* mj0; const u32* mi1; const u32* mj1; const u64* mcoef;  // mul terms per gate
    const u64* w0;                 // dense constants per gate, or nullptr
    const u64* in[PS_MAX_PAIRS];   // input tables
    u64* out;
    int log2_G, log2_S, log2_R;
    u32 num_gates;
};
void gate_eval(hipStream_t st, const EvalNode& n);

// ---- NTT (witness generation / hg_ntt) ----------------------------------------------------------
// In place, natural order in and out. 8 <= log2n <= 16 with a scratch buffer of batch*N u64: LDS-resident four-step
// (two launches); otherwise radix-2 stages in HBM. W = w^i for i < N.
void ntt_batch(hipStream_t st, u64* data, int log2n, size_t batch, const u64* W, u64 scale, u64* scratch);

}  // namespace dev
}  // namespace hg
