// ORACLE (test infrastructure). CPU restatement of the Lasso lookup node, following the in-tree
// reference files line by line (each function cites what it follows):
//   /root/reference/lasso/src/lasso.rs, table/range.rs, memory_checking/{prover,mod,verifier}.rs
// The sum-check core it calls is external (see sumcheck.hpp: parity unpinned there).
#pragma once
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include <stdexcept>
#include "field.hpp"
#include "poly.hpp"
#include "sumcheck.hpp"
#include "transcript.hpp"

namespace ORC_NS {

static const size_t LASSO_C = 4;          // sk_encryption_circuit.rs:30
static const size_t LASSO_LOGM = 16;      // sk_encryption_circuit.rs:29
static const size_t LASSO_M = 1u << 16;   // sk_encryption_circuit.rs:31

static inline unsigned ilog2_u64(uint64_t x) { return 63 - __builtin_clzll(x); }

struct Subtable {
    bool full;
    uint64_t bound;  // for BoundSubtable
    std::string id;  // range.rs:40-42 "full", :163-165 "bound_{b}"
    // BoundSubtable::materialize range.rs:58-72
    uint64_t cutoff() const {
        unsigned bits = ilog2_u64(bound);
        uint64_t rem = 1ull << (bits % LASSO_LOGM);
        return rem + bound % LASSO_M;
    }
    inline uint64_t value(uint64_t i) const { return full ? i : (i < cutoff() ? i : 0); }
    std::vector<uint64_t> materialize() const {  // range.rs:15-17, :58-72
        std::vector<uint64_t> t(LASSO_M);
        uint64_t c = full ? LASSO_M : cutoff();
        for (uint64_t i = 0; i < LASSO_M; i++) t[i] = i < c ? i : 0;
        return t;
    }
    // range.rs:19-26 and :74-112
    E evaluate_mle(const std::vector<E>& point) const {
        size_t b = point.size();
        E result = e_zero();
        if (full) {
            for (size_t i = 0; i < b; i++) result = e_add(result, e_mul_f(point[i], f_from_u64(1ull << i)));
            return result;
        }
        uint64_t co = cutoff();
        unsigned cl2 = ilog2_u64(co);
        uint64_t g_base = 1ull << cl2, num_extra = co - g_base;
        for (size_t i = 0; i < b; i++) {
            if (i < cl2) {
                result = e_add(result, e_mul_f(point[i], f_from_u64(1ull << i)));
            } else {
                E g_value = e_zero();
                if (i == cl2) {
                    for (uint64_t k = 0; k < num_extra; k++) {
                        E term = e_from_f(f_from_u64(g_base + k));
                        for (unsigned j = 0; j < cl2; j++)
                            term = e_mul(term, (k >> j) & 1 ? point[j] : e_sub(e_one(), point[j]));
                        g_value = e_add(g_value, term);
                    }
                }
                result = e_add(e_mul(e_sub(e_one(), point[i]), result), e_mul(point[i], g_value));
            }
        }
        return result;
    }
};

struct Lookup {
    uint64_t bound;  // RangeLookup bound (= 2*b+1 at the call sites sk_encryption_circuit.rs:329-340)
    std::string id;  // range.rs:256-258 "range_{bound}"
    // range.rs:234-250
    std::vector<size_t> chunk_bits() const {
        unsigned bits = ilog2_u64(bound);
        std::vector<size_t> v(bits / LASSO_LOGM, LASSO_LOGM);
        if (bound % LASSO_M != 0) {
            uint64_t rem = 1ull << (bits % LASSO_LOGM);
            uint64_t cutoff = rem + bound % LASSO_M;
            v.push_back(ilog2_u64(cutoff));
        }
        return v;
    }
    size_t total_bits() const { size_t s = 0; for (size_t b : chunk_bits()) s += b; return s; }
    // range.rs:207-228 -> list of (subtable, dimension indices)
    std::vector<std::pair<Subtable, std::vector<size_t>>> subtables() const {
        Subtable full{true, 0, "full"};
        Subtable rem{false, bound, "bound_" + std::to_string(bound)};
        unsigned bits = ilog2_u64(bound);
        size_t num_chunks = bits / LASSO_LOGM;
        std::vector<size_t> range;
        for (size_t i = 0; i < num_chunks; i++) range.push_back(i);
        if (bound % LASSO_M == 0) return {{full, range}};
        if (bound < LASSO_M) return {{rem, {0}}};
        return {{full, range}, {rem, {num_chunks}}};
    }
};

// lasso.rs:513-627
struct LassoPre {
    std::vector<Lookup> lookups;  // BTreeMap<String,_> order (lasso.rs:530-534): byte-wise string order
    std::vector<Subtable> subtables;
    std::vector<size_t> mem_subtable, mem_dim;
    std::vector<std::vector<size_t>> subtable_mems;
    std::vector<std::vector<size_t>> lookup_mems;
    std::vector<std::vector<uint64_t>> tables;
    size_t num_memories = 0;

    size_t lookup_index(uint64_t bound) const {
        for (size_t i = 0; i < lookups.size(); i++) if (lookups[i].bound == bound) return i;
        throw std::runtime_error("lookup id not found");
    }

    static LassoPre preprocess(const std::vector<uint64_t>& bounds) {
        LassoPre p;
        std::map<std::string, Lookup> m;  // std::string operator< is byte-wise lexicographic = Rust String Ord
        for (uint64_t b : bounds) { Lookup l{b, "range_" + std::to_string(b)}; m[l.id] = l; }
        for (auto& kv : m) p.lookups.push_back(kv.second);
        // unique subtables in first-seen order (lasso.rs:543-552)
        for (auto& l : p.lookups)
            for (auto& st : l.subtables()) {
                bool seen = false;
                for (auto& s : p.subtables) if (s.id == st.first.id) seen = true;
                if (!seen) p.subtables.push_back(st.first);
            }
        // union of dimension indices per subtable (lasso.rs:555-572)
        std::vector<std::vector<bool>> dims(p.subtables.size(), std::vector<bool>(LASSO_C + 4, false));
        for (auto& l : p.lookups)
            for (auto& st : l.subtables()) {
                size_t si = 0;
                while (p.subtables[si].id != st.first.id) si++;
                for (size_t d : st.second) dims[si][d] = true;
            }
        // memories (lasso.rs:574-586)
        for (size_t si = 0; si < p.subtables.size(); si++) {
            p.subtable_mems.push_back({});
            for (size_t d = 0; d < dims[si].size(); d++)
                if (dims[si][d]) {
                    p.subtable_mems[si].push_back(p.mem_subtable.size());
                    p.mem_subtable.push_back(si);
                    p.mem_dim.push_back(d);
                }
        }
        p.num_memories = p.mem_subtable.size();
        // lookup -> memories (lasso.rs:590-602)
        for (auto& l : p.lookups) {
            std::vector<size_t> mems;
            for (auto& st : l.subtables()) {
                size_t si = 0;
                while (p.subtables[si].id != st.first.id) si++;
                for (size_t mi : p.subtable_mems[si])
                    if (std::find(st.second.begin(), st.second.end(), p.mem_dim[mi]) != st.second.end()) mems.push_back(mi);
            }
            p.lookup_mems.push_back(mems);
        }
        for (auto& s : p.subtables) p.tables.push_back(s.materialize());  // lasso.rs:604-609
        return p;
    }
};

// lasso.rs:381-414 + :654-669 + range.rs:252-254: LE bits of the canonical repr, truncated to
// sum(chunk_bits), cut in 16-bit chunks, at most C of them (C * 16 = 64: only the low 64 bits of the repr matter).
static inline void subtable_lookup_indices_row(uint64_t value, size_t total_bits, uint32_t idx[LASSO_C]) {
    uint64_t v = total_bits >= 64 ? value : (value & ((1ull << total_bits) - 1));
    for (size_t c = 0; c < LASSO_C; c++) idx[c] = (uint32_t)((v >> (16 * c)) & 0xFFFF);
}

// The node's witness polynomials are tables of small non-negative integers (limb indices, counters, subtable values):
// the same in every field. They meet the field through f_from_u64.
struct LassoPolys {
    size_t nu;
    std::vector<std::vector<uint64_t>> dims;       // C x 2^nu
    std::vector<std::vector<uint64_t>> read_cts;   // alpha x 2^nu
    std::vector<std::vector<uint64_t>> final_cts;  // alpha x 2^16
    std::vector<std::vector<uint64_t>> e_polys;    // alpha x 2^nu
};

// base-field view of an integer table: Goldilocks shares the storage (every entry is below p), Fr converts
struct FTab {
    std::vector<F> own;
    const uint64_t* shared = nullptr;
    const F* p() const {
#if ORC_F_IS_U64
        return shared;
#else
        return own.data();
#endif
    }
};
static inline FTab ftab_from_u64(const std::vector<uint64_t>& v) {
    FTab t;
#if ORC_F_IS_U64
    t.shared = v.data();
#else
    t.own.resize(v.size());
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)v.size(); i++) t.own[i] = f_from_u64(v[i]);
#endif
    return t;
}

// lasso.rs:157-250. `row_lookup[j]` = index of row j's lookup in pre.lookups (rows = row_lookup.size()).
static inline LassoPolys polynomialize(const LassoPre& pre, size_t nu, const std::vector<uint8_t>& row_lookup,
                                       const F* inputs) {
    LassoPolys P;
    P.nu = nu;
    const size_t N = (size_t)1 << nu, rows = row_lookup.size();
    std::vector<size_t> tb(pre.lookups.size());
    for (size_t l = 0; l < pre.lookups.size(); l++) tb[l] = pre.lookups[l].total_bits();
    P.dims.assign(LASSO_C, std::vector<uint64_t>(N, 0));
#pragma omp parallel for schedule(static)
    for (long long j = 0; j < (long long)rows; j++) {
        uint32_t idx[LASSO_C];
        subtable_lookup_indices_row(f_low_u64(inputs[j]), tb[row_lookup[j]], idx);
        for (size_t c = 0; c < LASSO_C; c++) P.dims[c][j] = idx[c];
    }
    const size_t A = pre.num_memories;
    P.read_cts.assign(A, {});
    P.final_cts.assign(A, {});
    P.e_polys.assign(A, {});
    std::vector<std::vector<bool>> uses(pre.lookups.size(), std::vector<bool>(A, false));
    for (size_t l = 0; l < pre.lookups.size(); l++) for (size_t m : pre.lookup_mems[l]) uses[l][m] = true;
#pragma omp parallel for schedule(dynamic, 1)
    for (long long mm = 0; mm < (long long)A; mm++) {  // lasso.rs:170-204 (parallel over memories, serial over rows)
        size_t m = (size_t)mm;
        const std::vector<uint64_t>& seq = P.dims[pre.mem_dim[m]];
        const std::vector<uint64_t>& tab = pre.tables[pre.mem_subtable[m]];
        std::vector<uint64_t> fin(LASSO_M, 0), rd(N, 0), ev(N, 0);
        for (size_t j = 0; j < rows; j++) {
            if (uses[row_lookup[j]][m]) {
                uint64_t a = seq[j];
                uint64_t c = fin[a];
                rd[j] = c;
                fin[a] = c + 1;
                ev[j] = tab[a];
            }
        }
        P.read_cts[m].swap(rd);
        P.final_cts[m].swap(fin);
        P.e_polys[m].swap(ev);
    }
    return P;
}

// lasso.rs:422-454 with combine_lookups range.rs:184-195
static inline E lasso_sum_check_claim(const LassoPre& pre, const LassoPolys& P, const std::vector<uint8_t>& row_lookup,
                                      const std::vector<E>& r) {
    std::vector<E> eq = eq_table(r);
    const size_t rows = row_lookup.size();
    std::vector<F> mp(LASSO_C + 1);
    mp[0] = f_one();
    for (size_t i = 1; i < mp.size(); i++) mp[i] = f_mul(mp[i - 1], f_from_u64(LASSO_M));
    E total = e_zero();
#pragma omp parallel
    {
        E a = e_zero();
#pragma omp for nowait schedule(static)
        for (long long k = 0; k < (long long)rows; k++) {
            const std::vector<size_t>& mems = pre.lookup_mems[row_lookup[k]];
            F comb = f_zero();
            for (size_t i = 0; i < mems.size(); i++) comb = f_add(comb, f_mul(f_from_u64(P.e_polys[mems[i]][k]), mp[i]));
            a = e_add(a, e_mul_f(eq[k], comb));
        }
#pragma omp critical
        total = e_add(total, a);
    }
    return total;
}

// element type of the grand-product tables: base field (the reference: gamma, tau truncated to limb 0) or extension
// field (ProtocolMode::ext_memcheck). A tag instead of overloading on the type, because F and E coincide over Fr.
template <bool EXT> struct GpElem;
template <> struct GpElem<false> {
    typedef F T;
    static inline T mul(T a, T b) { return f_mul(a, b); }
    static inline E lift(T a) { return e_from_f(a); }
    static inline ScTable table(const T* p, size_t n) { return ScTable::from_f(p, n); }
};
template <> struct GpElem<true> {
    typedef E T;
    static inline T mul(T a, T b) { return e_mul(a, b); }
    static inline E lift(T a) { return a; }
    static inline ScTable table(const T* p, size_t n) { return ScTable::view_e(p, n); }
};

// prover.rs:183-266. vs: nb tables of `len` values. Returns (final claims, point).
template <bool EXT>
static inline std::pair<std::vector<E>, std::vector<E>> prove_grand_product(
    const std::vector<const typename GpElem<EXT>::T*>& vs, size_t len, TranscriptW& tr, std::vector<E>* record = nullptr) {
    typedef GpElem<EXT> G;
    typedef typename G::T T;
    const size_t nb = vs.size();
    size_t nv = ilog2_u64(len);  // table has nv variables; bottom layer has nv-1
    // level arrays: A_0 = v (view), A_k = A_{k-1}[lo] * A_{k-1}[hi]   (Layer::bottom :310-315, Layer::up :332-354)
    std::vector<std::vector<std::vector<T>>> lev(nb);
    // one region for all tables and levels: (table, block of 4096 outputs) items per level, a barrier between levels
    for (size_t b = 0; b < nb; b++) {
        lev[b].resize(nv);  // lev[b][0] unused (view of vs[b])
        for (size_t k = 1; k < nv; k++) lev[b][k].resize(len >> k);
    }
    for (size_t k = 1; k < nv; k++) {
        const size_t h = len >> k, BLK = 4096, nblk = (h + BLK - 1) / BLK;
#pragma omp parallel for schedule(static) collapse(2) num_threads(orc_threads_for(nb * h, 8192))
        for (long long bb = 0; bb < (long long)nb; bb++)
            for (long long q = 0; q < (long long)nblk; q++) {
                const T* prev = k == 1 ? vs[bb] : lev[bb][k - 1].data();
                T* out = lev[bb][k].data();
                const size_t i0 = (size_t)q * BLK, i1 = std::min(h, i0 + BLK);
                for (size_t i = i0; i < i1; i++) out[i] = G::mul(prev[i], prev[i + h]);
            }
    }
    auto level_ptr = [&](size_t b, size_t k) -> const T* { return k == 0 ? vs[b] : lev[b][k].data(); };
    // root products (prover.rs:197-221): written, since claimed_v_0s are all None
    std::vector<E> claims(nb);
    for (size_t b = 0; b < nb; b++) {
        const T* top = level_ptr(b, nv - 1);
        claims[b] = G::lift(G::mul(top[0], top[1]));
        tr.write_e(claims[b]);
    }
    std::vector<E> x;
    for (size_t n = 0; n < nv; n++) {  // layer with num_vars n  <->  level nv-1-n
        size_t k = nv - 1 - n;
        size_t h = (size_t)1 << n;
        std::vector<E> evals;
        if (n == 0) {
            x.clear();
            for (size_t b = 0; b < nb; b++) {
                const T* a = level_ptr(b, k);
                evals.push_back(G::lift(a[0]));
                evals.push_back(G::lift(a[1]));
            }
        } else {
            E gamma = tr.squeeze();  // prover.rs:238
            ScFunc g{SC_GRANDPROD, n, powers_e(gamma, nb)};  // prover.rs:268-279
            E claim = e_zero();      // prover.rs:281-286
            for (size_t b = 0; b < nb; b++) claim = e_add(claim, e_mul(claims[b], g.pw[b]));
            std::vector<ScTable> Tb;
            for (size_t b = 0; b < nb; b++) {
                const T* a = level_ptr(b, k);
                Tb.push_back(G::table(a, h));
                Tb.push_back(G::table(a + h, h));
            }
            ScResult r = prove_sum_check(g, claim, std::move(Tb), tr, record);
            x = r.point;
            evals = r.evals;
        }
        tr.write_es(evals);      // prover.rs:257
        E mu = tr.squeeze();     // prover.rs:259
        for (size_t b = 0; b < nb; b++)  // prover.rs:288-294
            claims[b] = e_add(evals[2 * b], e_mul(mu, e_sub(evals[2 * b + 1], evals[2 * b])));
        x.push_back(mu);
    }
    return {claims, x};
}

// verifier.rs:178-235
static inline std::pair<std::vector<E>, std::vector<E>> verify_grand_product(size_t num_vars, size_t nb, TranscriptR& tr) {
    std::vector<E> claims = tr.read_es(nb);
    std::vector<E> x;
    for (size_t n = 0; n < num_vars; n++) {
        std::vector<E> evals;
        if (n == 0) {
            evals = tr.read_es(2 * nb);
            for (size_t b = 0; b < nb; b++)
                if (!e_eq(claims[b], e_mul(evals[2 * b], evals[2 * b + 1])))
                    throw std::runtime_error("InvalidSumCheck: unmatched sum check output");
            x.clear();
        } else {
            E gamma = tr.squeeze();
            std::vector<E> pw = powers_e(gamma, nb);
            E claim = e_zero();
            for (size_t b = 0; b < nb; b++) claim = e_add(claim, e_mul(claims[b], pw[b]));
            auto r = verify_sum_check(3, n, claim, tr);
            x = r.second;
            evals = tr.read_es(2 * nb);
        }
        E mu = tr.squeeze();
        for (size_t b = 0; b < nb; b++)
            claims[b] = e_add(evals[2 * b], e_mul(mu, e_sub(evals[2 * b + 1], evals[2 * b])));
        x.push_back(mu);
    }
    return {claims, x};
}

struct LassoNodeDef {
    size_t nu = 0;                     // num_vars
    std::vector<uint8_t> row_lookup;   // per row: index into pre.lookups (lasso.rs:35 `lookups`, resolved)
};

// chunks: dimension index ascending, memories of a chunk in ascending memory index (lasso.rs:303-336)
static inline std::vector<std::pair<size_t, std::vector<size_t>>> lasso_chunks(const LassoPre& pre) {
    std::map<size_t, std::vector<size_t>> cm;
    for (size_t m = 0; m < pre.num_memories; m++) cm[pre.mem_dim[m]].push_back(m);
    return std::vector<std::pair<size_t, std::vector<size_t>>>(cm.begin(), cm.end());
}

struct LassoClaim { std::vector<E> r; E value; };

// optional per-phase capture for kernel-level parity tests
struct LassoTrace {
    std::vector<E> collation_sums, gp1_sums, gp2_sums;
    LassoPolys* polys_out = nullptr;
};

// MemoryCheckingProver::new (prover.rs:35-89) + prove (prover.rs:158-181): multiset hashes of every memory in memory-GKR
// order, then the two grand products. EXT = false is the reference (gamma, tau truncated to base limb 0, prover.rs:38-39:
// base-field hash tables); EXT = true keeps them in E (ProtocolMode::ext_memcheck).
// QUIRK reproduced (lasso.rs:317-319): read_ts / final_cts are indexed by the CHUNK index, not the memory index.
template <bool EXT>
static inline std::pair<std::vector<E>, std::vector<E>> lasso_memory_checking(const LassoPre& pre, const LassoPolys& P, E gamma_e, E tau_e,
                                                                              TranscriptW& tr, LassoTrace* trace) {
    typedef GpElem<EXT> G;
    typedef typename G::T T;
    const size_t N = (size_t)1 << P.nu;
    T gamma, tau;
    if constexpr (EXT) { gamma = gamma_e; tau = tau_e; }
    else { gamma = e_limb0(gamma_e); tau = e_limb0(tau_e); }
    const T gamma2 = G::mul(gamma, gamma);
    // hash(a, v, t) = a + v*gamma + t*gamma^2 - tau (prover.rs:44) on small non-negative integers a, v, t
    auto lift_u = [](uint64_t x) -> T { if constexpr (EXT) return e_from_f(f_from_u64(x)); else return f_from_u64(x); };
    auto add = [](T a, T b) -> T { if constexpr (EXT) return e_add(a, b); else return f_add(a, b); };
    auto sub = [](T a, T b) -> T { if constexpr (EXT) return e_sub(a, b); else return f_sub(a, b); };
    auto hash = [&](uint64_t a, uint64_t v, uint64_t t) -> T {
        return sub(add(add(lift_u(a), G::mul(lift_u(v), gamma)), G::mul(lift_u(t), gamma2)), tau);
    };
    auto chunks = lasso_chunks(pre);
    std::vector<size_t> order, chunk_of;  // memory-GKR order
    for (auto& ch : chunks) for (size_t m : ch.second) { order.push_back(m); chunk_of.push_back(ch.first); }
    const size_t A = order.size();
    std::vector<std::vector<T>> init(A), rd(A), wr(A), fin(A);
#pragma omp parallel for schedule(dynamic, 1)
    for (long long ii = 0; ii < (long long)A; ii++) {
        size_t i = (size_t)ii, m = order[i], c = chunk_of[i];
        const std::vector<uint64_t>& dim = P.dims[c];
        const std::vector<uint64_t>& rts = P.read_cts[c];
        const std::vector<uint64_t>& fct = P.final_cts[c];
        const std::vector<uint64_t>& tab = pre.tables[pre.mem_subtable[m]];
        const std::vector<uint64_t>& ep = P.e_polys[m];
        init[i].resize(LASSO_M); fin[i].resize(LASSO_M); rd[i].resize(N); wr[i].resize(N);
        (void)dim; (void)rts; (void)ep;
        for (size_t a = 0; a < LASSO_M; a++) {
            init[i][a] = hash(a, tab[a], 0);
            fin[i][a] = hash(a, tab[a], fct[a]);
        }
    }
    {   // read / write hashes: (memory, block of rows) items, so that more threads than memories have work
        const size_t BLK = 16384, nblk = (N + BLK - 1) / BLK;
#pragma omp parallel for schedule(static) collapse(2)
        for (long long ii = 0; ii < (long long)A; ii++)
            for (long long q = 0; q < (long long)nblk; q++) {
                const size_t i = (size_t)ii, m = order[i], c = chunk_of[i];
                const uint64_t* dim = P.dims[c].data();
                const uint64_t* rts = P.read_cts[c].data();
                const uint64_t* ep = P.e_polys[m].data();
                T* rdp = rd[i].data();
                T* wrp = wr[i].data();
                const size_t j0 = (size_t)q * BLK, j1 = std::min(N, j0 + BLK);
                for (size_t j = j0; j < j1; j++) {
                    rdp[j] = hash(dim[j], ep[j], rts[j]);
                    wrp[j] = hash(dim[j], ep[j], rts[j] + 1);
                }
            }
    }
    std::vector<const T*> v1, v2;
    for (size_t i = 0; i < A; i++) v1.push_back(rd[i].data());
    for (size_t i = 0; i < A; i++) v1.push_back(wr[i].data());
    for (size_t i = 0; i < A; i++) v2.push_back(init[i].data());
    for (size_t i = 0; i < A; i++) v2.push_back(fin[i].data());
    auto g1 = prove_grand_product<EXT>(v1, N, tr, trace ? &trace->gp1_sums : nullptr);
    auto g2 = prove_grand_product<EXT>(v2, LASSO_M, tr, trace ? &trace->gp2_sums : nullptr);
    return {g1.second, g2.second};
}

// lasso.rs:57-114
static inline LassoClaim lasso_prove(const LassoPre& pre, const LassoNodeDef& node, const F* inputs,
                                     TranscriptW& tr, LassoTrace* trace = nullptr) {
    const size_t nu = node.nu, N = (size_t)1 << nu;
    double tq = omp_get_wtime();
    LassoPolys P = polynomialize(pre, nu, node.row_lookup, inputs);  // :64
    orc_times().add("lasso polynomialize", tq); tq = omp_get_wtime();
    // :77 assert inputs == lookup_outputs: RangeLookup::output is the identity (range.rs:230-232) -> holds
    std::vector<E> r = tr.squeeze_n(nu);  // :85
    // prove_collation_sum_check :254-288
    E claimed_sum = lasso_sum_check_claim(pre, P, node.row_lookup, r);
    tr.write_e(claimed_sum);  // :269
    orc_times().add("lasso claimed sum", tq); tq = omp_get_wtime();
    {
        std::vector<E> pw(pre.num_memories);  // distribute_powers(poly(0..alpha), M)  range.rs:197-204
        F c = f_one();
        for (size_t i = 0; i < pre.num_memories; i++) { pw[i] = e_from_f(c); c = f_mul(c, f_from_u64(LASSO_M)); }
        ScFunc g{SC_COLLATION, nu, pw};
        std::vector<FTab> ft(pre.num_memories);
        std::vector<ScTable> T;
        for (size_t m = 0; m < pre.num_memories; m++) { ft[m] = ftab_from_u64(P.e_polys[m]); T.push_back(ScTable::from_f(ft[m].p(), N)); }
        prove_sum_check(g, claimed_sum, std::move(T), tr, trace ? &trace->collation_sums : nullptr);  // :278-279, result dropped :97
    }
    orc_times().add("lasso collation sum-check", tq); tq = omp_get_wtime();
    E gamma_e = tr.squeeze(), tau_e = tr.squeeze();  // :99
    auto xy = tr.mode.ext_memcheck ? lasso_memory_checking<true>(pre, P, gamma_e, tau_e, tr, trace)
                                   : lasso_memory_checking<false>(pre, P, gamma_e, tau_e, tr, trace);
    const std::vector<E>& x = xy.first;
    const std::vector<E>& y = xy.second;
    orc_times().add("lasso memory checking", tq); tq = omp_get_wtime();
    {
        std::vector<E> eqx = eq_table(x), eqy = eq_table(y);
        for (auto& ch : lasso_chunks(pre)) {  // prover.rs:173-178, mod.rs:80-93
            size_t c = ch.first;
            tr.write_e(dot_eq_u64(eqx, P.dims[c].data(), N));
            tr.write_e(dot_eq_u64(eqx, P.read_cts[c].data(), N));
            tr.write_e(dot_eq_u64(eqy, P.final_cts[c].data(), LASSO_M));
            for (size_t m : ch.second) tr.write_e(dot_eq_u64(eqx, P.e_polys[m].data(), N));
        }
    }
    orc_times().add("lasso openings", tq);
    if (trace && trace->polys_out) *trace->polys_out = std::move(P);
    return LassoClaim{r, claimed_sum};  // :97,113
}

// lasso.rs:116-139 + verifier.rs:130-176 (+ verify_memories :61-95)
static inline LassoClaim lasso_verify(const LassoPre& pre, size_t nu, TranscriptR& tr) {
    std::vector<E> r = tr.squeeze_n(nu);
    E claimed_sum = tr.read_e();
    verify_sum_check(2, nu, claimed_sum, tr);  // result ignored (lasso.rs:129-130)
    E gamma_e = tr.squeeze(), tau_e = tr.squeeze();
    // verifier.rs hashes with the same truncated gamma, tau as the prover unless ProtocolMode::ext_memcheck keeps them in E
    const E gamma = tr.mode.ext_memcheck ? gamma_e : e_from_f(e_limb0(gamma_e));
    const E tau = tr.mode.ext_memcheck ? tau_e : e_from_f(e_limb0(tau_e));
    const E gamma2 = e_mul(gamma, gamma);
    auto hash = [&](E a, E v, E t) { return e_sub(e_add(e_add(a, e_mul(v, gamma)), e_mul(t, gamma2)), tau); };
    auto chunks = lasso_chunks(pre);
    size_t A = pre.num_memories;
    auto rw = verify_grand_product(nu, 2 * A, tr);
    auto ifr = verify_grand_product(LASSO_LOGM, 2 * A, tr);
    const std::vector<E>& y = ifr.second;
    size_t off = 0;
    for (auto& ch : chunks) {
        size_t nm = ch.second.size();
        E dim_x = tr.read_e(), rts_x = tr.read_e(), fct_y = tr.read_e();
        std::vector<E> e_xs = tr.read_es(nm);
        E id_y = e_zero();
        for (size_t i = 0; i < y.size(); i++) id_y = e_add(id_y, e_mul_f(y[i], f_from_u64(1ull << i)));
        for (size_t i = 0; i < nm; i++) {
            size_t m = ch.second[i];
            if (!e_eq(rw.first[off + i], hash(dim_x, e_xs[i], rts_x))) throw std::runtime_error("memory check: read hash mismatch");
            if (!e_eq(rw.first[A + off + i], hash(dim_x, e_xs[i], e_add(rts_x, e_one())))) throw std::runtime_error("memory check: write hash mismatch");
            E st_y = pre.subtables[pre.mem_subtable[m]].evaluate_mle(y);
            if (!e_eq(ifr.first[off + i], hash(id_y, st_y, e_zero()))) throw std::runtime_error("memory check: init hash mismatch");
            if (!e_eq(ifr.first[A + off + i], hash(id_y, st_y, fct_y))) throw std::runtime_error("memory check: final hash mismatch");
        }
        off += nm;
    }
    return LassoClaim{r, claimed_sum};
}

}  // namespace ORC_NS
