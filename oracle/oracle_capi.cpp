// ORACLE (test infrastructure): C entry points so that tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg can drive the CPU restatement through ctypes. Nothing in the product
// library (hyper-greco_amd/) includes, links or calls this file.
#include <cstring>
#include <string>
#include <omp.h>
#include "bfv.hpp"

using namespace orc;

extern "C" {

struct orc_params {
    uint64_t n, k, s_bound, e_bound, k1_bound;
    const uint64_t *r1_bounds, *r2_bounds, *qis, *k0is;
};
struct orc_inputs {
    const uint64_t *s, *e, *k1, *ais, *r1is, *r2is, *ct0is;
};

static BfvParams to_params(const orc_params* p) {
    BfvParams q;
    q.n = p->n; q.k = p->k; q.s_bound = p->s_bound; q.e_bound = p->e_bound; q.k1_bound = p->k1_bound;
    q.r1_bounds.assign(p->r1_bounds, p->r1_bounds + p->k);
    q.r2_bounds.assign(p->r2_bounds, p->r2_bounds + p->k);
    q.qis.assign(p->qis, p->qis + p->k);
    q.k0is.assign(p->k0is, p->k0is + p->k);
    return q;
}
static BfvInputs to_inputs(const BfvParams& q, const orc_inputs* in) {
    BfvInputs r;
    size_t SZ = (size_t)1 << q.log2_size(), PZ = (size_t)1 << q.n_log2();
    r.s.assign(in->s, in->s + SZ); r.e.assign(in->e, in->e + SZ); r.k1.assign(in->k1, in->k1 + SZ);
    for (size_t i = 0; i < q.k; i++) {
        r.ais.push_back(Values(in->ais + i * SZ, in->ais + (i + 1) * SZ));
        r.r1is.push_back(Values(in->r1is + i * SZ, in->r1is + (i + 1) * SZ));
    }
    r.r2is.assign(in->r2is, in->r2is + q.k * PZ);
    r.ct0is.assign(in->ct0is, in->ct0is + q.k * SZ);
    return r;
}
static void set_err(char* err, size_t cap, const std::string& s) {
    if (err && cap) { strncpy(err, s.c_str(), cap - 1); err[cap - 1] = 0; }
}

// ---- KAT helpers -------------------------------------------------------------------------------
void orc_keccak256(const uint8_t* in, size_t len, uint8_t* out32) { keccak256(in, len, out32); }
void orc_challenge_chain(size_t n, uint64_t* out) { ChallengeChain c; for (size_t i = 0; i < n; i++) out[i] = c.next_f(); }
void orc_f_binop(int op, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    for (size_t i = 0; i < n; i++) out[i] = op == 0 ? f_add(a[i], b[i]) : op == 1 ? f_sub(a[i], b[i]) : op == 2 ? f_mul(a[i], b[i]) : f_inv(a[i]);
}
void orc_e_binop(int op, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    for (size_t i = 0; i < n; i++) {
        E x{a[2 * i], a[2 * i + 1]}, y{b[2 * i], b[2 * i + 1]};
        E r = op == 0 ? e_add(x, y) : op == 1 ? e_sub(x, y) : op == 2 ? e_mul(x, y) : e_inv(x);
        out[2 * i] = r.c0; out[2 * i + 1] = r.c1;
    }
}
uint64_t orc_root_of_unity(size_t log2n) { return gl_root_of_unity(log2n); }
void orc_ntt(const uint64_t* in, size_t log2n, int inverse, uint64_t* out) {
    auto v = ntt(in, log2n, inverse != 0);
    memcpy(out, v.data(), v.size() * 8);
}
void orc_eq_table(const uint64_t* r, size_t n, uint64_t* out) {
    std::vector<E> rr(n);
    for (size_t i = 0; i < n; i++) rr[i] = E{r[2 * i], r[2 * i + 1]};
    auto t = eq_table(rr);
    memcpy(out, t.data(), t.size() * 16);
}
void orc_mle_eval_f(const uint64_t* tab, size_t nvars, const uint64_t* pt, uint64_t* out2) {
    std::vector<E> p(nvars);
    for (size_t i = 0; i < nvars; i++) p[i] = E{pt[2 * i], pt[2 * i + 1]};
    E v = mle_eval_f(tab, nvars, p.data());
    out2[0] = v.c0; out2[1] = v.c1;
}
void orc_fft_table(const uint64_t* r, size_t L, int inverse, uint64_t* out) {
    std::vector<E> rr(L);
    for (size_t i = 0; i < L; i++) rr[i] = E{r[2 * i], r[2 * i + 1]};
    auto t = fft_table(rr, L, inverse != 0);
    memcpy(out, t.data(), t.size() * 16);
}

// range.rs:293-331 identities: dense MLE of the materialized subtable vs evaluate_mle closed form.
// bound == 0 selects FullLimbSubtable. point: 16 E coordinates. Returns 1 if equal.
int orc_subtable_mle_identity(uint64_t bound, const uint64_t* point, uint64_t* dense_out2, uint64_t* closed_out2) {
    Subtable s = bound ? Subtable{false, bound, "bound_" + std::to_string(bound)} : Subtable{true, 0, "full"};
    std::vector<E> p(LASSO_LOGM);
    for (size_t i = 0; i < LASSO_LOGM; i++) p[i] = E{point[2 * i], point[2 * i + 1]};
    auto tab = s.materialize();
    E d = mle_eval_f(tab.data(), LASSO_LOGM, p.data());
    E c = s.evaluate_mle(p);
    dense_out2[0] = d.c0; dense_out2[1] = d.c1; closed_out2[0] = c.c0; closed_out2[1] = c.c1;
    return e_eq(d, c) ? 1 : 0;
}
uint64_t orc_subtable_cutoff(uint64_t bound) { return Subtable{false, bound, ""}.cutoff(); }

// Lasso memory map as text: "subtable_id@dim,..." ; lookups as "range_x:bits:m0/m1/..;..."
int orc_lasso_layout(const orc_params* p, char* out, size_t cap) {
    BfvParams q = to_params(p);
    LassoPre pre = bfv_setup(q);
    std::string s;
    for (size_t m = 0; m < pre.num_memories; m++) {
        if (m) s += ",";
        s += pre.subtables[pre.mem_subtable[m]].id + "@" + std::to_string(pre.mem_dim[m]);
    }
    s += "|";
    for (size_t l = 0; l < pre.lookups.size(); l++) {
        if (l) s += ";";
        s += pre.lookups[l].id + ":" + std::to_string(pre.lookups[l].total_bits()) + ":";
        for (size_t i = 0; i < pre.lookup_mems[l].size(); i++) { if (i) s += "/"; s += std::to_string(pre.lookup_mems[l][i]); }
    }
    if (s.size() + 1 > cap) return -1;
    strcpy(out, s.c_str());
    return (int)s.size();
}

// ---- circuit -----------------------------------------------------------------------------------
// Evaluates the circuit; copies out the lasso node's input table (2^nu) and the `sum` node output (k*2^L).
int orc_circuit_eval(const orc_params* p, const orc_inputs* in, uint64_t* lasso_in, size_t lasso_cap, uint64_t* sum_out,
                     uint64_t* info /* [nu, num_nodes, rows] */, char* err, size_t errcap) {
    try {
        BfvParams q = to_params(p);
        BfvInputs bi = to_inputs(q, in);
        BfvCircuit bc;
        bc.pre.reset(new LassoPre(bfv_setup(q)));
        bfv_configure(q, bc);
        auto vals = circuit_evaluate(bc.c, bfv_input_list(q, bi));
        size_t lin = bc.c.preds[bc.lasso_id][0];
        if (info) { info[0] = bc.c.nodes[bc.lasso_id].lasso.nu; info[1] = bc.c.nodes.size(); info[2] = bc.c.nodes[bc.lasso_id].lasso.row_lookup.size(); }
        if (lasso_in) { if (vals[lin].size() > lasso_cap) throw std::runtime_error("lasso_in buffer too small"); memcpy(lasso_in, vals[lin].data(), vals[lin].size() * 8); }
        if (sum_out) memcpy(sum_out, vals[bc.sum_id].data(), vals[bc.sum_id].size() * 8);
        return 0;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return -1; }
}

int orc_prove(const orc_params* p, const orc_inputs* in, int threads, uint8_t* proof, size_t cap, size_t* len,
              double* timings_ms /* [witness, prove] */, char* err, size_t errcap) {
    try {
        omp_set_num_threads(threads > 0 ? threads : 1);
        BfvParams q = to_params(p);
        BfvInputs bi = to_inputs(q, in);
        BfvProveTimings tm;
        std::vector<uint8_t> pr = bfv_prove(q, bi, &tm);
        if (timings_ms) { timings_ms[0] = tm.witness_ms; timings_ms[1] = tm.prove_ms; }
        *len = pr.size();
        if (pr.size() > cap) throw std::runtime_error("proof buffer too small");
        memcpy(proof, pr.data(), pr.size());
        return 0;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return -1; }
}

int orc_verify(const orc_params* p, const orc_inputs* in, int threads, const uint8_t* proof, size_t len, char* err, size_t errcap) {
    omp_set_num_threads(threads > 0 ? threads : 1);
    BfvParams q = to_params(p);
    BfvInputs bi = to_inputs(q, in);
    std::string e;
    bool ok = bfv_verify(q, bi, proof, len, &e);
    if (!ok) set_err(err, errcap, e);
    return ok ? 0 : -1;
}

// ---- node level: Lasso -------------------------------------------------------------------------
// Proves the Lasso node alone on `lasso_in` (2^nu values) with a FRESH transcript (challenge chain
// starts at H1). trace_* receive the raw per-round hypercube sums (t = 0,2[,3]) for kernel parity.
int orc_lasso_prove(const orc_params* p, const uint64_t* lasso_in, int threads, uint8_t* proof, size_t cap, size_t* len,
                    uint64_t* claim_out /* [nu*2 point | 2 value] */, char* err, size_t errcap) {
    try {
        omp_set_num_threads(threads > 0 ? threads : 1);
        BfvParams q = to_params(p);
        BfvCircuit bc;
        bc.pre.reset(new LassoPre(bfv_setup(q)));
        bfv_configure(q, bc);
        const LassoNodeDef& d = bc.c.nodes[bc.lasso_id].lasso;
        TranscriptW tr;
        LassoClaim lc = lasso_prove(*bc.pre, d, lasso_in, tr);
        *len = tr.stream.size();
        if (tr.stream.size() > cap) throw std::runtime_error("proof buffer too small");
        memcpy(proof, tr.stream.data(), tr.stream.size());
        if (claim_out) {
            for (size_t i = 0; i < lc.r.size(); i++) { claim_out[2 * i] = lc.r[i].c0; claim_out[2 * i + 1] = lc.r[i].c1; }
            claim_out[2 * lc.r.size()] = lc.value.c0; claim_out[2 * lc.r.size() + 1] = lc.value.c1;
        }
        return 0;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return -1; }
}

// The integer tables of the Lasso node (lasso.rs:157-250 polynomialize): limb indices, counters and subtable values are
// small non-negative integers, the same in every field - the BN254 oracle (oracle/bn254.py) starts from them.
// Call with dims == nullptr to query sizes. row_lookup[j] for j >= rows is 255. mem_cutoff: 65536 = full subtable.
int orc_lasso_polys(const orc_params* p, const uint64_t* lasso_in, size_t* nu_out, size_t* a_out, size_t* rows_out, uint64_t* dims,
                    uint64_t* read_cts, uint64_t* final_cts, uint64_t* e_polys, uint8_t* row_lookup, int* mem_dim, uint64_t* mem_cutoff,
                    char* err, size_t errcap) {
    try {
        BfvParams q = to_params(p);
        BfvCircuit bc;
        bc.pre.reset(new LassoPre(bfv_setup(q)));
        bfv_configure(q, bc);
        const LassoNodeDef& d = bc.c.nodes[bc.lasso_id].lasso;
        const LassoPre& pre = *bc.pre;
        const size_t N = (size_t)1 << d.nu, A = pre.num_memories;
        *nu_out = d.nu; *a_out = A; *rows_out = d.row_lookup.size();
        if (!dims) return 0;
        LassoPolys P = polynomialize(pre, d.nu, d.row_lookup, lasso_in);
        for (size_t c = 0; c < LASSO_C; c++) memcpy(dims + c * N, P.dims[c].data(), N * 8);
        for (size_t m = 0; m < A; m++) {
            memcpy(read_cts + m * N, P.read_cts[m].data(), N * 8);
            memcpy(final_cts + m * LASSO_M, P.final_cts[m].data(), LASSO_M * 8);
            memcpy(e_polys + m * N, P.e_polys[m].data(), N * 8);
            mem_dim[m] = (int)pre.mem_dim[m];
            const Subtable& st = pre.subtables[pre.mem_subtable[m]];
            mem_cutoff[m] = st.full ? LASSO_M : st.cutoff();
        }
        for (size_t j = 0; j < N; j++) row_lookup[j] = j < d.row_lookup.size() ? d.row_lookup[j] : 255;
        return 0;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return -1; }
}

int orc_lasso_verify(const orc_params* p, const uint8_t* proof, size_t len, char* err, size_t errcap) {
    try {
        BfvParams q = to_params(p);
        BfvCircuit bc;
        bc.pre.reset(new LassoPre(bfv_setup(q)));
        bfv_configure(q, bc);
        TranscriptR tr(proof, len);
        lasso_verify(*bc.pre, bc.c.nodes[bc.lasso_id].lasso.nu, tr);
        return 0;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return -1; }
}

// ---- kernel level: one sum-check over caller-supplied tables -------------------------------------
// kind: 0 collation, 1 grand product, 2 prodsum. tables: ntab pointers; is_base[i] says whether table i
// holds 2^nv u64 base values or 2^nv (c0,c1) pairs. pw: powers (E) for kinds 0/1. The challenge chain
// starts at position `chain_skip` (number of E challenges already consumed).
int orc_sumcheck(int kind, size_t nv, size_t ntab, const uint64_t* const* tables, const int* is_base, const uint64_t* pw, size_t npw,
                 const uint64_t* claim2, size_t chain_skip, int threads,
                 uint64_t* msgs /* nv*(d+1)*2 */, uint64_t* point /* nv*2 */, uint64_t* evals /* ntab*2 */, uint64_t* sums /* nv*d*2 raw */) {
    omp_set_num_threads(threads > 0 ? threads : 1);
    ScFunc g{(ScKind)kind, nv, {}};
    for (size_t i = 0; i < npw; i++) g.pw.push_back(E{pw[2 * i], pw[2 * i + 1]});
    std::vector<ScTable> T;
    size_t N = (size_t)1 << nv;
    for (size_t i = 0; i < ntab; i++) {
        if (is_base[i]) T.push_back(ScTable::from_f(tables[i], N));
        else { std::vector<E> v(N); memcpy(v.data(), tables[i], N * 16); T.push_back(ScTable::from_e(v)); }
    }
    TranscriptW tr;
    for (size_t i = 0; i < chain_skip; i++) tr.squeeze();
    std::vector<E> rec;
    ScResult r = prove_sum_check(g, E{claim2[0], claim2[1]}, std::move(T), tr, &rec);
    int d = g.degree();
    // stream holds nv*(d+1) E elements big-endian; return them decoded
    TranscriptR rd(tr.stream.data(), tr.stream.size());
    for (size_t i = 0; i < nv * (d + 1); i++) { E x = rd.read_e(); msgs[2 * i] = x.c0; msgs[2 * i + 1] = x.c1; }
    for (size_t i = 0; i < nv; i++) { point[2 * i] = r.point[i].c0; point[2 * i + 1] = r.point[i].c1; }
    for (size_t i = 0; i < ntab; i++) { evals[2 * i] = r.evals[i].c0; evals[2 * i + 1] = r.evals[i].c1; }
    if (sums) for (size_t i = 0; i < rec.size(); i++) { sums[2 * i] = rec[i].c0; sums[2 * i + 1] = rec[i].c1; }
    return 0;
}

}  // extern "C"
