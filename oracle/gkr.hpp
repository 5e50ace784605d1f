// ORACLE (test infrastructure). Restatement of the GKR engine the reference calls into:
// third-party crate `gkr` (github.com/nulltea/gkr-lasso, un-pinned: /root/reference/Cargo.toml:10,63-64;
// source NOT under /root/reference). PARITY UNPINNED for everything in this file; what is restated is
// the published algorithm each node type names, anchored on the reference's call sites:
//   Circuit / insert / connect / evaluate      sk_encryption_circuit.rs:102-290, :434-442
//   prove_gkr / verify_gkr                     sk_encryption_circuit.rs:455-457, :509-510
//   VanillaNode::new(arity, log2_sub_in, gates, reps), VanillaGate::{new,relay,mul,sum,constant}
//                                              sk_encryption_circuit.rs:99-284, :525-531   (Libra, eprint 2019/317)
//   FftNode::{forward,inverse}(log2)           sk_encryption_circuit.rs:224,249,251       (zkCNN, eprint 2021/673)
//   InputNode::new(log2, reps)                 sk_encryption_circuit.rs:122,126,147,358-360
// Conventions fixed here (and mirrored by the HIP product):
//   G1 node order: Kahn topological order with smallest NodeId first; proving visits it in reverse.
//   G2 a node with m > 1 output claims squeezes m challenges alpha_a and proves sum_a alpha_a out(r_a);
//      m == 1 squeezes nothing (alpha = 1).
//   G3 Vanilla (Libra): phase 1 over the left/linear operand x with bookkeeping tables T_i, writes
//      u_i = in_i(r_x); phase 2 (only if the node has mul gates) over the right operand y, writes
//      w_i = in_i(r_y). Data-parallel repetitions are the HIGH index bits of inputs and outputs.
//   G4 FFT (zkCNN): out(r) = sum_x F(r,x) in(x), F(r,x) = prod_b (1 - r_b + r_b w^(2^b x)),
//      natural order in and out, w = the 2^L-th root derived from ROOT_OF_UNITY = 7^((p-1)/2^32);
//      inverse uses w^-1 and the factor 2^-L. Writes u = in(r_x).
#pragma once
#include <vector>
#include <queue>
#include <functional>
#include <stdexcept>
#include "field.hpp"
#include "poly.hpp"
#include "sumcheck.hpp"
#include "lasso.hpp"

namespace ORC_NS {

// natural-order in/out NTT: out[z] = sum_x in[x] w^(xz); inverse: w^-1 and 1/N scaling
static inline std::vector<F> ntt(const F* in, size_t log2n, bool inverse) {
    size_t N = (size_t)1 << log2n;
    std::vector<F> a(N);
    for (size_t i = 0; i < N; i++) {
        size_t r = 0;
        for (size_t b = 0; b < log2n; b++) if (i >> b & 1) r |= (size_t)1 << (log2n - 1 - b);
        a[r] = in[i];
    }
    F w = f_root_of_unity(log2n);  // G4: the 2^log2n-th root derived from PrimeField::ROOT_OF_UNITY
    if (inverse) w = f_inv(w);
    for (size_t s = 1; s <= log2n; s++) {
        size_t m = (size_t)1 << s, h = m >> 1;
        F wm = f_pow(w, N >> s);
        std::vector<F> tw(h);
        tw[0] = f_one();
        for (size_t j = 1; j < h; j++) tw[j] = f_mul(tw[j - 1], wm);
        for (size_t k = 0; k < N; k += m)
            for (size_t j = 0; j < h; j++) {
                F t = f_mul(tw[j], a[k + j + h]), u = a[k + j];
                a[k + j] = f_add(u, t);
                a[k + j + h] = f_sub(u, t);
            }
    }
    if (inverse) {
        F ninv = f_inv(f_from_u64(N));
        for (auto& x : a) x = f_mul(x, ninv);
    }
    return a;
}

// F(r, x) for all x in [0, 2^L), O(N): prod_b (1 + r_b (w^(2^b x) - 1)), factor b depends on x mod 2^(L-b)
static inline std::vector<E> fft_table(const std::vector<E>& r, size_t L, bool inverse) {
    size_t N = (size_t)1 << L;
    F w = f_root_of_unity(L);
    if (inverse) w = f_inv(w);
    std::vector<F> W(N);
    W[0] = f_one();
    {   // powers of w: runs of 4096 in parallel, each started from w^(run start)
        const size_t RUN = 4096, nrun = (N + RUN - 1) / RUN;
        const F wrun = f_pow(w, RUN);
        std::vector<F> start(nrun);
        start[0] = f_one();
        for (size_t q = 1; q < nrun; q++) start[q] = f_mul(start[q - 1], wrun);
#pragma omp parallel for schedule(static) num_threads(orc_threads_for(N, 8192))
        for (long long q = 0; q < (long long)nrun; q++) {
            size_t i0 = (size_t)q * RUN, i1 = std::min(N, i0 + RUN);
            W[i0] = start[q];
            for (size_t i = i0 + 1; i < i1; i++) W[i] = f_mul(W[i - 1], w);
        }
    }
    std::vector<E> cur(1, inverse ? e_from_f(f_inv(f_from_u64(N))) : e_one());
    for (size_t bb = L; bb-- > 0;) {
        size_t sz = (size_t)1 << (L - bb);
        std::vector<E> nxt(sz);
#pragma omp parallel for schedule(static) num_threads(orc_threads_for(sz, 4096))
        for (long long xx = 0; xx < (long long)sz; xx++) {
            const size_t x = (size_t)xx;
            F wx = W[(x << bb) & (N - 1)];
            E f = e_add_f(e_mul_f(r[bb], f_sub(wx, f_one())), f_one());
            nxt[x] = e_mul(cur[x & (sz / 2 - 1)], f);
        }
        cur.swap(nxt);
    }
    return cur;
}

enum NodeKind { NK_INPUT, NK_VANILLA, NK_FFT, NK_LASSO };

// gate constants are non-negative integers (1, q_i, k0_i, range bounds): the same in every field, lifted with f_from_u64
struct LinTerm { uint32_t gate, in, j; uint64_t c; };
struct MulTerm { uint32_t gate, i0, j0, i1, j1; uint64_t c; };
struct ConstTerm { uint32_t gate; uint64_t c; };

struct EvalClaim { std::vector<E> point; E value; };

struct Node {
    NodeKind kind = NK_INPUT;
    // input
    size_t log2_size = 0;  // total log2 output size
    // vanilla
    size_t arity = 0, log2_sub_in = 0, log2_sub_out = 0, log2_reps = 0, num_gates = 0;
    std::vector<ConstTerm> w0;
    std::vector<LinTerm> lin;
    std::vector<MulTerm> mul;
    // fft
    bool inverse = false;
    // lasso
    LassoNodeDef lasso;
    const LassoPre* pre = nullptr;
};

static inline size_t ceil_log2(size_t x) { size_t l = 0; while (((size_t)1 << l) < x) l++; return l; }
static inline size_t exact_log2(size_t x) { size_t l = ceil_log2(x); if (((size_t)1 << l) != x) throw std::runtime_error("not a power of two"); return l; }

struct Circuit {
    std::vector<Node> nodes;
    std::vector<std::vector<size_t>> preds, succs;
    size_t insert(Node n) { nodes.push_back(std::move(n)); preds.push_back({}); succs.push_back({}); return nodes.size() - 1; }
    void connect(size_t from, size_t to) { preds[to].push_back(from); succs[from].push_back(to); }
    std::vector<size_t> topo() const {  // G1
        std::vector<size_t> indeg(nodes.size());
        for (size_t i = 0; i < nodes.size(); i++) indeg[i] = preds[i].size();
        std::priority_queue<size_t, std::vector<size_t>, std::greater<size_t>> q;
        for (size_t i = 0; i < nodes.size(); i++) if (!indeg[i]) q.push(i);
        std::vector<size_t> order;
        while (!q.empty()) {
            size_t u = q.top(); q.pop();
            order.push_back(u);
            for (size_t v : succs[u]) if (--indeg[v] == 0) q.push(v);
        }
        if (order.size() != nodes.size()) throw std::runtime_error("circuit has a cycle");
        return order;
    }
    size_t log2_out(size_t id) const {
        const Node& n = nodes[id];
        switch (n.kind) {
            case NK_INPUT: case NK_FFT: return n.log2_size;
            case NK_VANILLA: return n.log2_sub_out + n.log2_reps;
            default: return 0;
        }
    }
};

// node constructors mirroring the reference call shapes
static inline Node input_node(size_t log2_size, size_t num_reps) {
    Node n; n.kind = NK_INPUT; n.log2_size = log2_size + exact_log2(num_reps); return n;
}
static inline Node fft_node(size_t log2_size, bool inverse) {
    Node n; n.kind = NK_FFT; n.log2_size = log2_size; n.inverse = inverse; return n;
}
struct GateBuilder {  // VanillaNode::new(input_arity, log2_sub_input_size, gates, num_reps)
    Node n;
    GateBuilder(size_t arity, size_t log2_sub_in, size_t num_reps) {
        n.kind = NK_VANILLA; n.arity = arity; n.log2_sub_in = log2_sub_in; n.log2_reps = exact_log2(num_reps);
    }
    uint32_t g = 0;
    void relay(size_t i, size_t j) { n.lin.push_back({g++, (uint32_t)i, (uint32_t)j, 1}); }                 // VanillaGate::relay
    void relay_mul_const(size_t i, size_t j, uint64_t c) { n.lin.push_back({g++, (uint32_t)i, (uint32_t)j, c}); }  // :525-527
    void relay_add_const(size_t i, size_t j, uint64_t c) { n.w0.push_back({g, c}); n.lin.push_back({g++, (uint32_t)i, (uint32_t)j, 1}); }  // :529-531
    void constant(uint64_t c) { if (c) n.w0.push_back({g, c}); g++; }                                      // VanillaGate::constant
    void mul(size_t i0, size_t j0, size_t i1, size_t j1) { n.mul.push_back({g++, (uint32_t)i0, (uint32_t)j0, (uint32_t)i1, (uint32_t)j1, 1}); }
    void sum(const std::vector<std::pair<size_t, size_t>>& ws) { for (auto& w : ws) n.lin.push_back({g, (uint32_t)w.first, (uint32_t)w.second, 1}); g++; }
    Node finish() { n.num_gates = g; n.log2_sub_out = ceil_log2(g); return n; }
};

typedef std::vector<F> Values;

static inline Values vanilla_evaluate(const Node& n, const std::vector<const Values*>& in) {
    size_t G = (size_t)1 << n.log2_sub_out, S = (size_t)1 << n.log2_sub_in, R = (size_t)1 << n.log2_reps;
    Values out(G * R, f_zero());
#pragma omp parallel for schedule(static)
    for (long long rr = 0; rr < (long long)R; rr++) {
        size_t rep = (size_t)rr;
        F* o = out.data() + rep * G;
        for (auto& t : n.w0) o[t.gate] = f_add(o[t.gate], f_from_u64(t.c));
        for (auto& t : n.lin) o[t.gate] = f_add(o[t.gate], f_mul(f_from_u64(t.c), (*in[t.in])[rep * S + t.j]));
        for (auto& t : n.mul) o[t.gate] = f_add(o[t.gate], f_mul(f_from_u64(t.c), f_mul((*in[t.i0])[rep * S + t.j0], (*in[t.i1])[rep * S + t.j1])));
    }
    return out;
}

static inline std::vector<Values> circuit_evaluate(const Circuit& c, const std::vector<Values>& inputs) {
    std::vector<Values> vals(c.nodes.size());
    // inputs are assigned in NodeId order of the input nodes (chain_par! order, sk_encryption_circuit.rs:408)
    size_t next_in = 0;
    for (size_t id = 0; id < c.nodes.size(); id++)
        if (c.nodes[id].kind == NK_INPUT) {
            if (next_in >= inputs.size()) throw std::runtime_error("too few inputs");
            if (inputs[next_in].size() != ((size_t)1 << c.nodes[id].log2_size)) throw std::runtime_error("input size mismatch");
            vals[id] = inputs[next_in++];
        }
    for (size_t id : c.topo()) {
        const Node& n = c.nodes[id];
        std::vector<const Values*> in;
        for (size_t p : c.preds[id]) in.push_back(&vals[p]);
        switch (n.kind) {
            case NK_INPUT: break;
            case NK_VANILLA: vals[id] = vanilla_evaluate(n, in); break;
            case NK_FFT: vals[id] = ntt(in[0]->data(), n.log2_size, n.inverse); break;
            case NK_LASSO: vals[id] = Values(1, f_zero()); break;  // lasso.rs:53-55
        }
    }
    return vals;
}

static inline std::vector<E> combined_eq(const std::vector<EvalClaim>& cl, const std::vector<E>& alpha) {
    std::vector<E> eqc = eq_table(cl[0].point);
    if (cl.size() == 1 && e_eq(alpha[0], e_one())) return eqc;
    const long long n = (long long)eqc.size();
#pragma omp parallel for schedule(static) num_threads(orc_threads_for((size_t)n, 4096))
    for (long long i = 0; i < n; i++) eqc[i] = e_mul(eqc[i], alpha[0]);
    for (size_t a = 1; a < cl.size(); a++) {
        std::vector<E> t = eq_table(cl[a].point);
#pragma omp parallel for schedule(static) num_threads(orc_threads_for((size_t)n, 4096))
        for (long long i = 0; i < n; i++) eqc[i] = e_add(eqc[i], e_mul(t[i], alpha[a]));
    }
    return eqc;
}
static inline E combined_value(const std::vector<EvalClaim>& cl, const std::vector<E>& alpha) {
    E v = e_zero();
    for (size_t a = 0; a < cl.size(); a++) v = e_add(v, e_mul(cl[a].value, alpha[a]));
    return v;
}

struct VanillaUse { std::vector<bool> left, right; bool has_mul; };
static inline VanillaUse vanilla_use(const Node& n) {
    VanillaUse u{std::vector<bool>(n.arity, false), std::vector<bool>(n.arity, false), !n.mul.empty()};
    for (auto& t : n.lin) u.left[t.in] = true;
    for (auto& t : n.mul) { u.left[t.i0] = true; u.right[t.i1] = true; }
    return u;
}

// G3 prover. Returns per-input sub-claims.
static inline std::vector<std::vector<EvalClaim>> vanilla_prove(const Node& n, const std::vector<EvalClaim>& cl,
                                                                const std::vector<E>& alpha,
                                                                const std::vector<const Values*>& in, TranscriptW& tr) {
    size_t G = (size_t)1 << n.log2_sub_out, S = (size_t)1 << n.log2_sub_in, R = (size_t)1 << n.log2_reps;
    size_t nin = n.log2_sub_in + n.log2_reps;
    std::vector<E> eqc = combined_eq(cl, alpha);
    E claim = combined_value(cl, alpha);
    // Bookkeeping loops in parallel (field sums are exact, so any order gives the same tables): a "slot" is one (repetition,
    // band of input positions j); a slot's writes go to its own part of the tables, every slot scans the node's gate list. Nodes
    // with many repetitions split by repetition, single-repetition nodes (up to 1.7 M gates) by bands of j.
    const size_t bands = R >= 16 ? 1 : std::max<size_t>(1, std::min<size_t>(S / 1024, (size_t)omp_get_max_threads() / R));
    const long long slots = (long long)(R * bands);
    auto band_of = [&](size_t j) { return j * bands / S; };
    {
        E dsum = e_zero();
#pragma omp parallel num_threads(orc_threads_for(R * n.w0.size(), 4096))
        {
            E a = e_zero();
#pragma omp for nowait schedule(static)
            for (long long rr = 0; rr < (long long)R; rr++)
                for (auto& t : n.w0) a = e_add(a, e_mul_f(eqc[(size_t)rr * G + t.gate], f_from_u64(t.c)));
#pragma omp critical
            dsum = e_add(dsum, a);
        }
        claim = e_sub(claim, dsum);
    }
    VanillaUse use = vanilla_use(n);
    // phase 1 bookkeeping tables
    std::vector<std::vector<E>> T(n.arity);
    for (size_t i = 0; i < n.arity; i++) if (use.left[i]) T[i].assign(S * R, e_zero());
#pragma omp parallel for schedule(dynamic, 1) num_threads(orc_threads_for((size_t)slots * (n.lin.size() + n.mul.size()), 8192))
    for (long long sl = 0; sl < slots; sl++) {
        const size_t rep = (size_t)sl / bands, band = (size_t)sl % bands;
        for (auto& t : n.lin) {
            if (bands > 1 && band_of(t.j) != band) continue;
            E& d = T[t.in][rep * S + t.j]; d = e_add(d, e_mul_f(eqc[rep * G + t.gate], f_from_u64(t.c)));
        }
        for (auto& t : n.mul) {
            if (bands > 1 && band_of(t.j0) != band) continue;
            E& d = T[t.i0][rep * S + t.j0];
            d = e_add(d, e_mul_f(eqc[rep * G + t.gate], f_mul(f_from_u64(t.c), (*in[t.i1])[rep * S + t.j1])));
        }
    }
    std::vector<ScTable> tabs;
    std::vector<size_t> li;
    for (size_t i = 0; i < n.arity; i++) if (use.left[i]) {
        li.push_back(i);
        tabs.push_back(ScTable::from_f(in[i]->data(), S * R));
        tabs.push_back(ScTable::from_e(T[i]));
    }
    ScFunc g{SC_PRODSUM, nin, {}};
    ScResult r1 = prove_sum_check(g, claim, std::move(tabs), tr);
    std::vector<E> u(n.arity, e_zero());
    for (size_t k = 0; k < li.size(); k++) { u[li[k]] = r1.evals[2 * k]; tr.write_e(u[li[k]]); }
    std::vector<std::vector<EvalClaim>> sub(n.arity);
    for (size_t i : li) sub[i].push_back(EvalClaim{r1.point, u[i]});
    if (use.has_mul) {
        std::vector<E> eqx = eq_table(r1.point);
        E claim2 = r1.claim;
        {
            E dsum = e_zero();
#pragma omp parallel num_threads(orc_threads_for(R * n.lin.size(), 4096))
            {
                E a = e_zero();
#pragma omp for nowait schedule(static)
                for (long long rr = 0; rr < (long long)R; rr++) {
                    const size_t rep = (size_t)rr;
                    for (auto& t : n.lin) a = e_add(a, e_mul(u[t.in], e_mul(e_mul_f(eqc[rep * G + t.gate], f_from_u64(t.c)), eqx[rep * S + t.j])));
                }
#pragma omp critical
                dsum = e_add(dsum, a);
            }
            claim2 = e_sub(claim2, dsum);
        }
        std::vector<std::vector<E>> B(n.arity);
        for (size_t i = 0; i < n.arity; i++) if (use.right[i]) B[i].assign(S * R, e_zero());
#pragma omp parallel for schedule(dynamic, 1) num_threads(orc_threads_for((size_t)slots * n.mul.size(), 8192))
        for (long long sl = 0; sl < slots; sl++) {
            const size_t rep = (size_t)sl / bands, band = (size_t)sl % bands;
            for (auto& t : n.mul) {
                if (bands > 1 && band_of(t.j1) != band) continue;
                E& d = B[t.i1][rep * S + t.j1];
                d = e_add(d, e_mul(e_mul(e_mul_f(eqc[rep * G + t.gate], f_from_u64(t.c)), eqx[rep * S + t.j0]), u[t.i0]));
            }
        }
        std::vector<ScTable> tabs2;
        std::vector<size_t> ri;
        for (size_t i = 0; i < n.arity; i++) if (use.right[i]) {
            ri.push_back(i);
            tabs2.push_back(ScTable::from_f(in[i]->data(), S * R));
            tabs2.push_back(ScTable::from_e(B[i]));
        }
        ScResult r2 = prove_sum_check(g, claim2, std::move(tabs2), tr);
        for (size_t k = 0; k < ri.size(); k++) {
            E w = r2.evals[2 * k];
            tr.write_e(w);
            sub[ri[k]].push_back(EvalClaim{r2.point, w});
        }
    }
    return sub;
}

static inline std::vector<std::vector<EvalClaim>> vanilla_verify(const Node& n, const std::vector<EvalClaim>& cl,
                                                                 const std::vector<E>& alpha, TranscriptR& tr) {
    size_t G = (size_t)1 << n.log2_sub_out, S = (size_t)1 << n.log2_sub_in, R = (size_t)1 << n.log2_reps;
    size_t nin = n.log2_sub_in + n.log2_reps;
    std::vector<E> eqc = combined_eq(cl, alpha);
    E claim = combined_value(cl, alpha);
    for (size_t rep = 0; rep < R; rep++)
        for (auto& t : n.w0) claim = e_sub(claim, e_mul_f(eqc[rep * G + t.gate], f_from_u64(t.c)));
    VanillaUse use = vanilla_use(n);
    auto r1 = verify_sum_check(2, nin, claim, tr);
    std::vector<E> u(n.arity, e_zero());
    std::vector<std::vector<EvalClaim>> sub(n.arity);
    for (size_t i = 0; i < n.arity; i++) if (use.left[i]) { u[i] = tr.read_e(); sub[i].push_back(EvalClaim{r1.second, u[i]}); }
    std::vector<E> eqx = eq_table(r1.second);
    E lin_part = e_zero();
    for (size_t rep = 0; rep < R; rep++)
        for (auto& t : n.lin) lin_part = e_add(lin_part, e_mul(u[t.in], e_mul(e_mul_f(eqc[rep * G + t.gate], f_from_u64(t.c)), eqx[rep * S + t.j])));
    if (!use.has_mul) {
        if (!e_eq(r1.first, lin_part)) throw std::runtime_error("vanilla node: final evaluation mismatch");
        return sub;
    }
    E claim2 = e_sub(r1.first, lin_part);
    auto r2 = verify_sum_check(2, nin, claim2, tr);
    std::vector<E> w(n.arity, e_zero());
    for (size_t i = 0; i < n.arity; i++) if (use.right[i]) { w[i] = tr.read_e(); sub[i].push_back(EvalClaim{r2.second, w[i]}); }
    std::vector<E> eqy = eq_table(r2.second);
    E fin = e_zero();
    for (size_t rep = 0; rep < R; rep++)
        for (auto& t : n.mul)
            fin = e_add(fin, e_mul(e_mul(w[t.i1], u[t.i0]),
                                   e_mul(e_mul(e_mul_f(eqc[rep * G + t.gate], f_from_u64(t.c)), eqx[rep * S + t.j0]), eqy[rep * S + t.j1])));
    if (!e_eq(r2.first, fin)) throw std::runtime_error("vanilla node: phase-2 final evaluation mismatch");
    return sub;
}

static inline std::vector<E> fft_combined_table(const Node& n, const std::vector<EvalClaim>& cl, const std::vector<E>& alpha) {
    std::vector<E> Fc;
    for (size_t a = 0; a < cl.size(); a++) {
        std::vector<E> t = fft_table(cl[a].point, n.log2_size, n.inverse);
        if (a == 0 && cl.size() == 1) return t;
        if (a == 0) { Fc.assign(t.size(), e_zero()); }
#pragma omp parallel for schedule(static) num_threads(orc_threads_for(t.size(), 4096))
        for (long long i = 0; i < (long long)t.size(); i++) Fc[i] = e_add(Fc[i], e_mul(t[i], alpha[a]));
    }
    return Fc;
}

static inline std::vector<std::vector<EvalClaim>> fft_prove(const Node& n, const std::vector<EvalClaim>& cl,
                                                            const std::vector<E>& alpha, const Values& in, TranscriptW& tr) {
    std::vector<E> Fc = fft_combined_table(n, cl, alpha);
    E claim = combined_value(cl, alpha);
    std::vector<ScTable> tabs;
    tabs.push_back(ScTable::from_f(in.data(), in.size()));
    tabs.push_back(ScTable::from_e(Fc));
    ScFunc g{SC_PRODSUM, n.log2_size, {}};
    ScResult r = prove_sum_check(g, claim, std::move(tabs), tr);
    tr.write_e(r.evals[0]);
    return {{EvalClaim{r.point, r.evals[0]}}};
}

static inline std::vector<std::vector<EvalClaim>> fft_verify(const Node& n, const std::vector<EvalClaim>& cl,
                                                             const std::vector<E>& alpha, TranscriptR& tr) {
    E claim = combined_value(cl, alpha);
    auto r = verify_sum_check(2, n.log2_size, claim, tr);
    E u = tr.read_e();
    std::vector<E> Fc = fft_combined_table(n, cl, alpha);
    E fr = mle_eval_e(Fc.data(), n.log2_size, r.second.data());
    if (!e_eq(r.first, e_mul(u, fr))) throw std::runtime_error("fft node: final evaluation mismatch");
    return {{EvalClaim{r.second, u}}};
}

// G1/G2 driver. output_claims: (node id, claim). Returns the claims accumulated on every node
// (callers read the input nodes' entries).
static inline std::vector<std::vector<EvalClaim>> prove_gkr(const Circuit& c, const std::vector<Values>& vals,
                                                            const std::vector<std::pair<size_t, EvalClaim>>& output_claims,
                                                            TranscriptW& tr) {
    std::vector<std::vector<EvalClaim>> claims(c.nodes.size());
    for (auto& oc : output_claims) claims[oc.first].push_back(oc.second);
    std::vector<size_t> order = c.topo();
    for (size_t k = order.size(); k-- > 0;) {
        size_t id = order[k];
        const Node& n = c.nodes[id];
        if (n.kind == NK_INPUT) continue;
        const std::vector<EvalClaim>& cl = claims[id];
        if (cl.empty()) throw std::runtime_error("node without claim");
        std::vector<E> alpha = cl.size() > 1 ? tr.squeeze_n(cl.size()) : std::vector<E>{e_one()};
        std::vector<const Values*> in;
        for (size_t p : c.preds[id]) in.push_back(&vals[p]);
        std::vector<std::vector<EvalClaim>> sub;
        const double t0 = omp_get_wtime();
        switch (n.kind) {
            case NK_VANILLA: sub = vanilla_prove(n, cl, alpha, in, tr); orc_times().add("node vanilla (all)", t0); break;
            case NK_FFT: sub = fft_prove(n, cl, alpha, *in[0], tr); orc_times().add("node fft (all)", t0); break;
            case NK_LASSO: {
                LassoClaim lc = lasso_prove(*n.pre, n.lasso, in[0]->data(), tr);
                sub = {{EvalClaim{lc.r, lc.value}}};
                orc_times().add("node lasso (all)", t0);
                break;
            }
            default: break;
        }
        for (size_t i = 0; i < c.preds[id].size(); i++)
            for (auto& s : sub[i]) claims[c.preds[id][i]].push_back(s);
    }
    orc_times().dump();
    return claims;
}

static inline std::vector<std::vector<EvalClaim>> verify_gkr(const Circuit& c,
                                                             const std::vector<std::pair<size_t, EvalClaim>>& output_claims,
                                                             TranscriptR& tr) {
    std::vector<std::vector<EvalClaim>> claims(c.nodes.size());
    for (auto& oc : output_claims) claims[oc.first].push_back(oc.second);
    std::vector<size_t> order = c.topo();
    for (size_t k = order.size(); k-- > 0;) {
        size_t id = order[k];
        const Node& n = c.nodes[id];
        if (n.kind == NK_INPUT) continue;
        const std::vector<EvalClaim>& cl = claims[id];
        if (cl.empty()) throw std::runtime_error("node without claim");
        std::vector<E> alpha = cl.size() > 1 ? tr.squeeze_n(cl.size()) : std::vector<E>{e_one()};
        std::vector<std::vector<EvalClaim>> sub;
        switch (n.kind) {
            case NK_VANILLA: sub = vanilla_verify(n, cl, alpha, tr); break;
            case NK_FFT: sub = fft_verify(n, cl, alpha, tr); break;
            case NK_LASSO: {
                LassoClaim lc = lasso_verify(*n.pre, n.lasso.nu, tr);
                sub = {{EvalClaim{lc.r, lc.value}}};
                break;
            }
            default: break;
        }
        for (size_t i = 0; i < c.preds[id].size(); i++)
            for (auto& s : sub[i]) claims[c.preds[id][i]].push_back(s);
    }
    return claims;
}

}  // namespace ORC_NS
