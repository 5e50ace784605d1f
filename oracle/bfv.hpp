// ORACLE (test infrastructure). CPU restatement of the BFV secret-key-encryption circuit and its
// prove / verify drivers, following /root/reference/bfv-gkr/src/sk_encryption_circuit.rs:
//   configure :86-293 and :351-363   prove :417-460   verify :462-517   ct0is_log2_size :519-522
// Inputs arrive already laid out as get_inputs (:365-415) produces them (the layout code lives in the
// product's host library and is tested against the reference's JSON fixtures).
#pragma once
#include <vector>
#include <memory>
#include "gkr.hpp"

namespace ORC_NS {

struct BfvParams {
    size_t n = 0, k = 0;
    uint64_t s_bound = 0, e_bound = 0, k1_bound = 0;
    std::vector<uint64_t> r1_bounds, r2_bounds, qis, k0is;
    size_t n_log2() const { return exact_log2(n); }
    size_t log2_size() const { return n_log2() + 1; }                    // :81-83
    size_t ct0is_log2_size() const { return log2_size() + exact_log2(k); }  // :519-522
};

struct BfvInputs {  // get_inputs :365-415 (field elements)
    Values s, e, k1;              // 2^L each
    std::vector<Values> ais, r1is;  // k x 2^L
    Values r2is;                  // k * 2^P
    Values ct0is;                 // k * 2^L
};

struct BfvCircuit {
    Circuit c;
    std::unique_ptr<LassoPre> pre;
    size_t lasso_id = 0, sum_id = 0;
};

// setup :319-349
static inline LassoPre bfv_setup(const BfvParams& p) {
    std::vector<uint64_t> b = {p.s_bound * 2 + 1, p.e_bound * 2 + 1, p.k1_bound * 2 + 1};
    for (size_t i = 0; i < p.k; i++) b.push_back(p.r1_bounds[i] * 2 + 1);
    for (size_t i = 0; i < p.k; i++) b.push_back(p.r2_bounds[i] * 2 + 1);
    return LassoPre::preprocess(b);
}

static inline void bfv_configure(const BfvParams& p, BfvCircuit& bc) {
    Circuit& c = bc.c;
    const size_t P = p.n_log2(), L = p.log2_size(), k = p.k;
    const size_t SZ = (size_t)1 << L;
    size_t s = c.insert(input_node(L, 1)), e = c.insert(input_node(L, 1)), k1 = c.insert(input_node(L, 1));  // :358-360
    size_t es, k1kis;
    { GateBuilder g(1, L, 1); for (size_t i = 0; i < k; i++) for (size_t j = 0; j < SZ; j++) g.relay(0, j); es = c.insert(g.finish()); }  // :97-103
    { GateBuilder g(1, L, 1); for (size_t i = 0; i < k; i++) for (size_t j = 0; j < SZ; j++) g.relay_mul_const(0, j, p.k0is[i]); k1kis = c.insert(g.finish()); }  // :105-115
    c.connect(e, es); c.connect(k1, k1kis);  // :117-120
    std::vector<size_t> ais, r1is;
    for (size_t i = 0; i < k; i++) ais.push_back(c.insert(input_node(L, 1)));   // :122-124
    for (size_t i = 0; i < k; i++) r1is.push_back(c.insert(input_node(L, 1)));  // :126-128
    size_t r1iqis;
    { GateBuilder g(k, L, 1); for (size_t i = 0; i < k; i++) for (size_t j = 0; j < SZ; j++) g.relay_mul_const(i, j, p.qis[i]); r1iqis = c.insert(g.finish()); }  // :130-141
    for (size_t i = 0; i < k; i++) c.connect(r1is[i], r1iqis);  // :143-145
    size_t r2is = c.insert(input_node(P, k));                   // :147
    size_t r2l = P + exact_log2(k);                             // :149, :295-297
    std::vector<size_t> chunks;
    for (size_t st = 0; st < ((size_t)1 << r2l); st += SZ) {    // :150-161
        GateBuilder g(1, r2l, 1);
        size_t en = std::min(st + SZ, (size_t)1 << r2l);
        for (size_t j = st; j < en; j++) g.relay(0, j);
        for (size_t j = en - st; j < SZ; j++) g.constant(0);
        size_t node = c.insert(g.finish());
        c.connect(r2is, node);
        chunks.push_back(node);
    }
    size_t lasso_in;
    {   // :163-181
        std::vector<uint64_t> bounds;
        for (size_t i = 0; i < k; i++) bounds.push_back(p.r1_bounds[i]);
        for (size_t i = 0; i < chunks.size(); i++) bounds.push_back(p.r2_bounds[0]);
        bounds.push_back(p.s_bound); bounds.push_back(p.e_bound); bounds.push_back(p.k1_bound);
        GateBuilder g(chunks.size() + k + 3, L, 1);
        for (size_t i = 0; i < bounds.size(); i++) for (size_t j = 0; j < SZ; j++) g.relay_add_const(i, j, bounds[i]);
        lasso_in = c.insert(g.finish());
    }
    {   // :182-210
        size_t r2i_l = k == 1 ? L : P;
        LassoNodeDef d;
        const LassoPre& pre = *bc.pre;
        for (size_t i = 0; i < k; i++) { uint8_t id = (uint8_t)pre.lookup_index(p.r1_bounds[i] * 2 + 1); d.row_lookup.insert(d.row_lookup.end(), SZ, id); }
        for (size_t i = 0; i < k; i++) { uint8_t id = (uint8_t)pre.lookup_index(p.r2_bounds[i] * 2 + 1); d.row_lookup.insert(d.row_lookup.end(), (size_t)1 << r2i_l, id); }
        d.row_lookup.insert(d.row_lookup.end(), SZ, (uint8_t)pre.lookup_index(p.s_bound * 2 + 1));
        d.row_lookup.insert(d.row_lookup.end(), SZ, (uint8_t)pre.lookup_index(p.e_bound * 2 + 1));
        d.row_lookup.insert(d.row_lookup.end(), SZ, (uint8_t)pre.lookup_index(p.k1_bound * 2 + 1));
        d.nu = ceil_log2(d.row_lookup.size());
        Node n; n.kind = NK_LASSO; n.lasso = d; n.pre = bc.pre.get();
        bc.lasso_id = c.insert(n);
    }
    for (size_t i = 0; i < k; i++) c.connect(r1is[i], lasso_in);    // :211-213
    for (size_t ch : chunks) c.connect(ch, lasso_in);               // :215-217
    c.connect(s, lasso_in); c.connect(e, lasso_in); c.connect(k1, lasso_in);  // :219-222
    c.connect(lasso_in, bc.lasso_id);
    size_t s_eval = c.insert(fft_node(L, false));  // :224-225
    c.connect(s, s_eval);
    size_t s_eval_copy;
    { GateBuilder g(1, L, 1); for (size_t j = 0; j < SZ; j++) g.relay(0, j); s_eval_copy = c.insert(g.finish()); }  // :227-235
    c.connect(s_eval, s_eval_copy);
    size_t sai_par;
    { GateBuilder g(k, L, 1); for (size_t i = 0; i < k; i++) for (size_t j = 0; j < SZ; j++) g.relay(i, j); sai_par = c.insert(g.finish()); }  // :237-243
    for (size_t i = 0; i < k; i++) {  // :245-260
        size_t ai_eval = c.insert(fft_node(L, false));
        size_t sai_eval;
        { GateBuilder g(2, L, 1); for (size_t j = 0; j < SZ; j++) g.mul(0, j, 1, j); sai_eval = c.insert(g.finish()); }
        size_t sai = c.insert(fft_node(L, true));
        c.connect(ais[i], ai_eval);
        c.connect(s_eval_copy, sai_eval); c.connect(ai_eval, sai_eval);
        c.connect(sai_eval, sai);
        c.connect(sai, sai_par);
    }
    size_t cyclo;
    {   // :262-278
        size_t r2sz = ((size_t)1 << P) - 1;
        GateBuilder g(1, P, k);
        for (size_t j = 0; j < r2sz; j++) g.relay(0, j);
        g.constant(0);
        for (size_t j = 0; j < r2sz; j++) g.relay(0, j);
        g.constant(0);
        cyclo = c.insert(g.finish());
    }
    {   // :280-285
        GateBuilder g(5, L, k);
        for (size_t j = 0; j < SZ; j++) g.sum({{0, j}, {1, j}, {2, j}, {3, j}, {4, j}});
        bc.sum_id = c.insert(g.finish());
    }
    c.connect(r2is, cyclo);  // :287-290
    c.connect(sai_par, bc.sum_id); c.connect(es, bc.sum_id); c.connect(k1kis, bc.sum_id);
    c.connect(r1iqis, bc.sum_id); c.connect(cyclo, bc.sum_id);
}

static inline std::vector<Values> bfv_input_list(const BfvParams& p, const BfvInputs& in) {
    std::vector<Values> v = {in.s, in.e, in.k1};  // chain_par! order :408
    for (size_t i = 0; i < p.k; i++) v.push_back(in.ais[i]);
    for (size_t i = 0; i < p.k; i++) v.push_back(in.r1is[i]);
    v.push_back(in.r2is);
    return v;
}

struct BfvProveTimings { double witness_ms = 0, prove_ms = 0; };

static inline double now_ms();

// prove :417-460
static inline std::vector<uint8_t> bfv_prove(const BfvParams& p, const BfvInputs& in, BfvProveTimings* tm = nullptr, ProtocolMode mode = ProtocolMode());
static inline bool bfv_verify(const BfvParams& p, const BfvInputs& in, const uint8_t* proof, size_t len, std::string* err, ProtocolMode mode = ProtocolMode());

}  // namespace ORC_NS

#include <chrono>
namespace ORC_NS {
static inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static inline std::vector<uint8_t> bfv_prove(const BfvParams& p, const BfvInputs& in, BfvProveTimings* tm, ProtocolMode mode) {
    BfvCircuit bc;
    bc.pre.reset(new LassoPre(bfv_setup(p)));
    TranscriptW tr(mode);  // :431
    bfv_configure(p, bc);  // :433-437
    double t0 = now_ms();
    std::vector<Values> vals = circuit_evaluate(bc.c, bfv_input_list(p, in));  // :442
    std::vector<E> point = tr.squeeze_n(p.ct0is_log2_size());                  // :445
    E value = mle_eval_f(in.ct0is.data(), point.size(), point.data());         // :446
    std::vector<std::pair<size_t, EvalClaim>> oc = {{bc.lasso_id, EvalClaim{{}, e_zero()}}, {bc.sum_id, EvalClaim{point, value}}};  // :450
    double t1 = now_ms();
    prove_gkr(bc.c, vals, oc, tr);  // :455-457
    double t2 = now_ms();
    if (tm) { tm->witness_ms = t1 - t0; tm->prove_ms = t2 - t1; }
    return tr.stream;  // :459
}

static inline bool bfv_verify(const BfvParams& p, const BfvInputs& in, const uint8_t* proof, size_t len, std::string* err, ProtocolMode mode) {
    try {
        BfvCircuit bc;
        bc.pre.reset(new LassoPre(bfv_setup(p)));
        TranscriptR tr(proof, len, mode);  // :478
        std::vector<E> point = tr.squeeze_n(p.ct0is_log2_size());  // :482
        E value = mle_eval_f(in.ct0is.data(), point.size(), point.data());  // :495
        bfv_configure(p, bc);  // :503-507
        std::vector<std::pair<size_t, EvalClaim>> oc = {{bc.lasso_id, EvalClaim{{}, e_zero()}}, {bc.sum_id, EvalClaim{point, value}}};
        auto claims = verify_gkr(bc.c, oc, tr);  // :509-510
        std::vector<Values> inputs = bfv_input_list(p, in);
        size_t k = 0;
        for (size_t id = 0; id < bc.c.nodes.size(); id++) {  // :512-516
            if (bc.c.nodes[id].kind != NK_INPUT) continue;
            for (auto& cl : claims[id]) {
                E v = mle_eval_f(inputs[k].data(), cl.point.size(), cl.point.data());
                if (!e_eq(v, cl.value)) throw std::runtime_error("input claim mismatch at input " + std::to_string(k));
            }
            k++;
        }
        return true;
    } catch (const std::exception& ex) {
        if (err) *err = ex.what();
        return false;
    }
}

}  // namespace ORC_NS
