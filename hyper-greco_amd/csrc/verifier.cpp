// CPU verifier = BfvEncrypt::verify [REF bfv-gkr/src/sk_encryption_circuit.rs:462-517] with
// LassoNode::verify_claim_reduction [REF lasso/src/lasso.rs:116-139], MemoryCheckingVerifier
// [REF lasso/src/memory_checking/verifier.rs:61-95,130-235] and the sub-table MLE closed forms
// [REF lasso/src/table/range.rs:19-26,74-112]. Verification is a host-side job in the reference too;
// this file is the product's own implementation (independent of the test-only CPU restatement).
// The Vanilla / FFT node checks mirror the prover's conventions (the external `gkr`
// crate is unpinned).
#include <cstring>
#include <functional>
#include "host.hpp"

namespace hg {

namespace {

struct Reject : std::runtime_error { using std::runtime_error::runtime_error; };

struct ProofReader {
    const uint8_t* p; size_t len; size_t pos = 0;
    u64 read_f() {  // read_felt (transcript.rs:162-170): 8 bytes big-endian, canonical
        if (pos + 8 > len) throw Reject("proof: unexpected end of stream");
        u64 a = 0;
        for (int i = 0; i < 8; i++) a = (a << 8) | p[pos + i];
        pos += 8;
        if (a >= GL_P) throw Reject("proof: invalid field element");
        return a;
    }
    E2 read_e() { u64 a = read_f(); u64 b = read_f(); return e2(a, b); }
    std::vector<E2> read_es(size_t n) { std::vector<E2> v(n); for (auto& x : v) x = read_e(); return v; }
};

struct Claim { std::vector<E2> point; E2 value; };

std::vector<E2> eq_table(const std::vector<E2>& r) {
    std::vector<E2> t((size_t)1 << r.size());
    t[0] = e2_one();
    size_t s = 1;
    for (size_t i = 0; i < r.size(); i++) {
        for (size_t j = 0; j < s; j++) { E2 hi = e2_mul(t[j], r[i]); t[j + s] = hi; t[j] = e2_sub(t[j], hi); }
        s <<= 1;
    }
    return t;
}
E2 mle_eval(const u64* tab, const std::vector<E2>& pt) {
    std::vector<E2> eq = eq_table(pt);
    u64 c0 = 0, c1 = 0;
    for (size_t j = 0; j < eq.size(); j++) {
        u64 v = tab[j];
        if (v) { c0 = gl_add(c0, gl_mul(eq[j].c0, v)); c1 = gl_add(c1, gl_mul(eq[j].c1, v)); }
    }
    return e2(c0, c1);
}
E2 horner(const std::vector<E2>& c, E2 x) {
    E2 r = e2_zero();
    for (size_t i = c.size(); i-- > 0;) r = e2_add(e2_mul(r, x), c[i]);
    return r;
}

struct Verifier {
    ProofReader rd;
    ChallengeSource ch;

    // verify_sum_check: d+1 coefficients per round, 2 c0 + c1 + .. + cd == claim, claim <- p(r)
    std::pair<E2, std::vector<E2>> sumcheck(int deg, int nvars, E2 claim) {
        std::vector<E2> point;
        for (int i = 0; i < nvars; i++) {
            std::vector<E2> c = rd.read_es(deg + 1);
            E2 s = e2_dbl(c[0]);
            for (int k = 1; k <= deg; k++) s = e2_add(s, c[k]);
            if (!e2_eq(s, claim)) throw Reject("InvalidSumCheck: round polynomial does not match the running claim");
            E2 r = ch.squeeze();
            claim = horner(c, r);
            point.push_back(r);
        }
        return {claim, point};
    }

    // verify_grand_product (verifier.rs:178-235)
    std::pair<std::vector<E2>, std::vector<E2>> grand_product(int num_vars, int nb) {
        std::vector<E2> claims = rd.read_es(nb);
        std::vector<E2> x;
        for (int n = 0; n < num_vars; n++) {
            std::vector<E2> evals;
            if (n == 0) {
                evals = rd.read_es(2 * (size_t)nb);
                for (int b = 0; b < nb; b++)
                    if (!e2_eq(claims[b], e2_mul(evals[2 * b], evals[2 * b + 1]))) throw Reject("InvalidSumCheck: unmatched sum check output");
                x.clear();
            } else {
                E2 gamma = ch.squeeze();
                E2 claim = e2_zero(), g = e2_one();
                for (int b = 0; b < nb; b++) { claim = e2_add(claim, e2_mul(claims[b], g)); g = e2_mul(g, gamma); }
                auto r = sumcheck(3, n, claim);
                x = r.second;
                evals = rd.read_es(2 * (size_t)nb);
            }
            E2 mu = ch.squeeze();
            for (int b = 0; b < nb; b++) claims[b] = e2_add(evals[2 * b], e2_mul(mu, e2_sub(evals[2 * b + 1], evals[2 * b])));
            x.push_back(mu);
        }
        return {claims, x};
    }

    // sub-table MLE closed forms (range.rs:19-26, 74-112)
    static E2 subtable_mle(u64 bound, const std::vector<E2>& y) {
        E2 result = e2_zero();
        const size_t b = y.size();
        if (bound == 0) {
            for (size_t i = 0; i < b; i++) result = e2_add(result, e2_mul_f(y[i], 1ULL << i));
            return result;
        }
        int bits = 63 - __builtin_clzll(bound);
        u64 cutoff = (1ULL << (bits % LassoPlan::LOGM)) + bound % (1ULL << LassoPlan::LOGM);
        int cl2 = 63 - __builtin_clzll(cutoff);
        u64 g_base = 1ULL << cl2, extra = cutoff - g_base;
        for (size_t i = 0; i < b; i++) {
            if ((int)i < cl2) { result = e2_add(result, e2_mul_f(y[i], 1ULL << i)); continue; }
            E2 g_value = e2_zero();
            if ((int)i == cl2)
                for (u64 k = 0; k < extra; k++) {
                    E2 term = e2(gl_from_u64(g_base + k), 0);
                    for (int j = 0; j < cl2; j++) term = e2_mul(term, (k >> j) & 1 ? y[j] : e2_sub(e2_one(), y[j]));
                    g_value = e2_add(g_value, term);
                }
            result = e2_add(e2_mul(e2_sub(e2_one(), y[i]), result), e2_mul(y[i], g_value));
        }
        return result;
    }

    Claim lasso(const LassoPlan& lp) {  // lasso.rs:116-139
        std::vector<E2> r = ch.squeeze_n(lp.nu);
        E2 claimed = rd.read_e();
        sumcheck(2, lp.nu, claimed);  // collation: final evaluation not checked by the reference either (lasso.rs:129-130)
        E2 ge = ch.squeeze(), te = ch.squeeze();
        const u64 gamma = ge.c0, tau = te.c0, gamma2 = gl_mul(gamma, gamma);  // verifier.rs:139-140
        auto hash = [&](E2 a, E2 v, E2 t) { return e2_sub_f(e2_add(e2_add(a, e2_mul_f(v, gamma)), e2_mul_f(t, gamma2)), tau); };
        const int A = lp.alpha;
        auto rw = grand_product(lp.nu, 2 * A);
        auto ifr = grand_product(LassoPlan::LOGM, 2 * A);
        const std::vector<E2>& y = ifr.second;
        E2 id_y = e2_zero();
        for (size_t i = 0; i < y.size(); i++) id_y = e2_add(id_y, e2_mul_f(y[i], 1ULL << i));
        int off = 0;
        for (auto& chk : lp.chunks) {  // verify_memories (verifier.rs:61-95)
            const int nm = (int)chk.second.size();
            E2 dim_x = rd.read_e(), rts_x = rd.read_e(), fct_y = rd.read_e();
            std::vector<E2> e_xs = rd.read_es(nm);
            for (int i = 0; i < nm; i++) {
                int m = chk.second[i];
                if (!e2_eq(rw.first[off + i], hash(dim_x, e_xs[i], rts_x))) throw Reject("memory check: read hash mismatch");
                if (!e2_eq(rw.first[A + off + i], hash(dim_x, e_xs[i], e2_add_f(rts_x, 1)))) throw Reject("memory check: write hash mismatch");
                E2 st = subtable_mle(lp.subtable_bound[lp.mems[m].subtable], y);
                if (!e2_eq(ifr.first[off + i], hash(id_y, st, e2_zero()))) throw Reject("memory check: init hash mismatch");
                if (!e2_eq(ifr.first[A + off + i], hash(id_y, st, fct_y))) throw Reject("memory check: final hash mismatch");
            }
            off += nm;
        }
        return Claim{r, claimed};
    }

    static std::vector<E2> combined_eq(const std::vector<Claim>& cl, const std::vector<E2>& alpha) {
        std::vector<E2> eqc = eq_table(cl[0].point);
        if (cl.size() == 1) return eqc;
        for (auto& x : eqc) x = e2_mul(x, alpha[0]);
        for (size_t a = 1; a < cl.size(); a++) {
            std::vector<E2> t = eq_table(cl[a].point);
            for (size_t i = 0; i < t.size(); i++) eqc[i] = e2_add(eqc[i], e2_mul(t[i], alpha[a]));
        }
        return eqc;
    }

    std::vector<std::vector<Claim>> vanilla(const HNode& n, const std::vector<Claim>& cl, const std::vector<E2>& alpha) {
        const size_t G = (size_t)1 << n.log2_sub_out, S = (size_t)1 << n.log2_sub_in, R = (size_t)1 << n.log2_reps;
        const int nin = n.log2_sub_in + n.log2_reps;
        std::vector<E2> eqc = combined_eq(cl, alpha);
        E2 claim = e2_zero();
        for (size_t a = 0; a < cl.size(); a++) claim = e2_add(claim, e2_mul(cl[a].value, alpha[a]));
        for (size_t rep = 0; rep < R; rep++) for (auto& t : n.w0) claim = e2_sub(claim, e2_mul_f(eqc[rep * G + t.gate], t.c));
        auto r1 = sumcheck(2, nin, claim);
        std::vector<E2> u(n.arity, e2_zero());
        std::vector<std::vector<Claim>> sub(n.arity);
        for (int i = 0; i < n.arity; i++) if (n.left_use[i]) { u[i] = rd.read_e(); sub[i].push_back(Claim{r1.second, u[i]}); }
        std::vector<E2> eqx = eq_table(r1.second);
        E2 lin = e2_zero();
        for (size_t rep = 0; rep < R; rep++)
            for (auto& t : n.lin) lin = e2_add(lin, e2_mul(u[t.in], e2_mul(e2_mul_f(eqc[rep * G + t.gate], t.c), eqx[rep * S + t.j])));
        if (n.mul.empty()) {
            if (!e2_eq(r1.first, lin)) throw Reject("vanilla node: final evaluation mismatch");
            return sub;
        }
        auto r2 = sumcheck(2, nin, e2_sub(r1.first, lin));
        std::vector<E2> w(n.arity, e2_zero());
        for (int i = 0; i < n.arity; i++) if (n.right_use[i]) { w[i] = rd.read_e(); sub[i].push_back(Claim{r2.second, w[i]}); }
        std::vector<E2> eqy = eq_table(r2.second);
        E2 fin = e2_zero();
        for (size_t rep = 0; rep < R; rep++)
            for (auto& t : n.mul)
                fin = e2_add(fin, e2_mul(e2_mul(w[t.i1], u[t.i0]), e2_mul(e2_mul(e2_mul_f(eqc[rep * G + t.gate], t.c), eqx[rep * S + t.j0]), eqy[rep * S + t.j1])));
        if (!e2_eq(r2.first, fin)) throw Reject("vanilla node: phase-2 final evaluation mismatch");
        return sub;
    }

    // F(r, x) = scale * prod_b (1 + r_b (w^(2^b x) - 1)), built in O(N): factor b depends on x mod 2^(L-b)
    static std::vector<E2> fft_row(const std::vector<E2>& r, int L, bool inverse) {
        const size_t N = (size_t)1 << L;
        u64 w = root_of_unity(L);
        if (inverse) w = gl_inv(w);
        std::vector<u64> W(N);
        W[0] = 1;
        for (size_t i = 1; i < N; i++) W[i] = gl_mul(W[i - 1], w);
        std::vector<E2> cur(1, inverse ? e2(gl_inv(gl_from_u64(N)), 0) : e2_one());
        for (int b = L - 1; b >= 0; b--) {
            const size_t sz = (size_t)1 << (L - b);
            std::vector<E2> nxt(sz);
            for (size_t x = 0; x < sz; x++) {
                E2 f = e2_add_f(e2_mul_f(r[b], gl_sub(W[(x << b) & (N - 1)], 1)), 1);
                nxt[x] = e2_mul(cur[x & (sz / 2 - 1)], f);
            }
            cur.swap(nxt);
        }
        return cur;
    }

    std::vector<std::vector<Claim>> fft(const HNode& n, const std::vector<Claim>& cl, const std::vector<E2>& alpha) {
        E2 claim = e2_zero();
        for (size_t a = 0; a < cl.size(); a++) claim = e2_add(claim, e2_mul(cl[a].value, alpha[a]));
        auto r = sumcheck(2, n.log2_size, claim);
        E2 u = rd.read_e();
        std::vector<E2> eqx = eq_table(r.second);
        E2 fr = e2_zero();
        for (size_t a = 0; a < cl.size(); a++) {
            std::vector<E2> F = fft_row(cl[a].point, n.log2_size, n.inverse);
            E2 s = e2_zero();
            for (size_t x = 0; x < F.size(); x++) s = e2_add(s, e2_mul(F[x], eqx[x]));
            fr = e2_add(fr, e2_mul(s, alpha[a]));
        }
        if (!e2_eq(r.first, e2_mul(u, fr))) throw Reject("fft node: final evaluation mismatch");
        return {{Claim{r.second, u}}};
    }
};

}  // namespace

// returns "" on accept, the rejection reason otherwise
std::string verify_proof(const Params& p, const LassoPlan& lp, const HCircuit& c, const Witness& w, const uint8_t* proof, size_t len) {
    try {
        Verifier V{ProofReader{proof, len}, ChallengeSource{}};
        std::vector<E2> point = V.ch.squeeze_n(p.ct0is_log2());     // sk_encryption_circuit.rs:482
        E2 value = mle_eval(w.ct0is.data(), point);                 // :495
        std::vector<std::vector<Claim>> claims(c.nodes.size());
        claims[c.lasso_id].push_back(Claim{{}, e2_zero()});         // :500
        claims[c.sum_id].push_back(Claim{point, value});
        for (size_t q = c.topo.size(); q-- > 0;) {                  // verify_gkr :509-510
            int id = c.topo[q];
            const HNode& n = c.nodes[id];
            if (n.kind == NK_INPUT) continue;
            const std::vector<Claim>& cl = claims[id];
            if (cl.empty()) throw Reject("node without claim");
            std::vector<E2> alpha = cl.size() > 1 ? V.ch.squeeze_n(cl.size()) : std::vector<E2>{e2_one()};
            std::vector<std::vector<Claim>> sub;
            if (n.kind == NK_VANILLA) sub = V.vanilla(n, cl, alpha);
            else if (n.kind == NK_FFT) sub = V.fft(n, cl, alpha);
            else sub = {{V.lasso(lp)}};
            for (size_t i = 0; i < n.preds.size(); i++) for (auto& s : sub[i]) claims[n.preds[i]].push_back(s);
        }
        // izip_eq!(inputs, input_claims): input.evaluate(point) == value (:512-516)
        const size_t SZ = p.SZ();
        std::vector<const u64*> tabs = {w.s.data(), w.e.data(), w.k1.data()};
        for (int i = 0; i < p.k; i++) tabs.push_back(&w.ais[i * SZ]);
        for (int i = 0; i < p.k; i++) tabs.push_back(&w.r1is[i * SZ]);
        tabs.push_back(w.r2is.data());
        for (size_t k = 0; k < c.input_ids.size(); k++)
            for (auto& cl : claims[c.input_ids[k]])
                if (!e2_eq(mle_eval(tabs[k], cl.point), cl.value)) throw Reject("input claim mismatch at input " + std::to_string(k));
        return "";
    } catch (const Reject& r) {
        return r.what();
    }
}

}  // namespace hg
