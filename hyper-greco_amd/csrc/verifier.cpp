// CPU verifier = BfvEncrypt::verify [REF bfv-gkr/src/sk_encryption_circuit.rs:462-517] with
// LassoNode::verify_claim_reduction [REF lasso/src/lasso.rs:116-139], MemoryCheckingVerifier
// [REF lasso/src/memory_checking/verifier.rs:61-95,130-235] and the sub-table MLE closed forms
// [REF lasso/src/table/range.rs:19-26,74-112]. Verification is a host-side job in the reference too;
// this file is the product's own implementation (independent of the test-only CPU restatement).
// The Vanilla / FFT node checks mirror the prover's conventions (the external `gkr`
// crate is unpinned).
// One template, two fields: GlField (F = Goldilocks, E = GoldilocksExt2: the `goldilocks` test family) and BnField
// (F = E = bn256::Fr: the `bn254` test family, sk_encryption_circuit.rs:614-626).
#include <cstring>
#include <functional>
#include <map>
#include <algorithm>
#include <omp.h>
#include "host.hpp"
#include "bn254_field.hpp"

namespace hg {

namespace {

struct Reject : std::runtime_error { using std::runtime_error::runtime_error; };

struct ProofBytes {
    const uint8_t* p; size_t len; size_t pos = 0;
    const uint8_t* take(size_t n) {
        if (pos + n > len) throw Reject("proof: unexpected end of stream");
        const uint8_t* q = p + pos;
        pos += n;
        return q;
    }
};

// ---- the two fields ---------------------------------------------------------------------------------------------
struct GlField {
    typedef E2 E;
    static E zero() { return e2_zero(); }
    static E one() { return e2_one(); }
    static E add(E a, E b) { return e2_add(a, b); }
    static E sub(E a, E b) { return e2_sub(a, b); }
    static E mul(E a, E b) { return e2_mul(a, b); }
    static bool eq(E a, E b) { return e2_eq(a, b); }
    static E from_u(u64 c) { return e2(gl_from_u64(c), 0); }       // a non-negative integer constant
    static E mul_u(E a, u64 c) { return e2_mul_f(a, gl_from_u64(c)); }
    static E mul_table(E a, u64 v) { return e2_mul_f(a, v); }       // a * (witness table entry, a canonical base element)
    static E base0(E c) { return e2(c.c0, 0); }                     // as_bases()[0] (verifier.rs:139-140)
    static E inv_u(u64 c) { return e2(gl_inv(gl_from_u64(c)), 0); }
    static E root(int L, bool inverse) { u64 w = root_of_unity(L); return e2(inverse ? gl_inv(w) : w, 0); }
    static E read(ProofBytes& b) {  // read_felt (transcript.rs:162-170): 8 bytes big-endian, canonical; two bases per element
        u64 c[2];
        for (int h = 0; h < 2; h++) {
            const uint8_t* q = b.take(8);
            u64 a = 0;
            for (int i = 0; i < 8; i++) a = (a << 8) | q[i];
            if (a >= GL_P) throw Reject("proof: invalid field element");
            c[h] = a;
        }
        return e2(c[0], c[1]);
    }
    struct Chal {  // hash state kept explicitly (host.hpp FsTranscript): the fixed chain, or - mode bit 0 - absorbing what is read
        FsTranscript t;
        E squeeze() { return t.squeeze(); }
        void absorb(E v) { t.absorb_read(v.c0); t.absorb_read(v.c1); }  // read_felt of an absorbing transcript (transcript.rs:210-222)
        void set_mode(int mode) { t.absorb = (mode & 1) != 0; }
    };
};

struct BnField {  // elements in Montgomery form
    typedef bn::Fr E;
    static E zero() { return bn::fr_zero(); }
    static E one() { return bn::fr_one_mont(); }
    static E add(E a, E b) { return bn::fr_add(a, b); }
    static E sub(E a, E b) { return bn::fr_sub(a, b); }
    static E mul(E a, E b) { return bn::fr_mul(a, b); }
    static bool eq(E a, E b) { return bn::fr_eq(a, b); }
    static E from_u(u64 c) { return bn::fr_to_mont(bn::fr_make(c, 0, 0, 0)); }
    static E mul_u(E a, u64 c) { return bn::fr_mul(a, from_u(c)); }
    // witness entries are small signed integers kept in the Goldilocks form (host.cpp: witness_from_json_bn254)
    static E mul_table(E a, u64 v) { return v < (1ULL << 63) ? bn::fr_mul(a, from_u(v)) : bn::fr_sub(zero(), bn::fr_mul(a, from_u(GL_P - v))); }
    static E base0(E c) { return c; }
    static E pow(E b, const u64 e[4]) {
        E r = one();
        for (int w = 3; w >= 0; w--)
            for (int bit = 63; bit >= 0; bit--) { r = mul(r, r); if ((e[w] >> bit) & 1) r = mul(r, b); }
        return r;
    }
    static E inv(E a) { const u64 e[4] = {bn::FR_P0 - 2, bn::FR_P1, bn::FR_P2, bn::FR_P3}; return pow(a, e); }
    static E inv_u(u64 c) { return inv(from_u(c)); }
    static E root(int L, bool inverse) {  // halo2curves ROOT_OF_UNITY = 7^((r-1)/2^28), squared down to order 2^L
        if (L > 28) throw Reject("bn254: two-adicity is 28");
        E w = bn::fr_to_mont(bn::fr_make(0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL));
        for (int i = L; i < 28; i++) w = mul(w, w);
        return inverse ? inv(w) : w;
    }
    static E read(ProofBytes& b) {  // 32 bytes big-endian, canonical (transcript.rs:162-170, 183-189)
        const uint8_t* q = b.take(32);
        bn::Fr v;
        for (int w = 0; w < 4; w++) {
            u64 a = 0;
            for (int i = 0; i < 8; i++) a = (a << 8) | q[8 * (3 - w) + i];
            v.l[w] = a;
        }
        if (bn::fr_geq_p(v)) throw Reject("proof: invalid field element");
        return bn::fr_to_mont(v);
    }
    struct Chal {  // c_j = LE(Keccak^j("")) mod r (transcript.rs:146-157, 198-203); one element per squeeze (E = F)
        uint8_t h[32];
        Chal() { keccak256(nullptr, 0, h); }
        void absorb(E) {}
        void set_mode(int mode) { if (mode != 0) throw Reject("bn254: protocol modes are implemented for the Goldilocks family only"); }
        E squeeze() {
            bn::Fr v;
            memcpy(v.l, h, 32);
            while (bn::fr_geq_p(v)) v = bn::fr_sub_p(v);
            uint8_t nx[32];
            keccak256(h, 32, nx);
            memcpy(h, nx, 32);
            return bn::fr_to_mont(v);
        }
    };
};

// The verifier's table-sized loops (eq tables, MLE evaluations of the public inputs, the wiring-predicate sums of the Vanilla nodes,
// DFT rows) run on the host's cores: the reference's verifier is rayon-parallel in the same places (README.md:44,56: 108 / 529 ms on
// 10 cores); single-threaded this one took seconds at n=32768 k=16. Field sums are exact, so the order of a reduction is free.
static inline int threads_for(size_t work, size_t grain) {
    const size_t mx = (size_t)hg_omp_threads(), want = work / grain;
    return (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(mx, 64), want));
}
template <class F, class Fn> typename F::E par_sum(size_t n, size_t grain, Fn fn) {
    typedef typename F::E E;
    E total = F::zero();
#pragma omp parallel num_threads(threads_for(n, grain))
    {
        E a = F::zero();
#pragma omp for nowait schedule(static)
        for (long long i = 0; i < (long long)n; i++) a = F::add(a, fn((size_t)i));
#pragma omp critical
        total = F::add(total, a);
    }
    return total;
}
// sum over (rep, t), rep < R, t < nt, of fn(rep, t) into `nacc` buckets chosen by the term (fn adds into acc[]): the wiring-predicate
// sums of the Vanilla nodes. Every thread takes a contiguous run of the R * nt terms and steps (rep, t) along it - no division per
// term - and the per-input (per input pair) factors u_i (w_i1 u_i0) are applied once per bucket at the end, not once per term.
template <class F, class Fn> std::vector<typename F::E> par_buckets(size_t R, size_t nt, size_t nacc, size_t grain, Fn fn) {
    typedef typename F::E E;
    std::vector<E> total(nacc, F::zero());
    const size_t n = R * nt;
    if (!n) return total;
#pragma omp parallel num_threads(threads_for(n, grain))
    {
        std::vector<E> acc(nacc, F::zero());
        const size_t T = (size_t)omp_get_num_threads(), me = (size_t)omp_get_thread_num();
        const size_t q0 = n * me / T, q1 = n * (me + 1) / T;
        size_t rep = q0 / nt, t = q0 % nt;
        for (size_t q = q0; q < q1; q++) {
            fn(rep, t, acc.data());
            if (++t == nt) { t = 0; rep++; }
        }
#pragma omp critical
        for (size_t i = 0; i < nacc; i++) total[i] = F::add(total[i], acc[i]);
    }
    return total;
}
template <class F> std::vector<typename F::E> eq_table(const std::vector<typename F::E>& r) {
    typedef typename F::E E;
    std::vector<E> t((size_t)1 << r.size());
    t[0] = F::one();
    size_t s = 1;
    for (size_t i = 0; i < r.size(); i++) {
        const E ri = r[i];
#pragma omp parallel for schedule(static) num_threads(threads_for(s, 4096))
        for (long long j = 0; j < (long long)s; j++) { E hi = F::mul(t[j], ri); t[j + s] = hi; t[j] = F::sub(t[j], hi); }
        s <<= 1;
    }
    return t;
}
template <class F> typename F::E mle_eval(const u64* tab, const std::vector<typename F::E>& pt) {
    typedef typename F::E E;
    std::vector<E> eq = eq_table<F>(pt);
    return par_sum<F>(eq.size(), 4096, [&](size_t j) -> E { return tab[j] ? F::mul_table(eq[j], tab[j]) : F::zero(); });
}
template <class F> typename F::E horner(const std::vector<typename F::E>& c, typename F::E x) {
    typename F::E r = F::zero();
    for (size_t i = c.size(); i-- > 0;) r = F::add(F::mul(r, x), c[i]);
    return r;
}

template <class F> struct Verifier {
    typedef typename F::E E;
    static constexpr size_t NOPOS = (size_t)-1;
    struct Claim { std::vector<E> point; E value; size_t off = NOPOS; };   // off: the point as a run of the challenge chain (E units)
    ProofBytes bytes;
    struct CountingChal {   // the field's challenge source plus the number of E challenges squeezed so far
        typename F::Chal c;
        size_t n = 0;
        E squeeze() { n++; return c.squeeze(); }
        void absorb(E v) { c.absorb(v); }
        void set_mode(int mode) { c.set_mode(mode); }
    } ch;
    bool ext_memcheck = false;  // mode bit 1: gamma, tau stay in E (prover.rs:36-39 truncates them; README.md:108)
    // the table-sized checks elsewhere (Goldilocks, mode 0): see VerifyBackend in host.hpp
    VerifyBackend* dev = nullptr;
    std::vector<std::function<void()>> deferred;   // checks that need a ticket: run after dev->finish()
    std::function<E()> late_claim = nullptr;       // a late-bound term of the NEXT node's initial claim (the output evaluation)
    E read_e() { E v = F::read(bytes); ch.absorb(v); return v; }
    std::vector<E> read_es(size_t n) { std::vector<E> v(n); for (auto& x : v) x = read_e(); return v; }
    std::vector<E> squeeze_n(size_t n) { std::vector<E> v(n); for (auto& x : v) x = ch.squeeze(); return v; }

    // verify_sum_check: d+1 coefficients per round, 2 c0 + c1 + .. + cd == claim, claim <- p(r)
    // `late` (optional): a term only known after dev->finish() that must be subtracted from the INITIAL claim - the first round's
    // check is then deferred; every later claim is p(r) of a round polynomial read from the proof
    std::pair<E, std::vector<E>> sumcheck(int deg, int nvars, E claim, std::function<E()> late = nullptr, size_t* point_off = nullptr) {
        std::vector<E> point;
        if (point_off) *point_off = ch.n;
        // a late term is folded into round 0's check: a sum-check without rounds has no place for it (no Vanilla node of the BFV
        // circuit has such a shape; refuse it instead of silently comparing the raw claim)
        if (late && nvars == 0) throw Error("verifier: a deferred claim term on a sum-check without rounds is not supported");
        for (int i = 0; i < nvars; i++) {
            std::vector<E> c = read_es(deg + 1);
            E s = F::add(c[0], c[0]);
            for (int k = 1; k <= deg; k++) s = F::add(s, c[k]);
            if (i == 0 && late) {
                const E known = claim;
                deferred.push_back([s, known, late] { if (!F::eq(s, F::sub(known, late()))) throw Reject("InvalidSumCheck: round polynomial does not match the running claim"); });
            } else
            if (!F::eq(s, claim)) throw Reject("InvalidSumCheck: round polynomial does not match the running claim");
            E r = ch.squeeze();
            claim = horner<F>(c, r);
            point.push_back(r);
        }
        return {claim, point};
    }

    // verify_grand_product (verifier.rs:178-235)
    std::pair<std::vector<E>, std::vector<E>> grand_product(int num_vars, int nb) {
        std::vector<E> claims = read_es(nb);
        std::vector<E> x;
        for (int n = 0; n < num_vars; n++) {
            std::vector<E> evals;
            if (n == 0) {
                evals = read_es(2 * (size_t)nb);
                for (int b = 0; b < nb; b++)
                    if (!F::eq(claims[b], F::mul(evals[2 * b], evals[2 * b + 1]))) throw Reject("InvalidSumCheck: unmatched sum check output");
                x.clear();
            } else {
                E gamma = ch.squeeze();
                E claim = F::zero(), g = F::one();
                for (int b = 0; b < nb; b++) { claim = F::add(claim, F::mul(claims[b], g)); g = F::mul(g, gamma); }
                auto r = sumcheck(3, n, claim);
                x = r.second;
                evals = read_es(2 * (size_t)nb);
            }
            E mu = ch.squeeze();
            for (int b = 0; b < nb; b++) claims[b] = F::add(evals[2 * b], F::mul(mu, F::sub(evals[2 * b + 1], evals[2 * b])));
            x.push_back(mu);
        }
        return {claims, x};
    }

    // sub-table MLE closed forms (range.rs:19-26, 74-112)
    static E subtable_mle(u64 bound, const std::vector<E>& y) {
        E result = F::zero();
        const size_t b = y.size();
        if (bound == 0) {
            for (size_t i = 0; i < b; i++) result = F::add(result, F::mul_u(y[i], 1ULL << i));
            return result;
        }
        int bits = 63 - __builtin_clzll(bound);
        u64 cutoff = (1ULL << (bits % LassoPlan::LOGM)) + bound % (1ULL << LassoPlan::LOGM);
        int cl2 = 63 - __builtin_clzll(cutoff);
        u64 g_base = 1ULL << cl2, extra = cutoff - g_base;
        for (size_t i = 0; i < b; i++) {
            if ((int)i < cl2) { result = F::add(result, F::mul_u(y[i], 1ULL << i)); continue; }
            E g_value = F::zero();
            if ((int)i == cl2 && extra) {
                // sum_{k < extra} (g_base + k) eq(y[0..cl2), k) (range.rs:93-105 walks the extra terms one by one: up to 2^16 of them,
                // cl2 products each - 10 ms of the verification at n=32768) in closed form: [0, extra) is a union of dyadic blocks, one per
                // set bit i of `extra` (the bits above i as in `extra`, bit i clear, the bits below free), and over a block
                //   sum eq = P (1 - y_i),   sum k eq = P (1 - y_i) (V + sum_{j < i} 2^j y_j)
                // with P the eq factor and V the value of the fixed high bits. The same field element: every step is an exact identity.
                std::vector<E> low(cl2 + 1, F::zero());   // low[i] = sum_{j < i} 2^j y_j
                for (int j = 0; j < cl2; j++) low[j + 1] = F::add(low[j], F::mul_u(y[j], 1ULL << j));
                E P = F::one(), S0 = F::zero(), S1 = F::zero();
                u64 V = 0;
                for (int q = cl2 - 1; q >= 0; q--) {
                    const E one_minus = F::sub(F::one(), y[q]);
                    if ((extra >> q) & 1) {
                        const E wgt = F::mul(P, one_minus);
                        S0 = F::add(S0, wgt);
                        S1 = F::add(S1, F::mul(wgt, F::add(F::from_u(V), low[q])));
                        P = F::mul(P, y[q]);
                        V += 1ULL << q;
                    } else P = F::mul(P, one_minus);
                }
                g_value = F::add(F::mul(F::from_u(g_base), S0), S1);
            }
            result = F::add(F::mul(F::sub(F::one(), y[i]), result), F::mul(y[i], g_value));
        }
        return result;
    }

    Claim lasso(const LassoPlan& lp) {  // lasso.rs:116-139
        const size_t r_off = ch.n;
        std::vector<E> r = squeeze_n(lp.nu);
        E claimed = read_e();
        sumcheck(2, lp.nu, claimed);  // collation: final evaluation not checked by the reference either (lasso.rs:129-130)
        const E gamma_e = ch.squeeze(), tau_e = ch.squeeze();
        const E gamma = ext_memcheck ? gamma_e : F::base0(gamma_e), tau = ext_memcheck ? tau_e : F::base0(tau_e), gamma2 = F::mul(gamma, gamma);  // verifier.rs:139-140
        auto hash = [&](E a, E v, E t) { return F::sub(F::add(F::add(a, F::mul(v, gamma)), F::mul(t, gamma2)), tau); };
        const int A = lp.alpha;
        auto rw = grand_product(lp.nu, 2 * A);
        auto ifr = grand_product(LassoPlan::LOGM, 2 * A);
        const std::vector<E>& y = ifr.second;
        E id_y = F::zero();
        for (size_t i = 0; i < y.size(); i++) id_y = F::add(id_y, F::mul_u(y[i], 1ULL << i));
        int off = 0;
        for (auto& chk : lp.chunks) {  // verify_memories (verifier.rs:61-95)
            const int nm = (int)chk.second.size();
            E dim_x = read_e(), rts_x = read_e(), fct_y = read_e();
            std::vector<E> e_xs = read_es(nm);
            for (int i = 0; i < nm; i++) {
                int m = chk.second[i];
                if (!F::eq(rw.first[off + i], hash(dim_x, e_xs[i], rts_x))) throw Reject("memory check: read hash mismatch");
                if (!F::eq(rw.first[A + off + i], hash(dim_x, e_xs[i], F::add(rts_x, F::one())))) throw Reject("memory check: write hash mismatch");
                E st = subtable_mle(lp.subtable_bound[lp.mems[m].subtable], y);
                if (!F::eq(ifr.first[off + i], hash(id_y, st, F::zero()))) throw Reject("memory check: init hash mismatch");
                if (!F::eq(ifr.first[A + off + i], hash(id_y, st, fct_y))) throw Reject("memory check: final hash mismatch");
            }
            off += nm;
        }
        return Claim{r, claimed, r_off};
    }

    static std::vector<E> combined_eq(const std::vector<Claim>& cl, const std::vector<E>& alpha) {
        std::vector<E> eqc = eq_table<F>(cl[0].point);
        if (cl.size() == 1) return eqc;
        const long long ne = (long long)eqc.size();
        const E a0 = alpha[0];
#pragma omp parallel for schedule(static) num_threads(threads_for((size_t)ne, 4096))
        for (long long i = 0; i < ne; i++) eqc[i] = F::mul(eqc[i], a0);
        for (size_t a = 1; a < cl.size(); a++) {
            std::vector<E> t = eq_table<F>(cl[a].point);
            const E aa = alpha[a];
#pragma omp parallel for schedule(static) num_threads(threads_for((size_t)ne, 4096))
            for (long long i = 0; i < ne; i++) eqc[i] = F::add(eqc[i], F::mul(t[i], aa));
        }
        return eqc;
    }

    static VerifyBackend::ClaimOffs claim_offs(const std::vector<Claim>& cl, size_t alpha_off) {
        VerifyBackend::ClaimOffs o;
        for (auto& c : cl) { if (c.off == NOPOS) throw Reject("verifier: a claim point is not a run of the challenge chain"); o.point_off.push_back(c.off); }
        o.unit = cl.size() == 1; o.alpha_off = alpha_off;
        return o;
    }
    // the same node checks with the table sums taken from the backend (tickets), the comparisons deferred
    std::vector<std::vector<Claim>> vanilla_dev(int id, const HNode& n, const std::vector<Claim>& cl, const std::vector<E>& alpha, size_t alpha_off) {
        const int nin = n.log2_sub_in + n.log2_reps;
        VerifyBackend* D = dev;
        D->begin_node(id, claim_offs(cl, alpha_off));
        E claim = F::zero();
        for (size_t a = 0; a < cl.size(); a++) claim = F::add(claim, F::mul(cl[a].value, alpha[a]));
        std::function<E()> late1 = nullptr;
        if (!n.w0.empty()) { const int t = D->const_sum(); late1 = [D, t] { return D->value(t); }; }
        if (late_claim) {   // (subtracted from the claim like the constant sum)
            const std::function<E()> a = late1, b = late_claim;
            late1 = [a, b] { return a ? F::add(a(), b()) : b(); };
        }
        size_t x_off = 0;
        auto r1 = sumcheck(2, nin, claim, late1, &x_off);
        std::vector<E> u(n.arity, F::zero());
        std::vector<std::vector<Claim>> sub(n.arity);
        for (int i = 0; i < n.arity; i++) if (n.left_use[i]) { u[i] = read_e(); sub[i].push_back(Claim{r1.second, u[i], x_off}); }
        D->set_x(x_off);
        std::function<E()> lin = [] { return F::zero(); };
        if (!n.lin.empty()) {
            const std::vector<int> tk = D->lin_terms();
            lin = [D, tk, u] { E s = F::zero(); for (size_t i = 0; i < tk.size(); i++) if (tk[i] >= 0) s = F::add(s, F::mul(u[i], D->value(tk[i]))); return s; };
        }
        if (n.mul.empty()) {
            const E fin1 = r1.first;
            deferred.push_back([fin1, lin] { if (!F::eq(fin1, lin())) throw Reject("vanilla node: final evaluation mismatch"); });
            D->end_node();
            return sub;
        }
        size_t y_off = 0;
        auto r2 = sumcheck(2, nin, r1.first, n.lin.empty() ? std::function<E()>(nullptr) : lin, &y_off);
        std::vector<E> w(n.arity, F::zero());
        for (int i = 0; i < n.arity; i++) if (n.right_use[i]) { w[i] = read_e(); sub[i].push_back(Claim{r2.second, w[i], y_off}); }
        std::vector<E2> u2(u.begin(), u.end());
        D->set_y(y_off, u2);
        const std::vector<int> tk = D->mul_terms();
        const E fin2 = r2.first;
        deferred.push_back([D, tk, w, fin2] {
            E s = F::zero();
            for (size_t i = 0; i < tk.size(); i++) if (tk[i] >= 0) s = F::add(s, F::mul(w[i], D->value(tk[i])));
            if (!F::eq(fin2, s)) throw Reject("vanilla node: phase-2 final evaluation mismatch");
        });
        D->end_node();
        return sub;
    }
    std::vector<std::vector<Claim>> fft_dev(int id, const HNode& n, const std::vector<Claim>& cl, const std::vector<E>& alpha, size_t alpha_off) {
        VerifyBackend* D = dev;
        D->begin_node(id, claim_offs(cl, alpha_off));
        E claim = F::zero();
        for (size_t a = 0; a < cl.size(); a++) claim = F::add(claim, F::mul(cl[a].value, alpha[a]));
        size_t x_off = 0;
        auto r = sumcheck(2, n.log2_size, claim, nullptr, &x_off);
        const E u = read_e();
        D->set_x(x_off);
        const int t = D->fft_term();
        const E fin = r.first;
        deferred.push_back([D, t, u, fin] { if (!F::eq(fin, F::mul(u, D->value(t)))) throw Reject("fft node: final evaluation mismatch"); });
        D->end_node();
        return {{Claim{r.second, u, x_off}}};
    }

    std::vector<std::vector<Claim>> vanilla(const HNode& n, const std::vector<Claim>& cl, const std::vector<E>& alpha) {
        const size_t G = (size_t)1 << n.log2_sub_out, S = (size_t)1 << n.log2_sub_in, R = (size_t)1 << n.log2_reps;
        const int nin = n.log2_sub_in + n.log2_reps;
        std::vector<E> eqc = combined_eq(cl, alpha);
        E claim = F::zero();
        for (size_t a = 0; a < cl.size(); a++) claim = F::add(claim, F::mul(cl[a].value, alpha[a]));
        if (!n.w0.empty()) {
            const size_t nw = n.w0.size();
            claim = F::sub(claim, par_buckets<F>(R, nw, 1, 4096, [&](size_t rep, size_t ti, E* acc) { const auto& t = n.w0[ti]; acc[0] = F::add(acc[0], F::mul_u(eqc[rep * G + t.gate], t.c)); })[0]);
        }
        auto r1 = sumcheck(2, nin, claim);
        std::vector<E> u(n.arity, F::zero());
        std::vector<std::vector<Claim>> sub(n.arity);
        for (int i = 0; i < n.arity; i++) if (n.left_use[i]) { u[i] = read_e(); sub[i].push_back(Claim{r1.second, u[i]}); }
        std::vector<E> eqx = eq_table<F>(r1.second);
        E lin = F::zero();
        if (!n.lin.empty()) {
            const std::vector<E> per_in = par_buckets<F>(R, n.lin.size(), (size_t)n.arity, 2048, [&](size_t rep, size_t ti, E* acc) {
                const auto& t = n.lin[ti];
                acc[t.in] = F::add(acc[t.in], F::mul(F::mul_u(eqc[rep * G + t.gate], t.c), eqx[rep * S + t.j]));
            });
            for (int i = 0; i < n.arity; i++) lin = F::add(lin, F::mul(u[i], per_in[i]));
        }
        if (n.mul.empty()) {
            if (!F::eq(r1.first, lin)) throw Reject("vanilla node: final evaluation mismatch");
            return sub;
        }
        auto r2 = sumcheck(2, nin, F::sub(r1.first, lin));
        std::vector<E> w(n.arity, F::zero());
        for (int i = 0; i < n.arity; i++) if (n.right_use[i]) { w[i] = read_e(); sub[i].push_back(Claim{r2.second, w[i]}); }
        std::vector<E> eqy = eq_table<F>(r2.second);
        const size_t ar = (size_t)n.arity;
        const std::vector<E> per_pair = par_buckets<F>(R, n.mul.size(), ar * ar, 2048, [&](size_t rep, size_t ti, E* acc) {
            const auto& t = n.mul[ti];
            E& a = acc[(size_t)t.i1 * ar + t.i0];
            a = F::add(a, F::mul(F::mul(F::mul_u(eqc[rep * G + t.gate], t.c), eqx[rep * S + t.j0]), eqy[rep * S + t.j1]));
        });
        E fin = F::zero();
        for (size_t i1 = 0; i1 < ar; i1++)
            for (size_t i0 = 0; i0 < ar; i0++) fin = F::add(fin, F::mul(F::mul(w[i1], u[i0]), per_pair[i1 * ar + i0]));
        if (!F::eq(r2.first, fin)) throw Reject("vanilla node: phase-2 final evaluation mismatch");
        return sub;
    }

    // F(r, x) = scale * prod_b (1 + r_b (w^(2^b x) - 1)), built in O(N): factor b depends on x mod 2^(L-b)
    std::map<std::pair<int, bool>, std::vector<E>> w_tables;   // powers of the root per (size, direction): shared by every FFT node of a proof
    const std::vector<E>& fft_powers(int L, bool inverse) {
        const size_t N = (size_t)1 << L;
        const E w = F::root(L, inverse);
        std::vector<E>& W = w_tables[{L, inverse}];
        if (W.empty()) {   // powers of w in runs of 4096, each run started from a stride power
            W.resize(N);
            const size_t RUN = 4096, nrun = (N + RUN - 1) / RUN;
            E wrun = F::one();
            for (size_t i = 0; i < std::min(RUN, N); i++) wrun = F::mul(wrun, w);
            std::vector<E> start(nrun);
            start[0] = F::one();
            for (size_t q = 1; q < nrun; q++) start[q] = F::mul(start[q - 1], wrun);
#pragma omp parallel for schedule(static) num_threads(threads_for(N, 8192))
            for (long long q = 0; q < (long long)nrun; q++) {
                const size_t i0 = (size_t)q * RUN, i1 = std::min(N, i0 + RUN);
                W[i0] = start[q];
                for (size_t i = i0 + 1; i < i1; i++) W[i] = F::mul(W[i - 1], w);
            }
        }
        return W;
    }
    std::vector<E> fft_row(const std::vector<E>& r, int L, bool inverse) {   // (reads the table of powers: fill it before going parallel)
        const size_t N = (size_t)1 << L;
        const std::vector<E>& W = w_tables.at({L, inverse});
        std::vector<E> cur(1, inverse ? F::inv_u((u64)N) : F::one());
        for (int b = L - 1; b >= 0; b--) {
            const size_t sz = (size_t)1 << (L - b);
            std::vector<E> nxt(sz);
            const E rb = r[b];
#pragma omp parallel for schedule(static) num_threads(threads_for(sz, 4096))
            for (long long xx = 0; xx < (long long)sz; xx++) {
                const size_t x = (size_t)xx;
                E f = F::add(F::mul(rb, F::sub(W[(x << b) & (N - 1)], F::one())), F::one());
                nxt[x] = F::mul(cur[x & (sz / 2 - 1)], f);
            }
            cur.swap(nxt);
        }
        return cur;
    }

    std::vector<std::vector<Claim>> fft(const HNode& n, const std::vector<Claim>& cl, const std::vector<E>& alpha) {
        E claim = F::zero();
        for (size_t a = 0; a < cl.size(); a++) claim = F::add(claim, F::mul(cl[a].value, alpha[a]));
        auto r = sumcheck(2, n.log2_size, claim);
        E u = read_e();
        // The final check (an eq table, one DFT row per claim and their dot products: ~4 N products) does not feed the walk: it is
        // queued and the 2k+1 FFT nodes' checks run side by side at the end (run_node_checks), each single-threaded - inside one node
        // the loops are too short for the cores (16 threads on 2^16 entries reached a fifth of their sum).
        const int L = n.log2_size;
        const bool inverse = n.inverse;
        (void)fft_powers(L, inverse);   // (the shared table of powers is filled here, on the walk's thread)
        const std::vector<E> pt = r.second;
        const E fin = r.first;
        node_checks.push_back([this, cl, alpha, pt, fin, u, L, inverse]() -> std::string {
            std::vector<E> eqx = eq_table<F>(pt);
            E fr = F::zero();
            for (size_t a = 0; a < cl.size(); a++) {
                std::vector<E> row = fft_row(cl[a].point, L, inverse);
                E s = F::zero();
                for (size_t x = 0; x < row.size(); x++) s = F::add(s, F::mul(row[x], eqx[x]));
                fr = F::add(fr, F::mul(s, alpha[a]));
            }
            return F::eq(fin, F::mul(u, fr)) ? std::string() : std::string("fft node: final evaluation mismatch");
        });
        return {{Claim{r.second, u}}};
    }
    // queued node checks, dealt to the cores; "" or the first failure in walk order
    std::vector<std::function<std::string()>> node_checks;
    std::string run_node_checks() {
        std::vector<std::string> why(node_checks.size());
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads_for(node_checks.size() * 65536, 65536))
        for (long long q = 0; q < (long long)node_checks.size(); q++) {
            try { why[q] = node_checks[q](); } catch (const std::exception& e) { why[q] = e.what(); }
        }
        for (auto& s : why) if (!s.empty()) return s;
        return std::string();
    }
};


template <class F>
static std::string verify_impl(const Params& p, const LassoPlan& lp, const HCircuit& c, const Witness& w, const uint8_t* proof, size_t len, int mode) {
    typedef typename F::E E;
    typedef typename Verifier<F>::Claim Claim;
    try {
        Verifier<F> V;
        V.bytes = ProofBytes{proof, len};
        V.ch.set_mode(mode);
        V.ext_memcheck = (mode & 2) != 0;
        double t_kind[3] = {0, 0, 0};
        const double tv0 = omp_get_wtime();
        std::vector<E> point = V.squeeze_n(p.ct0is_log2());         // sk_encryption_circuit.rs:482
        E value = mle_eval<F>(w.ct0is.data(), point);               // :495
        std::vector<std::vector<Claim>> claims(c.nodes.size());
        claims[c.lasso_id].push_back(Claim{{}, F::zero()});         // :500
        claims[c.sum_id].push_back(Claim{point, value});
        for (size_t q = c.topo.size(); q-- > 0;) {                  // verify_gkr :509-510
            int id = c.topo[q];
            const HNode& n = c.nodes[id];
            if (n.kind == NK_INPUT) continue;
            const std::vector<Claim>& cl = claims[id];
            if (cl.empty()) throw Reject("node without claim");
            std::vector<E> alpha = cl.size() > 1 ? V.squeeze_n(cl.size()) : std::vector<E>{F::one()};
            std::vector<std::vector<Claim>> sub;
            const double tn = omp_get_wtime();
            if (n.kind == NK_VANILLA) sub = V.vanilla(n, cl, alpha);
            else if (n.kind == NK_FFT) sub = V.fft(n, cl, alpha);
            else sub = {{V.lasso(lp)}};
            t_kind[n.kind == NK_VANILLA ? 0 : n.kind == NK_FFT ? 1 : 2] += omp_get_wtime() - tn;
            for (size_t i = 0; i < n.preds.size(); i++) for (auto& s : sub[i]) claims[n.preds[i]].push_back(s);
        }
        {
            const double tq = omp_get_wtime();
            const std::string why = V.run_node_checks();
            t_kind[1] += omp_get_wtime() - tq;
            if (!why.empty()) throw Reject(why);
        }
        // (the reference does not check that the proof stream is fully consumed either)
        // izip_eq!(inputs, input_claims): input.evaluate(point) == value (:512-516)
        const size_t SZ = p.SZ();
        std::vector<const u64*> tabs = {w.s.data(), w.e.data(), w.k1.data()};
        for (int i = 0; i < p.k; i++) tabs.push_back(&w.ais[i * SZ]);
        for (int i = 0; i < p.k; i++) tabs.push_back(&w.r1is[i * SZ]);
        tabs.push_back(w.r2is.data());
        // (the 2k+4 inputs carry several claims each; every check is an eq table plus a dot product over 2^L entries: the checks are
        // independent, so they are dealt to the cores and each runs its own loops single-threaded inside the region)
        const double t_checks = omp_get_wtime();
        std::vector<std::pair<size_t, const Claim*>> checks;
        for (size_t k = 0; k < c.input_ids.size(); k++)
            for (auto& cl : claims[c.input_ids[k]]) checks.push_back({k, &cl});
        long long bad = -1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads_for(checks.size(), 1))
        for (long long q = 0; q < (long long)checks.size(); q++) {
            if (!F::eq(mle_eval<F>(tabs[checks[q].first], checks[q].second->point), checks[q].second->value)) {
#pragma omp critical
                if (bad < 0 || (long long)checks[q].first < bad) bad = (long long)checks[q].first;
            }
        }
        if (bad >= 0) throw Reject("input claim mismatch at input " + std::to_string(bad));
        if (hg_times("verify"))
            fprintf(stderr, "[hg] verify: %.1f ms (vanilla nodes %.1f, fft nodes %.1f, lasso node %.1f, %zu input claims %.1f)\n", (omp_get_wtime() - tv0) * 1e3,
                    t_kind[0] * 1e3, t_kind[1] * 1e3, t_kind[2] * 1e3, checks.size(), (omp_get_wtime() - t_checks) * 1e3);
        return "";
    } catch (const Reject& r) {
        return r.what();
    }
}

// The same walk with the table-sized sums taken from a backend (Goldilocks, mode 0): the host parses the proof, checks the round
// polynomials and the Lasso scalars, and hands out tickets; the comparisons that need them run after dev.finish().
static std::string verify_with_backend(VerifyBackend& dev, const Params& p, const LassoPlan& lp, const HCircuit& c, const uint8_t* proof, size_t len) {
    typedef GlField F;
    typedef F::E E;
    typedef Verifier<F>::Claim Claim;
    try {
        Verifier<F> V;
        V.bytes = ProofBytes{proof, len};
        V.ch.set_mode(0);
        V.dev = &dev;
        VerifyBackend* D = &dev;
        const size_t p_off = V.ch.n;
        std::vector<E> point = V.squeeze_n(p.ct0is_log2());         // sk_encryption_circuit.rs:482
        const int t_out = D->mle_ct0is(p_off, p.ct0is_log2());      // :495 - the value itself is only known after finish()
        // the output claim's VALUE enters the sum node's running claim: late-bound like every other ticket
        std::vector<std::vector<Claim>> claims(c.nodes.size());
        claims[c.lasso_id].push_back(Claim{{}, F::zero(), V.ch.n});  // :500
        claims[c.sum_id].push_back(Claim{point, F::zero(), p_off});
        for (size_t q = c.topo.size(); q-- > 0;) {                  // verify_gkr :509-510
            int id = c.topo[q];
            const HNode& n = c.nodes[id];
            if (n.kind == NK_INPUT) continue;
            std::vector<Claim>& cl = claims[id];
            if (cl.empty()) throw Reject("node without claim");
            const size_t alpha_off = V.ch.n;
            std::vector<E> alpha = cl.size() > 1 ? V.squeeze_n(cl.size()) : std::vector<E>{F::one()};
            std::vector<std::vector<Claim>> sub;
            if (id == c.sum_id) {
                // claim = sum_a alpha_a value_a with value_0 = the output evaluation (a ticket): fold it into the node's first deferred
                // check by running the node with value_0 = 0 and a late term -alpha_0 * value(t_out)
                if (n.kind != NK_VANILLA) throw Reject("verifier: the output node is expected to be a Vanilla node");
                const E a0 = alpha[0];
                V.late_claim = [D, t_out, a0] { return F::sub(F::zero(), F::mul(a0, D->value(t_out))); };
            }
            if (n.kind == NK_VANILLA) sub = V.vanilla_dev(id, n, cl, alpha, alpha_off);
            else if (n.kind == NK_FFT) sub = V.fft_dev(id, n, cl, alpha, alpha_off);
            else sub = {{V.lasso(lp)}};
            V.late_claim = nullptr;
            for (size_t i = 0; i < n.preds.size(); i++) for (auto& s : sub[i]) claims[n.preds[i]].push_back(s);
        }
        // izip_eq!(inputs, input_claims): input.evaluate(point) == value (:512-516)
        for (size_t k = 0; k < c.input_ids.size(); k++)
            for (auto& cl : claims[c.input_ids[k]]) {
                if (cl.off == Verifier<F>::NOPOS) throw Reject("verifier: an input claim point is not a run of the challenge chain");
                const int t = D->mle_input(k, cl.off, (int)cl.point.size());
                const E want = cl.value;
                V.deferred.push_back([D, t, want, k] { if (!F::eq(D->value(t), want)) throw Reject("input claim mismatch at input " + std::to_string(k)); });
            }
        dev.finish();
        for (auto& f : V.deferred) f();
        return "";
    } catch (const Reject& r) {
        return r.what();
    }
}

}  // namespace

std::string verify_proof_with(VerifyBackend& dev, const Params& p, const LassoPlan& lp, const HCircuit& c, const uint8_t* proof, size_t len) {
    return verify_with_backend(dev, p, lp, c, proof, len);
}

// return "" on accept, the rejection reason otherwise
std::string verify_proof(const Params& p, const LassoPlan& lp, const HCircuit& c, const Witness& w, const uint8_t* proof, size_t len, int mode) {
    return verify_impl<GlField>(p, lp, c, w, proof, len, mode);
}
std::string verify_proof_bn254(const Params& p, const LassoPlan& lp, const HCircuit& c, const Witness& w, const uint8_t* proof, size_t len) {
    return verify_impl<BnField>(p, lp, c, w, proof, len, 0);
}

}  // namespace hg
