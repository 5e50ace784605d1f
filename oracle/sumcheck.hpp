// ORACLE (test infrastructure). Restatement of `gkr::sum_check::{prove_sum_check,
// verify_sum_check, Generic}` from the third-party crate `gkr` (github.com/nulltea/gkr-lasso,
// un-pinned: /root/reference/Cargo.toml:10,63-64; source NOT under /root/reference).
// Call sites that fix the interface: lasso.rs:278-279 (prove, returns (claim, point, evals)),
// lasso.rs:130, prover.rs:242-252, verifier.rs:218-221 (verify returns (claim, point)).
//
// PARITY UNPINNED — conventions restated from the published sum-check protocol and isolated here:
//   C1  round message = the d+1 coefficients (low -> high) of the round polynomial, each an E;
//       the polynomial interpolates evaluations at t = 0..d where eval(1) := claim - eval(0)
//       (derived, never summed). The in-tree callers REQUIRE such a derivation: both in-tree
//       sum-check functions multiply by `Expression::poly(0)` instead of an eq table
//       (lasso.rs:472, prover.rs:276), so the running claim is not the true hypercube sum, yet
//       the reference's prove->verify round trips pass (test.rs:37-44).
//   C2  Expression::distribute_powers(exprs, b) = sum_i b^i * exprs[i]  (lowest power first; the
//       only order consistent with in-tree combine_lookups range.rs:184-195 and
//       sum_check_claim prover.rs:281-286).
//   C3  fix_var binds the LOWEST remaining variable each round: T'[j] = T[2j] + r (T[2j+1]-T[2j]).
//   C4  prove_sum_check does not write final evaluations (callers do: prover.rs:257, or drop
//       them: lasso.rs:278-288).
//   verifier: checks 2*c0 + c1 + ... + cd == claim each round, then claim <- p(r).
//
// Three function shapes occur on the path:
//   COLLATION  g = p_0 * sum_{i<P} M^i p_i                 (deg 2)  lasso.rs:457-475, range.rs:197-204
//   GRANDPROD  g = p_0 * sum_{i<P/2} gam^i p_{2i} p_{2i+1} (deg 3)  prover.rs:268-279
//   PRODSUM    g = sum_{i<P/2} p_{2i} p_{2i+1}             (deg 2)  Libra / zkCNN node reductions
#pragma once
#include <vector>
#include <stdexcept>
#include "gl.hpp"
#include "poly.hpp"
#include "transcript.hpp"

namespace orc {

enum ScKind { SC_COLLATION = 0, SC_GRANDPROD = 1, SC_PRODSUM = 2 };

struct ScTable {
    const uint64_t* fp = nullptr;  // base-field view (round 0), not owned
    std::vector<E> e;              // extension values (after first fold, or from the start)
    bool base = false;
    size_t n = 0;                  // current length
    static ScTable from_f(const uint64_t* p, size_t n) { ScTable t; t.fp = p; t.base = true; t.n = n; return t; }
    static ScTable from_e(std::vector<E> v) { ScTable t; t.n = v.size(); t.e = std::move(v); return t; }
    inline E at(size_t i) const { return base ? E{fp[i], 0} : e[i]; }
};

struct ScFunc {
    ScKind kind;
    size_t num_vars;
    std::vector<E> pw;  // COLLATION: M^i (as E, c1 = 0); GRANDPROD: gamma^i
    int degree() const { return kind == SC_GRANDPROD ? 3 : 2; }
};

static inline std::vector<E> powers_e(E b, size_t n) {
    std::vector<E> p(n);
    E c = e_one();
    for (size_t i = 0; i < n; i++) { p[i] = c; c = e_mul(c, b); }
    return p;
}

// interpolate evaluations at t = 0..d (d = 2 or 3) to coefficients, low -> high
static inline std::vector<E> interpolate(const std::vector<E>& ev) {
    const uint64_t inv2 = f_inv(2), inv3 = f_inv(3), inv6 = f_inv(6);
    size_t d = ev.size() - 1;
    if (d == 2) {
        E d2 = e_add(e_sub(ev[2], e_dbl(ev[1])), ev[0]);
        E c2 = e_mul_f(d2, inv2);
        E c1 = e_sub(e_sub(ev[1], ev[0]), c2);
        return {ev[0], c1, c2};
    } else if (d == 3) {
        E d1 = e_sub(ev[1], ev[0]);
        E d2 = e_add(e_sub(ev[2], e_dbl(ev[1])), ev[0]);
        E t3e2 = e_add(e_dbl(ev[2]), ev[2]), t3e1 = e_add(e_dbl(ev[1]), ev[1]);
        E d3 = e_sub(e_add(e_sub(ev[3], t3e2), t3e1), ev[0]);
        E c3 = e_mul_f(d3, inv6);
        E c2 = e_sub(e_mul_f(d2, inv2), e_mul_f(d3, inv2));
        E c1 = e_add(e_sub(d1, e_mul_f(d2, inv2)), e_mul_f(d3, inv3));
        return {ev[0], c1, c2, c3};
    }
    throw std::runtime_error("interpolate: unsupported degree");
}

// true hypercube sums of g at t = 0, 2, .., d (index 1 left zero) for the current tables
static inline void sc_round_evals(const ScFunc& g, const std::vector<ScTable>& T, E out[4]) {
    const size_t P = T.size();
    const size_t half = T[0].n >> 1;
    const int d = g.degree();
    const bool all_base = [&] { for (auto& t : T) if (!t.base) return false; return true; }();
    E acc[4] = {e_zero(), e_zero(), e_zero(), e_zero()};
#pragma omp parallel
    {
        E a[4] = {e_zero(), e_zero(), e_zero(), e_zero()};
#pragma omp for nowait schedule(static)
        for (long long jj = 0; jj < (long long)half; jj++) {
            size_t j = (size_t)jj;
            if (g.kind == SC_COLLATION) {
                if (all_base) {
                    uint64_t in0 = 0, in2 = 0, p00 = 0, p02 = 0;
                    for (size_t i = 0; i < P; i++) {
                        uint64_t x = T[i].fp[2 * j], y = T[i].fp[2 * j + 1];
                        uint64_t v2 = f_sub(f_add(y, y), x);
                        if (i == 0) { p00 = x; p02 = v2; }
                        in0 = f_add(in0, f_mul(g.pw[i].c0, x));
                        in2 = f_add(in2, f_mul(g.pw[i].c0, v2));
                    }
                    a[0] = e_add_f(a[0], f_mul(p00, in0));
                    a[2] = e_add_f(a[2], f_mul(p02, in2));
                } else {
                    E in0 = e_zero(), in2 = e_zero(), p00 = e_zero(), p02 = e_zero();
                    for (size_t i = 0; i < P; i++) {
                        E x = T[i].at(2 * j), y = T[i].at(2 * j + 1);
                        E v2 = e_sub(e_dbl(y), x);
                        if (i == 0) { p00 = x; p02 = v2; }
                        in0 = e_add(in0, e_mul_f(x, g.pw[i].c0));
                        in2 = e_add(in2, e_mul_f(v2, g.pw[i].c0));
                    }
                    a[0] = e_add(a[0], e_mul(p00, in0));
                    a[2] = e_add(a[2], e_mul(p02, in2));
                }
            } else if (g.kind == SC_GRANDPROD) {
                if (all_base) {
                    E in[4] = {e_zero(), e_zero(), e_zero(), e_zero()};
                    uint64_t p0[4] = {0, 0, 0, 0};
                    for (size_t i = 0; i < P / 2; i++) {
                        uint64_t xl = T[2 * i].fp[2 * j], yl = T[2 * i].fp[2 * j + 1];
                        uint64_t xr = T[2 * i + 1].fp[2 * j], yr = T[2 * i + 1].fp[2 * j + 1];
                        uint64_t dl = f_sub(yl, xl), dr = f_sub(yr, xr);
                        uint64_t l2 = f_add(yl, dl), r2 = f_add(yr, dr);
                        uint64_t l3 = f_add(l2, dl), r3 = f_add(r2, dr);
                        if (i == 0) { p0[0] = xl; p0[2] = l2; p0[3] = l3; }
                        in[0] = e_add(in[0], e_mul_f(g.pw[i], f_mul(xl, xr)));
                        in[2] = e_add(in[2], e_mul_f(g.pw[i], f_mul(l2, r2)));
                        in[3] = e_add(in[3], e_mul_f(g.pw[i], f_mul(l3, r3)));
                    }
                    for (int t : {0, 2, 3}) a[t] = e_add(a[t], e_mul_f(in[t], p0[t]));
                } else {
                    E in[4] = {e_zero(), e_zero(), e_zero(), e_zero()};
                    E p0[4] = {e_zero(), e_zero(), e_zero(), e_zero()};
                    for (size_t i = 0; i < P / 2; i++) {
                        E xl = T[2 * i].at(2 * j), yl = T[2 * i].at(2 * j + 1);
                        E xr = T[2 * i + 1].at(2 * j), yr = T[2 * i + 1].at(2 * j + 1);
                        E dl = e_sub(yl, xl), dr = e_sub(yr, xr);
                        E l2 = e_add(yl, dl), r2 = e_add(yr, dr);
                        E l3 = e_add(l2, dl), r3 = e_add(r2, dr);
                        if (i == 0) { p0[0] = xl; p0[2] = l2; p0[3] = l3; }
                        in[0] = e_add(in[0], e_mul(g.pw[i], e_mul(xl, xr)));
                        in[2] = e_add(in[2], e_mul(g.pw[i], e_mul(l2, r2)));
                        in[3] = e_add(in[3], e_mul(g.pw[i], e_mul(l3, r3)));
                    }
                    for (int t : {0, 2, 3}) a[t] = e_add(a[t], e_mul(in[t], p0[t]));
                }
            } else {  // SC_PRODSUM
                for (size_t i = 0; i < P / 2; i++) {
                    const ScTable& A = T[2 * i];
                    const ScTable& B = T[2 * i + 1];
                    E xb = B.at(2 * j), yb = B.at(2 * j + 1);
                    E b2 = e_sub(e_dbl(yb), xb);
                    if (A.base) {
                        uint64_t xa = A.fp[2 * j], ya = A.fp[2 * j + 1];
                        uint64_t a2 = f_sub(f_add(ya, ya), xa);
                        a[0] = e_add(a[0], e_mul_f(xb, xa));
                        a[2] = e_add(a[2], e_mul_f(b2, a2));
                    } else {
                        E xa = A.e[2 * j], ya = A.e[2 * j + 1];
                        E a2 = e_sub(e_dbl(ya), xa);
                        a[0] = e_add(a[0], e_mul(xb, xa));
                        a[2] = e_add(a[2], e_mul(b2, a2));
                    }
                }
            }
        }
#pragma omp critical
        for (int t = 0; t <= d; t++) acc[t] = e_add(acc[t], a[t]);
    }
    for (int t = 0; t < 4; t++) out[t] = acc[t];
}

static inline void sc_fold(std::vector<ScTable>& T, E r) {
    for (auto& t : T) {
        size_t half = t.n >> 1;
        std::vector<E> nv(half);
        if (t.base) {
            const uint64_t* p = t.fp;
#pragma omp parallel for schedule(static)
            for (long long j = 0; j < (long long)half; j++) {
                uint64_t x = p[2 * j], y = p[2 * j + 1];
                nv[j] = e_add_f(e_mul_f(r, f_sub(y, x)), x);
            }
        } else {
#pragma omp parallel for schedule(static)
            for (long long j = 0; j < (long long)half; j++) {
                E x = t.e[2 * j], y = t.e[2 * j + 1];
                nv[j] = e_add(x, e_mul(r, e_sub(y, x)));
            }
        }
        t.e.swap(nv);
        t.base = false;
        t.fp = nullptr;
        t.n = half;
    }
}

struct ScResult {
    E claim;
    std::vector<E> point;
    std::vector<E> evals;
};

// `record` (optional) receives the raw true sums per round (t = 0,2[,3]) for kernel-level parity tests.
static inline ScResult prove_sum_check(const ScFunc& g, E claim, std::vector<ScTable> T, TranscriptW& tr,
                                       std::vector<E>* record = nullptr) {
    ScResult res;
    const int d = g.degree();
    for (size_t round = 0; round < g.num_vars; round++) {
        E s[4];
        sc_round_evals(g, T, s);
        if (record) { record->push_back(s[0]); record->push_back(s[2]); if (d == 3) record->push_back(s[3]); }
        std::vector<E> ev(d + 1);
        ev[0] = s[0];
        ev[1] = e_sub(claim, s[0]);  // C1: derived from the running claim
        for (int t = 2; t <= d; t++) ev[t] = s[t];
        std::vector<E> coeffs = interpolate(ev);
        tr.write_es(coeffs);
        E r = tr.squeeze();
        claim = horner(coeffs, r);
        sc_fold(T, r);
        res.point.push_back(r);
    }
    res.claim = claim;
    for (auto& t : T) res.evals.push_back(t.at(0));
    return res;
}

static inline std::pair<E, std::vector<E>> verify_sum_check(int degree, size_t num_vars, E claim, TranscriptR& tr) {
    std::vector<E> point;
    for (size_t round = 0; round < num_vars; round++) {
        std::vector<E> c = tr.read_es(degree + 1);
        E s = e_dbl(c[0]);
        for (int i = 1; i <= degree; i++) s = e_add(s, c[i]);
        if (!e_eq(s, claim)) throw std::runtime_error("InvalidSumCheck: round polynomial does not match claim");
        E r = tr.squeeze();
        claim = horner(c, r);
        point.push_back(r);
    }
    return {claim, point};
}

}  // namespace orc
