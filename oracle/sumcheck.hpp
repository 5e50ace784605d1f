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
#include <memory>
#include <algorithm>
#include "field.hpp"
#include "poly.hpp"
#include "transcript.hpp"

namespace ORC_NS {

enum ScKind { SC_COLLATION = 0, SC_GRANDPROD = 1, SC_PRODSUM = 2 };

struct ScTable {
    const F* fp = nullptr;         // base-field view (round 0), not owned
    const E* ep = nullptr;         // extension view (after the first fold, or from the start)
    std::unique_ptr<E[]> own[2];   // ping-pong storage for folded tables (uninitialised on allocation)
    int cur = 0;
    bool base = false;
    size_t n = 0;                  // current length
    static ScTable from_f(const F* p, size_t n) { ScTable t; t.fp = p; t.base = true; t.n = n; return t; }
    static ScTable view_e(const E* p, size_t n) { ScTable t; t.ep = p; t.n = n; return t; }  // not owned (first fold allocates)
    static ScTable from_e(const std::vector<E>& v) {
        ScTable t; t.n = v.size();
        t.own[0].reset(new E[v.size() ? v.size() : 1]);
        for (size_t i = 0; i < v.size(); i++) t.own[0][i] = v[i];
        t.ep = t.own[0].get();
        return t;
    }
    inline E at(size_t i) const { return base ? e_from_f(fp[i]) : ep[i]; }
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
    static const F inv2 = f_inv(f_from_u64(2)), inv3 = f_inv(f_from_u64(3)), inv6 = f_inv(f_from_u64(6));
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

// true hypercube sums of g at t = 0, 2, .., d (index 1 left zero) for the current tables.
// Blocked over j so that each table is streamed in contiguous runs (the tables sit at power-of-two
// strides; touching all of them per j would thrash the cache sets).
static inline void sc_round_evals(const ScFunc& g, const std::vector<ScTable>& T, E out[4]) {
    const size_t P = T.size();
    const size_t half = T[0].n >> 1;
    const int d = g.degree();
    const bool all_base = [&] { for (auto& t : T) if (!t.base) return false; return true; }();
    const size_t BLK = 256;
    const size_t nblk = (half + BLK - 1) / BLK;
    E acc[4] = {e_zero(), e_zero(), e_zero(), e_zero()};
    // (threads in proportion to the round's work: a late round of a few hundred pairs run by 128 threads costs a fork / join, eight
    // allocations per thread and a critical section for nothing - that is what made the baseline SLOWER beyond 16 threads)
    const int nthr = (int)std::min<size_t>(nblk, (size_t)orc_threads_for(P * half, 2048));
#pragma omp parallel num_threads(nthr)
    {
        E a[4] = {e_zero(), e_zero(), e_zero(), e_zero()};
        std::vector<E> s0(BLK), s2(BLK), s3(BLK), q0(BLK), q2(BLK), q3(BLK);
        std::vector<F> b0(BLK), b2(BLK);
#pragma omp for nowait schedule(static)
        for (long long bb = 0; bb < (long long)nblk; bb++) {
            const size_t j0 = (size_t)bb * BLK, cnt = std::min(BLK, half - j0);
            for (size_t u = 0; u < cnt; u++) s0[u] = s2[u] = s3[u] = e_zero();
            if (g.kind == SC_COLLATION) {
                if (all_base) for (size_t u = 0; u < cnt; u++) b0[u] = b2[u] = f_zero();
                for (size_t i = 0; i < P; i++) {
                    const F m = e_limb0(g.pw[i]);  // M^i is a base-field constant (range.rs:197-204)
                    if (all_base) {
                        const F* p = T[i].fp + 2 * j0;
                        for (size_t u = 0; u < cnt; u++) {
                            F x = p[2 * u], y = p[2 * u + 1];
                            F v2 = f_sub(f_add(y, y), x);
                            if (i == 0) { q0[u] = e_from_f(x); q2[u] = e_from_f(v2); }
                            b0[u] = f_add(b0[u], f_mul(m, x));
                            b2[u] = f_add(b2[u], f_mul(m, v2));
                        }
                    } else {
                        for (size_t u = 0; u < cnt; u++) {
                            E x = T[i].at(2 * (j0 + u)), y = T[i].at(2 * (j0 + u) + 1);
                            E v2 = e_sub(e_dbl(y), x);
                            if (i == 0) { q0[u] = x; q2[u] = v2; }
                            s0[u] = e_add(s0[u], e_mul_f(x, m));
                            s2[u] = e_add(s2[u], e_mul_f(v2, m));
                        }
                    }
                }
                if (all_base) for (size_t u = 0; u < cnt; u++) { s0[u] = e_from_f(b0[u]); s2[u] = e_from_f(b2[u]); }
                for (size_t u = 0; u < cnt; u++) { a[0] = e_add(a[0], e_mul(q0[u], s0[u])); a[2] = e_add(a[2], e_mul(q2[u], s2[u])); }
            } else if (g.kind == SC_GRANDPROD) {
                for (size_t i = 0; i < P / 2; i++) {
                    const E gm = g.pw[i];
                    if (all_base) {
                        const F* pl = T[2 * i].fp + 2 * j0;
                        const F* pr = T[2 * i + 1].fp + 2 * j0;
                        for (size_t u = 0; u < cnt; u++) {
                            F xl = pl[2 * u], yl = pl[2 * u + 1], xr = pr[2 * u], yr = pr[2 * u + 1];
                            F dl = f_sub(yl, xl), dr = f_sub(yr, xr);
                            F l2 = f_add(yl, dl), r2 = f_add(yr, dr);
                            F l3 = f_add(l2, dl), r3 = f_add(r2, dr);
                            if (i == 0) { q0[u] = e_from_f(xl); q2[u] = e_from_f(l2); q3[u] = e_from_f(l3); }
                            s0[u] = e_add(s0[u], e_mul_f(gm, f_mul(xl, xr)));
                            s2[u] = e_add(s2[u], e_mul_f(gm, f_mul(l2, r2)));
                            s3[u] = e_add(s3[u], e_mul_f(gm, f_mul(l3, r3)));
                        }
                    } else {
                        for (size_t u = 0; u < cnt; u++) {
                            size_t j = j0 + u;
                            E xl = T[2 * i].at(2 * j), yl = T[2 * i].at(2 * j + 1);
                            E xr = T[2 * i + 1].at(2 * j), yr = T[2 * i + 1].at(2 * j + 1);
                            E dl = e_sub(yl, xl), dr = e_sub(yr, xr);
                            E l2 = e_add(yl, dl), r2 = e_add(yr, dr);
                            E l3 = e_add(l2, dl), r3 = e_add(r2, dr);
                            if (i == 0) { q0[u] = xl; q2[u] = l2; q3[u] = l3; }
                            s0[u] = e_add(s0[u], e_mul(gm, e_mul(xl, xr)));
                            s2[u] = e_add(s2[u], e_mul(gm, e_mul(l2, r2)));
                            s3[u] = e_add(s3[u], e_mul(gm, e_mul(l3, r3)));
                        }
                    }
                }
                for (size_t u = 0; u < cnt; u++) {
                    a[0] = e_add(a[0], e_mul(s0[u], q0[u]));
                    a[2] = e_add(a[2], e_mul(s2[u], q2[u]));
                    a[3] = e_add(a[3], e_mul(s3[u], q3[u]));
                }
            } else {  // SC_PRODSUM
                for (size_t i = 0; i < P / 2; i++) {
                    const ScTable& A = T[2 * i];
                    const ScTable& B = T[2 * i + 1];
                    for (size_t u = 0; u < cnt; u++) {
                        size_t j = j0 + u;
                        E xb = B.at(2 * j), yb = B.at(2 * j + 1);
                        E b2 = e_sub(e_dbl(yb), xb);
                        if (A.base) {
                            F xa = A.fp[2 * j], ya = A.fp[2 * j + 1];
                            F a2 = f_sub(f_add(ya, ya), xa);
                            a[0] = e_add(a[0], e_mul_f(xb, xa));
                            a[2] = e_add(a[2], e_mul_f(b2, a2));
                        } else {
                            E xa = A.ep[2 * j], ya = A.ep[2 * j + 1];
                            E a2 = e_sub(e_dbl(ya), xa);
                            a[0] = e_add(a[0], e_mul(xb, xa));
                            a[2] = e_add(a[2], e_mul(b2, a2));
                        }
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
    const size_t half = T[0].n >> 1;
    for (auto& t : T) {
        int dst = t.base || t.ep != t.own[t.cur].get() ? 0 : 1 - t.cur;
        if (!t.own[dst]) t.own[dst].reset(new E[half ? half : 1]);
    }
    const size_t BLK = 1024, nblk = (half + BLK - 1) / BLK;
#pragma omp parallel for schedule(static) collapse(2) num_threads(orc_threads_for(T.size() * half, 4096))
    for (long long ti = 0; ti < (long long)T.size(); ti++)
        for (long long bb = 0; bb < (long long)nblk; bb++) {
            ScTable& t = T[ti];
            int dst = t.base || t.ep != t.own[t.cur].get() ? 0 : 1 - t.cur;
            E* o = t.own[dst].get();
            size_t j0 = (size_t)bb * BLK, j1 = std::min(half, j0 + BLK);
            if (t.base) {
                const F* p = t.fp;
                for (size_t j = j0; j < j1; j++) { F x = p[2 * j], y = p[2 * j + 1]; o[j] = e_add_f(e_mul_f(r, f_sub(y, x)), x); }
            } else {
                const E* p = t.ep;
                for (size_t j = j0; j < j1; j++) { E x = p[2 * j], y = p[2 * j + 1]; o[j] = e_add(x, e_mul(r, e_sub(y, x))); }
            }
        }
    for (auto& t : T) {
        int dst = t.base || t.ep != t.own[t.cur].get() ? 0 : 1 - t.cur;
        t.cur = dst;
        t.ep = t.own[dst].get();
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

}  // namespace ORC_NS
