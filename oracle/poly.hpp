// ORACLE (test infrastructure). Multilinear-polynomial helpers, little-endian variable order:
// coordinate i <-> bit i of the table index (evidence in-tree: prover.rs:262,310-315 splits on
// the MSB and appends mu last; range.rs:19-26 evaluates sum point[i]*2^i).
//   eq_table   = plonkish MultilinearPolynomial::eq_xy            (call site lasso.rs:432)
//   mle_eval   = gkr BoxMultilinearPoly::evaluate                 (call sites mod.rs:80-93,
//                                                                  sk_encryption_circuit.rs:446)
//   fold       = gkr fix_var on the lowest variable               (inside prove_sum_check)
// Both external; their published semantics (MLE over the boolean hypercube) are restated.
#pragma once
#include <vector>
#include <cstdint>
#include <cstddef>
#include "gl.hpp"

namespace orc {

static inline std::vector<E> eq_table(const E* r, size_t n) {
    std::vector<E> t((size_t)1 << n);
    t[0] = e_one();
    size_t s = 1;
    for (size_t i = 0; i < n; i++) {
        for (size_t j = 0; j < s; j++) {
            E hi = e_mul(t[j], r[i]);
            t[j + s] = hi;
            t[j] = e_sub(t[j], hi);
        }
        s <<= 1;
    }
    return t;
}
static inline std::vector<E> eq_table(const std::vector<E>& r) { return eq_table(r.data(), r.size()); }

// sum_j eq[j] * tab[j]  (tab in base field), parallel over j
static inline E dot_eq_f(const std::vector<E>& eq, const uint64_t* tab, size_t n) {
    uint64_t s0 = 0, s1 = 0;
#pragma omp parallel
    {
        uint64_t a0 = 0, a1 = 0;
#pragma omp for nowait
        for (long long j = 0; j < (long long)n; j++) {
            uint64_t v = tab[j];
            if (v) { a0 = f_add(a0, f_mul(eq[j].c0, v)); a1 = f_add(a1, f_mul(eq[j].c1, v)); }
        }
#pragma omp critical
        { s0 = f_add(s0, a0); s1 = f_add(s1, a1); }
    }
    return E{s0, s1};
}

static inline E mle_eval_f(const uint64_t* tab, size_t nvars, const E* pt) {
    std::vector<E> eq = eq_table(pt, nvars);
    return dot_eq_f(eq, tab, (size_t)1 << nvars);
}
static inline E mle_eval_e(const E* tab, size_t nvars, const E* pt) {
    std::vector<E> eq = eq_table(pt, nvars);
    E s = e_zero();
    for (size_t j = 0; j < ((size_t)1 << nvars); j++) s = e_add(s, e_mul(eq[j], tab[j]));
    return s;
}

static inline E horner(const std::vector<E>& c, E x) {
    E r = e_zero();
    for (size_t i = c.size(); i-- > 0;) r = e_add(e_mul(r, x), c[i]);
    return r;
}

}  // namespace orc
