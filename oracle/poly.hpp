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
#include "field.hpp"
#include <algorithm>
#include <omp.h>

// threads for a parallel region over `work` items of which one thread should get at least `grain` (fork / join and per-thread set-up
// dominate small regions on many-core hosts)
#ifndef ORC_THREADS_FOR
#define ORC_THREADS_FOR
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
// phase timers of a prove (ORC_TIMES=1 prints them when the prove ends): where a many-core baseline spends its time
struct OrcTimes {
    std::map<std::string, double> ms;
    bool on = getenv("ORC_TIMES") != nullptr;
    void add(const char* k, double t0) { if (on) ms[k] += (omp_get_wtime() - t0) * 1e3; }
    void dump() { if (!on) return; for (auto& kv : ms) fprintf(stderr, "[oracle] %-28s %9.2f ms\n", kv.first.c_str(), kv.second); ms.clear(); }
};
static inline OrcTimes& orc_times() { static OrcTimes t; return t; }
static inline int orc_threads_for(size_t work, size_t grain) {
    const size_t mx = (size_t)omp_get_max_threads(), want = work / (grain ? grain : 1);
    return (int)std::max<size_t>(1, std::min(mx, want));
}
#endif

namespace ORC_NS {

static inline std::vector<E> eq_table(const E* r, size_t n) {
    std::vector<E> t((size_t)1 << n);
    t[0] = e_one();
    size_t s = 1;
    for (size_t i = 0; i < n; i++) {
#pragma omp parallel for schedule(static) num_threads(orc_threads_for(s, 4096))
        for (long long jj = 0; jj < (long long)s; jj++) {
            const size_t j = (size_t)jj;
            E hi = e_mul(t[j], r[i]);
            t[j + s] = hi;
            t[j] = e_sub(t[j], hi);
        }
        s <<= 1;
    }
    return t;
}
static inline std::vector<E> eq_table(const std::vector<E>& r) { return eq_table(r.data(), r.size()); }

// sum_j eq[j] * tab[j]  (tab in the base field), parallel over j
static inline E dot_eq_f(const std::vector<E>& eq, const F* tab, size_t n) {
    E total = e_zero();
#pragma omp parallel num_threads(orc_threads_for(n, 8192))
    {
        E a = e_zero();
#pragma omp for nowait
        for (long long j = 0; j < (long long)n; j++) {
            F v = tab[j];
            if (!f_is_zero(v)) a = e_add(a, e_mul_f(eq[j], v));
        }
#pragma omp critical
        total = e_add(total, a);
    }
    return total;
}
// the same for a table of small non-negative integers (limb indices, counters, subtable values)
static inline E dot_eq_u64(const std::vector<E>& eq, const uint64_t* tab, size_t n) {
    E total = e_zero();
#pragma omp parallel num_threads(orc_threads_for(n, 8192))
    {
        E a = e_zero();
#pragma omp for nowait
        for (long long j = 0; j < (long long)n; j++) {
            uint64_t v = tab[j];
            if (v) a = e_add(a, e_mul_f(eq[j], f_from_u64(v)));
        }
#pragma omp critical
        total = e_add(total, a);
    }
    return total;
}

static inline E mle_eval_f(const F* tab, size_t nvars, const E* pt) {
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

}  // namespace ORC_NS
