"""The two algebraic identities the Lasso node's kernels rely on, checked on the CPU with plain Python integers over
Goldilocks / GoldilocksExt2 (X^2 = 7): no GPU, no oracle - this is the derivation of StJob::mk1 / mk2 and of the two-table
collation sum-check, executable.

1. Mirrored grand-product layer: every write row is its read row plus a constant c and carries kappa times its weight. Then for
   every round of the layer's sum-check (tables folded any number of times)
       sum_i w_i l_i r_i + kappa w_i (l_i + c)(r_i + c) = (1 + kappa) [ sum_i w_i l_i r_i + K1 S + K2 ],
   S = sum_i w_i (l_i + r_i), K1 = kappa c / (1 + kappa), K2 = kappa c^2 sum_i w_i / (1 + kappa); "write = read + c" survives a
   fold (affine, coefficients summing to one) and S folds like a table.
2. Collation: p0(t) * sum_m M^m E_m(t) only needs E_0 and C = sum_m M^m E_m, and fold(C) = sum_m M^m fold(E_m)."""
import random

P = 2**64 - 2**32 + 1


class E2:
    __slots__ = ("a", "b")

    def __init__(self, a, b=0):
        self.a, self.b = a % P, b % P

    def __add__(self, o):
        o = o if isinstance(o, E2) else E2(o)
        return E2(self.a + o.a, self.b + o.b)

    __radd__ = __add__

    def __sub__(self, o):
        o = o if isinstance(o, E2) else E2(o)
        return E2(self.a - o.a, self.b - o.b)

    def __mul__(self, o):
        o = o if isinstance(o, E2) else E2(o)
        return E2(self.a * o.a + 7 * self.b * o.b, self.a * o.b + self.b * o.a)

    __rmul__ = __mul__

    def __eq__(self, o):
        return self.a == o.a and self.b == o.b

    def inv(self):  # (a + bX)^-1 = (a - bX) / (a^2 - 7 b^2)
        d = pow((self.a * self.a - 7 * self.b * self.b) % P, P - 2, P)
        return E2(self.a * d, -self.b * d)


def rnd_e(rng):
    return E2(rng.randrange(P), rng.randrange(P))


def fold(tab, r):  # binds the lowest variable: t'[j] = t[2j] + r (t[2j+1] - t[2j])
    return [tab[2 * j] + r * (tab[2 * j + 1] - tab[2 * j]) for j in range(len(tab) // 2)]


def test_mirrored_grand_product_layer_identity_survives_every_round():
    rng = random.Random(2)
    G, n = 5, 16                                   # 5 memories, tables of 16 entries (4 rounds)
    gamma = rnd_e(rng)                             # the layer's batching challenge: weight of row b is gamma^b
    c = E2(rng.randrange(P))                       # gamma_memcheck^2: a base-field constant
    w = [E2(1)]
    for _ in range(1, 2 * G):
        w.append(w[-1] * gamma)
    kappa = w[G]                                   # weight of write row i = kappa * weight of read row i
    L = [[E2(rng.randrange(P)) for _ in range(n)] for _ in range(G)]   # left / right halves of the read rows (base field at round 0)
    R = [[E2(rng.randrange(P)) for _ in range(n)] for _ in range(G)]
    Lw = [[x + c for x in row] for row in L]       # the write rows
    Rw = [[x + c for x in row] for row in R]
    onek = E2(1) + kappa
    kp = kappa * onek.inv()
    lam = E2(0)
    for i in range(G):
        lam = lam + w[i]
    K1, K2 = kp * c, kp * c * c * lam
    S = [sum((w[i] * (L[i][k] + R[i][k]) for i in range(G)), E2(0)) for k in range(n)]
    while True:
        for k in range(len(S)):                    # the identity at every point of the current tables
            full = E2(0)
            reads = E2(0)
            for i in range(G):
                full = full + w[i] * L[i][k] * R[i][k] + w[G + i] * Lw[i][k] * Rw[i][k]
                reads = reads + w[i] * L[i][k] * R[i][k]
            assert full == onek * (reads + K1 * S[k] + K2)
            assert all(Lw[i][k] == L[i][k] + c and Rw[i][k] == R[i][k] + c for i in range(G))
        if len(S) == 1:
            break
        r = rnd_e(rng)                             # one sum-check round: fold everything, S as one more table
        L, R, Lw, Rw = [[fold(t, r) for t in T] for T in (L, R, Lw, Rw)]
        S_folded = fold(S, r)
        S = [sum((w[i] * (L[i][k] + R[i][k]) for i in range(G)), E2(0)) for k in range(len(S_folded))]
        assert all(S[k] == S_folded[k] for k in range(len(S)))
    # P_inf (a product of differences) only picks up the factor: (l_w(1) - l_w(0)) = (l(1) - l(0))


def test_collation_sum_check_needs_two_tables():
    rng = random.Random(3)
    A, n, M = 6, 16, 65536
    E = [[E2(rng.randrange(65536)) for _ in range(n)] for _ in range(A)]
    mp = [E2(pow(M, m, P)) for m in range(A)]
    C = [sum((mp[m] * E[m][k] for m in range(A)), E2(0)) for k in range(n)]
    p0 = E[0]
    while len(C) > 1:
        half = len(C) // 2
        # round sums at t = 0 and t = 2 from the A tables and from (E_0, C)
        for t in (0, 2):
            ev = lambda tab, j: tab[2 * j] + t * (tab[2 * j + 1] - tab[2 * j])
            many = sum((ev(p0, j) * sum((mp[m] * ev(E[m], j) for m in range(A)), E2(0)) for j in range(half)), E2(0))
            two = sum((ev(p0, j) * ev(C, j) for j in range(half)), E2(0))
            assert many == two
        r = rnd_e(rng)
        E = [fold(t, r) for t in E]
        C = fold(C, r)
        p0 = E[0]
        assert all(C[k] == sum((mp[m] * E[m][k] for m in range(A)), E2(0)) for k in range(len(C)))


def eq_table(z):  # eq(z, x) for x in {0,1}^n, little-endian: x_0 is the lowest bit
    tab = [E2(1)]
    for zk in z:
        tab = [t * (E2(1) - zk) for t in tab] + [t * zk for t in tab]
    return tab


def test_eq_factored_prodsum_rounds():
    """The eq-factored PRODSUM rounds (kernels.hpp PsJob::eq_n, kernels.hip ps_eq_step2_body), executable: a sum-check of
    g = sum_i a_i b_i whose bookkeeping tables are b_i = kappa_i eq(z', .) - z' may have Boolean coordinates (a wiring that relays one
    window of its input) - run two rounds per pass WITHOUT ever forming a b table:
      * A = sum_i kappa_i a_i; U_p = sum_j eq(z'_(t+2..); j) A[4j + p], p < 4;
      * round t:   s(0) = P_t (1 - z_t) S_0, s(2) = P_t (3 z_t - 1)(2 S_1 - S_0), S_0 = (1-z) U_0 + z U_2, S_1 = (1-z) U_1 + z U_3, z = z'_(t+1);
      * round t+1: the same with z_(t+1), P_(t+1) = P_t eq(z_t; r_t) and S'_0 = (1-r_t) U_0 + r_t U_1, S'_1 = (1-r_t) U_2 + r_t U_3;
      * every table folds twice in one step: a''[j] = sum_p c_p a[4j + p], c = (1-r_t, r_t) x (1-r_(t+1), r_(t+1)).
    Compared, round by round, with the plain sum-check on materialised b tables (s(0) = sum a(0) b(0), s(2) = sum a(2) b(2))."""
    rng = random.Random(5)
    n, npairs = 6, 3
    z = [rnd_e(rng) for _ in range(n - 2)] + [E2(1), E2(0)]          # the two top coordinates Boolean: a window of the input
    kappa = [rnd_e(rng) for _ in range(npairs)]
    a = [[E2(rng.randrange(P)) for _ in range(1 << n)] for _ in range(npairs)]
    b = [[kappa[i] * e for e in eq_table(z)] for i in range(npairs)]
    rs = [rnd_e(rng) for _ in range(n)]
    # reference: the plain rounds
    ref = []
    ta, tb = [list(t) for t in a], [list(t) for t in b]
    for t in range(n):
        s0 = sum((ta[i][2 * j] * tb[i][2 * j] for i in range(npairs) for j in range(len(ta[i]) // 2)), E2(0))
        s2 = sum(((2 * ta[i][2 * j + 1] - ta[i][2 * j]) * (2 * tb[i][2 * j + 1] - tb[i][2 * j]) for i in range(npairs) for j in range(len(ta[i]) // 2)), E2(0))
        ref.append((s0, s2))
        ta = [fold(x, rs[t]) for x in ta]
        tb = [fold(x, rs[t]) for x in tb]
    # eq-factored: two rounds per pass, no b table
    A = [sum((kappa[i] * a[i][x] for i in range(npairs)), E2(0)) for x in range(1 << n)]
    tabs = [list(t) for t in a]
    Pt = E2(1)
    for t in range(0, n - 2, 2):          # (the kernels hand the last rounds to the tail, which forms b = kappa_i P eq(z'_(t..); .) there)
        ra, rb = rs[t], rs[t + 1]
        suffix = eq_table(z[t + 2:])
        U = [sum((suffix[j] * A[4 * j + p] for j in range(len(A) // 4)), E2(0)) for p in range(4)]
        zt, zt1 = z[t], z[t + 1]
        S0, S1 = (E2(1) - zt1) * U[0] + zt1 * U[2], (E2(1) - zt1) * U[1] + zt1 * U[3]
        assert (Pt * (E2(1) - zt) * S0, Pt * (3 * zt - E2(1)) * (2 * S1 - S0)) == ref[t], t
        Pt1 = Pt * (zt * ra + (E2(1) - zt) * (E2(1) - ra))
        T0, T1 = (E2(1) - ra) * U[0] + ra * U[1], (E2(1) - ra) * U[2] + ra * U[3]
        assert (Pt1 * (E2(1) - zt1) * T0, Pt1 * (3 * zt1 - E2(1)) * (2 * T1 - T0)) == ref[t + 1], t + 1
        c = [(E2(1) - ra) * (E2(1) - rb), ra * (E2(1) - rb), (E2(1) - ra) * rb, ra * rb]
        dfold = lambda tab: [sum((c[p] * tab[4 * j + p] for p in range(4)), E2(0)) for j in range(len(tab) // 4)]
        assert dfold(A) == fold(fold(A, ra), rb)
        A = dfold(A)
        tabs = [dfold(x) for x in tabs]
        Pt = Pt1 * (zt1 * rb + (E2(1) - zt1) * (E2(1) - rb))
    # the hand-off: b_i = kappa_i P eq(z'_(t..); .) on the remaining variables, a_i the folded tables
    t = n - 2
    assert tabs == [fold(fold(fold(fold(x, rs[0]), rs[1]), rs[2]), rs[3]) for x in a]
    suffix = eq_table(z[t:])
    for i in range(npairs):
        bi = b[i]
        for q in range(t):
            bi = fold(bi, rs[q])
        assert bi == [kappa[i] * Pt * e for e in suffix]
