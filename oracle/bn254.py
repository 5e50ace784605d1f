"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the BN254 slice: halo2curves bn256::Fr arithmetic as Python integers, the
Fiat-Shamir challenge chain over Fr, and gkr::sum_check::prove_sum_check for the three shapes of the hot path.

Only tests/ may import this module; the product (hyper-greco_amd/csrc/bn254.hip) never does.

Follows [REF bfv-gkr/src/transcript.rs:146-157,198-203] (challenge = fe_mod_from_le_bytes(Keccak state), state re-hashed,
prover messages never absorbed), [REF lasso/src/lasso.rs:457-475] (collation g = poly(0) * sum_i M^i poly(i)),
[REF lasso/src/memory_checking/prover.rs:268-279] (grand-product g = poly(0) * sum_i gamma^i poly(2i) poly(2i+1)) and the
Libra / zkCNN pair-product sums of the external `gkr` crate.

PARITY UNPINNED for the conventions that live in the un-vendored `gkr` crate (same C1-C4 as the Goldilocks oracle,
DESIGN.md 2): a round message is the d+1 coefficients of the round polynomial, eval(1) = claim - eval(0), the lowest
variable is bound first. Pinned: the field modulus, and the first challenge over Fr (SURVEY.md 8(c)(5):
7173236656320612194178997223602979818891828541827642103715116037219761443523)."""

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # bn256::Fr modulus


def challenges(n, keccak256):
    """c_j = int.from_bytes(Keccak^j(""), "little") mod r, j = 1..n. `keccak256`: bytes -> 32 bytes."""
    out, h = [], keccak256(b"")
    for _ in range(n):
        out.append(int.from_bytes(h, "little") % R)
        h = keccak256(h)
    return out


def sumcheck(kind, tables, pw, claim, chal):
    """kind 0: g = p0 * sum_i pw_i p_i (deg 2); 1: g = p0 * sum_i pw_i p_2i p_2i+1 (deg 3); 2: g = sum_i p_2i p_2i+1 (deg 2).
    tables: lists of 2^nv ints; chal: the nv round challenges. Returns (msgs, evals, sums): per round the d+1
    coefficients, the fully folded table values, and the raw per-round sums g(0), g(2)[, g(3)]."""
    tabs = [list(t) for t in tables]
    nv = (len(tabs[0]) - 1).bit_length()
    d = 3 if kind == 1 else 2
    inv2, inv3, inv6 = pow(2, -1, R), pow(3, -1, R), pow(6, -1, R)
    msgs, sums = [], []
    for rd in range(nv):
        half = len(tabs[0]) // 2
        pts = [0, 2, 3][:d]
        ev = {}
        for t in pts:
            acc = 0
            for j in range(half):
                at = [(T[2 * j] + t * (T[2 * j + 1] - T[2 * j])) % R for T in tabs]
                if kind == 0:
                    s = sum(pw[i] * at[i] for i in range(len(tabs))) % R
                    acc += at[0] * s
                elif kind == 1:
                    s = sum(pw[i] * at[2 * i] * at[2 * i + 1] for i in range(len(tabs) // 2)) % R
                    acc += at[0] * s
                else:
                    acc += sum(at[2 * i] * at[2 * i + 1] for i in range(len(tabs) // 2))
            ev[t] = acc % R
        sums.append([ev[t] for t in pts])
        e0, e1, e2 = ev[0], (claim - ev[0]) % R, ev[2]
        d1 = (e1 - e0) % R
        d2 = (e2 - 2 * e1 + e0) % R
        if d == 2:
            c2 = d2 * inv2 % R
            c = [e0, (d1 - c2) % R, c2]
        else:
            d3 = (ev[3] - e0 - 3 * (e2 - e1)) % R
            c = [e0, (d1 - d2 * inv2 + d3 * inv3) % R, (d2 - d3) * inv2 % R, d3 * inv6 % R]
        msgs.append(c)
        r = chal[rd]
        claim = sum(c[k] * pow(r, k, R) for k in range(d + 1)) % R
        tabs = [[(T[2 * j] + r * (T[2 * j + 1] - T[2 * j])) % R for j in range(half)] for T in tabs]
    return msgs, [T[0] for T in tabs], sums


def verify_sumcheck(kind, msgs, evals, pw, claim, chal):
    """The verifier's side: every round polynomial must sum to the running claim; the last claim must equal g at the
    folded values. Returns True / False."""
    d = 3 if kind == 1 else 2
    for rd, c in enumerate(msgs):
        if (2 * c[0] + sum(c[1:])) % R != claim % R:  # h(0) + h(1)
            return False
        claim = sum(c[k] * pow(chal[rd], k, R) for k in range(d + 1)) % R
    if kind == 0:
        g = evals[0] * sum(pw[i] * evals[i] for i in range(len(evals)))
    elif kind == 1:
        g = evals[0] * sum(pw[i] * evals[2 * i] * evals[2 * i + 1] for i in range(len(evals) // 2))
    else:
        g = sum(evals[2 * i] * evals[2 * i + 1] for i in range(len(evals) // 2))
    return g % R == claim % R


ROOT_OF_UNITY_2_28 = pow(7, (R - 1) >> 28, R)  # halo2curves bn256::Fr::ROOT_OF_UNITY (S = 28, generator 7)


def root_of_unity(log2n):
    return pow(ROOT_OF_UNITY_2_28, 1 << (28 - log2n), R)


def ntt(a, inverse=False):
    """out[k] = sum_j a[j] w^(jk) (definition, O(n^2)); inverse: w^-1 and 1/n. Natural order in and out."""
    n = len(a)
    w = root_of_unity(n.bit_length() - 1)
    if inverse:
        w = pow(w, -1, R)
    out = [sum(a[j] * pow(w, j * k, R) for j in range(n)) % R for k in range(n)]
    if inverse:
        ninv = pow(n, -1, R)
        out = [x * ninv % R for x in out]
    return out


def mle_eval(table, point):
    """Multilinear extension of `table` (little-endian variable order) at `point`."""
    t = list(table)
    for r in point:
        t = [(t[2 * j] + r * (t[2 * j + 1] - t[2 * j])) % R for j in range(len(t) // 2)]
    return t[0]


def grand_product(tables, chal):
    """prove_grand_product [REF lasso/src/memory_checking/prover.rs:183-266]: tables = nb lists of 2^nv values; chal = the challenge
    stream from the point where the protocol is entered (list). Returns (proof elements in wire order, final claims, point)."""
    nb, ln = len(tables), len(tables[0])
    nv = ln.bit_length() - 1
    lev = [[list(t) for t in tables]]
    for k in range(1, nv):
        prev = lev[-1]
        h = len(prev[0]) // 2
        lev.append([[p[i] * p[i + h] % R for i in range(h)] for p in prev])   # Layer::bottom / Layer::up (MSB split)
    top = lev[nv - 1]
    proof = []
    claims = [t[0] * t[1] % R for t in top]
    proof += claims                                                                # root products (:197-221)
    it = iter(chal)
    x = []
    for n in range(nv):
        rows = lev[nv - 1 - n]
        h = 1 << n
        if n == 0:
            x = []
            evals = [v for t in rows for v in (t[0], t[1])]
        else:
            gamma = next(it)                                                       # :238
            pw = [pow(gamma, b, R) for b in range(nb)]
            claim = sum(c * w for c, w in zip(claims, pw)) % R                     # :281-286
            tabs = [half for t in rows for half in (t[:h], t[h:])]
            rs = [next(it) for _ in range(n)]
            msgs, evals, _ = sumcheck(1, tabs, pw, claim, rs)                      # g = poly(0) * sum gamma^b v_l v_r (:268-279)
            for m in msgs:
                proof += m
            x = list(rs)
        proof += evals                                                             # :257
        mu = next(it)                                                              # :259
        claims = [(evals[2 * b] + mu * (evals[2 * b + 1] - evals[2 * b])) % R for b in range(nb)]   # :288-294
        x.append(mu)
    return proof, claims, x


def eq_table(r):
    """eq(r, k) for k < 2^len(r), little-endian variable order."""
    t = [1]
    for ri in r:
        hi = [v * ri % R for v in t]
        t = [(v - h) % R for v, h in zip(t, hi)] + hi
    return t


def lasso_prove(P, chal, sections=None):
    """LassoNode::prove_claim_reduction over Fr [REF lasso/src/lasso.rs:57-114, 254-288, 303-336; memory_checking/prover.rs:35-89,
    158-181; memory_checking/mod.rs:80-93] from the node's integer tables P (tests/orclib.py: lasso_polys - limb indices,
    counters and subtable values are field-independent integers): dict with nu, A, rows, dims[4][N], read_cts[A][N],
    final_cts[A][65536], e_polys[A][N], row_lookup[N], mem_dim[A], mem_cutoff[A], lookup_mems[l].
    chal: challenge stream from the point where the node is entered. Returns (proof elements in wire order, r, claimed_sum)."""
    nu, A, rows, N, M = P["nu"], P["A"], P["rows"], 1 << P["nu"], 65536
    it = iter(chal)
    r = [next(it) for _ in range(nu)]                                              # lasso.rs:85
    eq = eq_table(r)
    mp = [pow(M, i, R) for i in range(5)]
    claimed = 0
    for k in range(rows):                                                          # lasso.rs:422-454, range.rs:184-195
        mems = P["lookup_mems"][P["row_lookup"][k]]
        comb = sum(P["e_polys"][m][k] * mp[i] for i, m in enumerate(mems))
        claimed += eq[k] * comb
    claimed %= R
    proof = [claimed]                                                              # lasso.rs:269
    pw = [pow(M, i, R) for i in range(A)]                                          # distribute_powers(.., M) range.rs:197-204
    msgs, _, _ = sumcheck(0, P["e_polys"], pw, claimed, [next(it) for _ in range(nu)])   # lasso.rs:278-279 (result dropped :97)
    for m in msgs:
        proof += m
    gamma, tau = next(it), next(it)                                                # lasso.rs:99; as_bases()[0] is the element itself
    g2 = gamma * gamma % R
    h = lambda a, v, t: (a + v * gamma + t * g2 - tau) % R                         # prover.rs:44
    order = [(m, c) for c in sorted(set(P["mem_dim"])) for m in range(A) if P["mem_dim"][m] == c]   # lasso.rs:303-336
    tab = lambda m, a: a if a < P["mem_cutoff"][m] else 0
    rd, wr, init, fin = [], [], [], []
    for m, c in order:                                                             # prover.rs:35-89; counters indexed by chunk (quirk)
        dim, rts, fct, ep = P["dims"][c], P["read_cts"][c], P["final_cts"][c], P["e_polys"][m]
        init.append([h(a, tab(m, a), 0) for a in range(M)])
        fin.append([h(a, tab(m, a), fct[a]) for a in range(M)])
        rd.append([h(dim[j], ep[j], rts[j]) for j in range(N)])
        wr.append([h(dim[j], ep[j], rts[j] + 1) for j in range(N)])
    rest = list(it)
    need1 = 1 + sum(2 + n for n in range(1, nu))
    p1, c1, x = grand_product(rd + wr, rest[:need1])                               # prover.rs:161-165
    p2, c2, y = grand_product(init + fin, rest[need1:])                            # prover.rs:167-171
    if sections is not None:
        sections.update(gp1=len(proof), gp2=len(proof) + len(p1), openings=len(proof) + len(p1) + len(p2), x=x, y=y,
                        gp1_claims=c1, gp2_claims=c2, gamma=gamma, tau=tau, order=order)
    proof += p1 + p2
    for c in sorted(set(P["mem_dim"])):                                            # prover.rs:173-178, mod.rs:80-93
        proof += [mle_eval(P["dims"][c], x), mle_eval(P["read_cts"][c], x), mle_eval(P["final_cts"][c], y)]
        proof += [mle_eval(P["e_polys"][m], x) for m in range(A) if P["mem_dim"][m] == c]
    return proof, r, claimed


def lasso_challenge_count(nu):
    return nu + nu + 2 + (1 + sum(2 + n for n in range(1, nu))) + (1 + sum(2 + n for n in range(1, 16)))


def _verify_rounds(d, nv, claim, elems, pos, it):
    """verify_sum_check (gkr crate; conventions C1): each round polynomial must satisfy h(0) + h(1) = claim; claim <- h(r)."""
    point = []
    for _ in range(nv):
        c = elems[pos:pos + d + 1]
        pos += d + 1
        if (2 * c[0] + sum(c[1:])) % R != claim % R:
            raise ValueError("InvalidSumCheck: round polynomial does not match claim")
        r = next(it)
        claim = sum(ck * pow(r, k, R) for k, ck in enumerate(c)) % R
        point.append(r)
    return claim, point, pos


def verify_grand_product(num_vars, nb, elems, pos, it):
    """[REF lasso/src/memory_checking/verifier.rs:178-235] (the final sum-check claim of a layer is not checked there either)."""
    claims = elems[pos:pos + nb]
    pos += nb
    x = []
    for n in range(num_vars):
        if n == 0:
            evals = elems[pos:pos + 2 * nb]
            pos += 2 * nb
            for b in range(nb):
                if claims[b] != evals[2 * b] * evals[2 * b + 1] % R:
                    raise ValueError("InvalidSumCheck: unmatched sum check output")
            x = []
        else:
            gamma = next(it)
            claim = sum(c * pow(gamma, b, R) for b, c in enumerate(claims)) % R
            _, x, pos = _verify_rounds(3, n, claim, elems, pos, it)
            evals = elems[pos:pos + 2 * nb]
            pos += 2 * nb
        mu = next(it)
        claims = [(evals[2 * b] + mu * (evals[2 * b + 1] - evals[2 * b])) % R for b in range(nb)]
        x.append(mu)
    return claims, x, pos


def lasso_verify(elems, nu, mem_dim, mem_cutoff, chal, partial=False):
    """LassoNode::verify_claim_reduction over Fr [REF lasso/src/lasso.rs:116-139, memory_checking/verifier.rs:61-95, 130-176]:
    raises ValueError on the first failed check, returns (r, claimed_sum) otherwise. elems: the proof as integers."""
    A = len(mem_dim)
    it = iter(chal)
    r = [next(it) for _ in range(nu)]
    claimed = elems[0]
    _, _, pos = _verify_rounds(2, nu, claimed, elems, 1, it)        # result ignored (lasso.rs:129-130)
    gamma, tau = next(it), next(it)
    h = lambda a, v, t: (a + v * gamma + t * gamma * gamma - tau) % R
    rw, x, pos = verify_grand_product(nu, 2 * A, elems, pos, it)
    ifr, y, pos = verify_grand_product(16, 2 * A, elems, pos, it)
    id_y = sum((1 << i) * yi for i, yi in enumerate(y)) % R
    off = 0
    for c in sorted(set(mem_dim)):
        mems = [m for m in range(A) if mem_dim[m] == c]
        dim_x, rts_x, fct_y = elems[pos:pos + 3]
        e_xs = elems[pos + 3:pos + 3 + len(mems)]
        pos += 3 + len(mems)
        for q, m in enumerate(mems):
            t_y = mle_eval([a if a < mem_cutoff[m] else 0 for a in range(65536)], y)
            if rw[off + q] != h(dim_x, e_xs[q], rts_x): raise ValueError("memory check: read hash mismatch")
            if rw[A + off + q] != h(dim_x, e_xs[q], (rts_x + 1) % R): raise ValueError("memory check: write hash mismatch")
            if ifr[off + q] != h(id_y, t_y, 0): raise ValueError("memory check: init hash mismatch")
            if ifr[A + off + q] != h(id_y, t_y, fct_y): raise ValueError("memory check: final hash mismatch")
        off += len(mems)
    if partial:                      # the node inside a larger proof: also report what was consumed
        return r, claimed, pos
    if pos != len(elems):
        raise ValueError("trailing proof elements")
    return r, claimed
