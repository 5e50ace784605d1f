"""CPU oracle (TEST INFRASTRUCTURE ONLY): the whole BFV secret-key-encryption proof over bn256::Fr (BASELINE config 5
family, `generate_sk_enc_test!("bn254", Fr, Fr, ..)`), as Python integers. Restates oracle/bfv.hpp + oracle/gkr.hpp
(the Goldilocks oracle) with E = F = Fr; the Lasso node comes from oracle/bn254.py.

Only tests/ may import this module; the product (hyper-greco_amd/csrc/bn254_gkr.hip) never does.

  configure / get_inputs / prove / verify   [REF bfv-gkr/src/sk_encryption_circuit.rs:86-293, 351-363, 365-415, 417-460, 462-517]
  Poly::{new, new_padded, new_shifted}      [REF bfv-gkr/src/poly.rs:12-44]
  VanillaNode (Libra), FftNode (zkCNN), prove_gkr / verify_gkr: external `gkr` crate - PARITY UNPINNED, conventions G1-G4 and
  C1-C4 exactly as in oracle/gkr.hpp and oracle/sumcheck.hpp.
"""
import heapq

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
GL_P = 0xFFFFFFFF00000001
INPUT, VANILLA, FFT, LASSO = range(4)


def lift_signed(v):
    """Goldilocks-form small signed integer (negative z stored as p - |z|) -> the same integer in Fr (r - |z|)."""
    v = int(v)
    return v if v < (1 << 63) else (R - (GL_P - v)) % R


# ---- circuit (sk_encryption_circuit.rs:86-293; node insertion order = NodeId order) ------------------------------------------
class Node:
    def __init__(self, kind, **kw):
        self.kind, self.preds, self.succs = kind, [], []
        self.log2_size = kw.get("log2_size", 0)
        self.inverse = kw.get("inverse", False)
        self.arity = self.log2_sub_in = self.log2_sub_out = self.log2_reps = self.num_gates = 0
        self.w0, self.lin, self.mul = [], [], []   # (gate, c) / (gate, in, j, c) / (gate, i0, j0, i1, j1, c)

    def log2_out(self):
        return self.log2_size if self.kind in (INPUT, FFT) else (self.log2_sub_out + self.log2_reps if self.kind == VANILLA else 0)


class Gates:
    def __init__(self, arity, log2_sub_in, reps):
        self.n = Node(VANILLA)
        self.n.arity, self.n.log2_sub_in, self.n.log2_reps = arity, log2_sub_in, reps.bit_length() - 1
        self.g = 0

    def relay(self, i, j, c=1, add=0):
        if add:
            self.n.w0.append((self.g, add))
        self.n.lin.append((self.g, i, j, c))
        self.g += 1

    def zero(self):
        self.g += 1

    def mul(self, i0, j0, i1, j1):
        self.n.mul.append((self.g, i0, j0, i1, j1, 1))
        self.g += 1

    def sum(self, terms):
        for i, j in terms:
            self.n.lin.append((self.g, i, j, 1))
        self.g += 1

    def done(self):
        self.n.num_gates = self.g
        self.n.log2_sub_out = max(self.g - 1, 0).bit_length()
        return self.n


class Circuit:
    def __init__(self):
        self.nodes = []

    def insert(self, n):
        self.nodes.append(n)
        return len(self.nodes) - 1

    def connect(self, a, b):
        self.nodes[b].preds.append(a)
        self.nodes[a].succs.append(b)

    def topo(self):  # G1: Kahn, smallest NodeId first
        indeg = [len(n.preds) for n in self.nodes]
        heap = [i for i, d in enumerate(indeg) if d == 0]
        heapq.heapify(heap)
        order = []
        while heap:
            u = heapq.heappop(heap)
            order.append(u)
            for v in self.nodes[u].succs:
                indeg[v] -= 1
                if indeg[v] == 0:
                    heapq.heappush(heap, v)
        assert len(order) == len(self.nodes)
        return order


def build_circuit(c):
    """c: constants dict (n, k, qis, k0is, *_bound(s)). Returns (Circuit, lasso_in_id, lasso_id, sum_id)."""
    n, k = c["n"], c["k"]
    P, L = n.bit_length() - 1, n.bit_length()
    SZ = 1 << L
    C = Circuit()
    inp = lambda log2, reps: C.insert(Node(INPUT, log2_size=log2 + reps.bit_length() - 1))
    s, e, k1 = inp(L, 1), inp(L, 1), inp(L, 1)                                     # :358-360
    g = Gates(1, L, 1)
    for i in range(k):
        for j in range(SZ):
            g.relay(0, j)
    es = C.insert(g.done())                                                        # :97-103
    g = Gates(1, L, 1)
    for i in range(k):
        for j in range(SZ):
            g.relay(0, j, c["k0is"][i])
    k1kis = C.insert(g.done())                                                     # :105-115
    C.connect(e, es)
    C.connect(k1, k1kis)
    ais = [inp(L, 1) for _ in range(k)]                                            # :122-124
    r1is = [inp(L, 1) for _ in range(k)]                                           # :126-128
    g = Gates(k, L, 1)
    for i in range(k):
        for j in range(SZ):
            g.relay(i, j, c["qis"][i])
    r1iqis = C.insert(g.done())                                                    # :130-141
    for i in range(k):
        C.connect(r1is[i], r1iqis)
    r2is = inp(P, k)                                                               # :147
    r2l = P + k.bit_length() - 1
    chunks = []
    for st in range(0, 1 << r2l, SZ):                                              # :149-161
        g = Gates(1, r2l, 1)
        en = min(st + SZ, 1 << r2l)
        for j in range(st, en):
            g.relay(0, j)
        for j in range(en - st, SZ):
            g.zero()
        node = C.insert(g.done())
        C.connect(r2is, node)
        chunks.append(node)
    bounds = [c["r1_bounds"][i] for i in range(k)] + [c["r2_bounds"][0]] * len(chunks) + [c["s_bound"], c["e_bound"], c["k1_bound"]]
    g = Gates(len(bounds), L, 1)                                                   # lasso_inputs_batched :163-181
    for i, b in enumerate(bounds):
        for j in range(SZ):
            g.relay(i, j, 1, b)
    lasso_in = C.insert(g.done())
    lasso = C.insert(Node(LASSO))                                                  # :182-210
    for i in range(k):
        C.connect(r1is[i], lasso_in)
    for ch in chunks:
        C.connect(ch, lasso_in)
    C.connect(s, lasso_in)
    C.connect(e, lasso_in)
    C.connect(k1, lasso_in)
    C.connect(lasso_in, lasso)
    s_eval = C.insert(Node(FFT, log2_size=L))                                      # :224-225
    C.connect(s, s_eval)
    g = Gates(1, L, 1)
    for j in range(SZ):
        g.relay(0, j)
    s_eval_copy = C.insert(g.done())                                               # :227-235
    C.connect(s_eval, s_eval_copy)
    g = Gates(k, L, 1)
    for i in range(k):
        for j in range(SZ):
            g.relay(i, j)
    sai_par = C.insert(g.done())                                                   # :237-243
    for i in range(k):                                                             # :245-260
        ai_eval = C.insert(Node(FFT, log2_size=L))
        g = Gates(2, L, 1)
        for j in range(SZ):
            g.mul(0, j, 1, j)
        sai_eval = C.insert(g.done())
        sai = C.insert(Node(FFT, log2_size=L, inverse=True))
        C.connect(ais[i], ai_eval)
        C.connect(s_eval_copy, sai_eval)
        C.connect(ai_eval, sai_eval)
        C.connect(sai_eval, sai)
        C.connect(sai, sai_par)
    r2sz = (1 << P) - 1                                                            # r2i_cyclo :262-278
    g = Gates(1, P, k)
    for j in range(r2sz):
        g.relay(0, j)
    g.zero()
    for j in range(r2sz):
        g.relay(0, j)
    g.zero()
    cyclo = C.insert(g.done())
    g = Gates(5, L, k)                                                             # sum :280-285
    for j in range(SZ):
        g.sum([(i, j) for i in range(5)])
    sum_id = C.insert(g.done())
    C.connect(r2is, cyclo)
    for x in (sai_par, es, k1kis, r1iqis, cyclo):
        C.connect(x, sum_id)
    return C, lasso_in, lasso, sum_id


def layout_inputs(n, k, w):
    """get_inputs / Poly::{new_padded,new_shifted} (sk_encryption_circuit.rs:365-415, poly.rs:12-44); w: the JSON dict
    (decimal strings, highest degree first). Returns the input tables in chain_par! order (:408) and ct0is."""
    L = n.bit_length()
    SZ = 1 << L
    ints = lambda x: [int(v) % R for v in x]
    padded = lambda x: ints(x) + [0] * (SZ - len(x))

    def shifted(x, size):
        v = [0] * max(size - len(x), 0) + ints(x)
        npow = 1 << (size - 1).bit_length()
        return v + [0] * (npow - len(v))

    inputs = [padded(w["s"]), shifted(w["e"], SZ - 1), shifted(w["k1"], SZ - 1)]
    inputs += [padded(w["ais"][z]) for z in range(k)]
    inputs += [padded(w["r1is"][z]) for z in range(k)]
    inputs.append([v for z in range(k) for v in ints(w["r2is"][z]) + [0]])
    ct0is = [v for z in range(k) for v in shifted(w["ct0is"][z], SZ)[1:] + [0]]
    return inputs, ct0is


# ---- evaluation -----------------------------------------------------------------------------------------------------------------
def root_of_unity(log2n):
    return pow(pow(7, (R - 1) >> 28, R), 1 << (28 - log2n), R)


def ntt(a, inverse=False):
    """Natural order in and out: out[z] = sum_x a[x] w^(xz); inverse: w^-1 and 1/N (radix 2)."""
    N = len(a)
    L = N.bit_length() - 1
    out = [0] * N
    for i in range(N):
        out[int(format(i, "0%db" % L)[::-1], 2) if L else 0] = a[i]
    w = root_of_unity(L)
    if inverse:
        w = pow(w, -1, R)
    for s in range(1, L + 1):
        m, h = 1 << s, 1 << (s - 1)
        wm = pow(w, N >> s, R)
        tw = [1] * h
        for j in range(1, h):
            tw[j] = tw[j - 1] * wm % R
        for k0 in range(0, N, m):
            for j in range(h):
                t = tw[j] * out[k0 + j + h] % R
                u = out[k0 + j]
                out[k0 + j] = (u + t) % R
                out[k0 + j + h] = (u - t) % R
    if inverse:
        ninv = pow(N, -1, R)
        out = [x * ninv % R for x in out]
    return out


def vanilla_evaluate(n, ins):
    G, S, Rp = 1 << n.log2_sub_out, 1 << n.log2_sub_in, 1 << n.log2_reps
    out = [0] * (G * Rp)
    for rep in range(Rp):
        o, b = rep * G, rep * S
        for gte, c in n.w0:
            out[o + gte] += c
        for gte, i, j, c in n.lin:
            out[o + gte] += c * ins[i][b + j]
        for gte, i0, j0, i1, j1, c in n.mul:
            out[o + gte] += c * ins[i0][b + j0] * ins[i1][b + j1]
    return [v % R for v in out]


def circuit_evaluate(C, inputs):
    vals = [None] * len(C.nodes)
    it = iter(inputs)
    for i, n in enumerate(C.nodes):
        if n.kind == INPUT:
            vals[i] = next(it)
            assert len(vals[i]) == 1 << n.log2_size
    for i in C.topo():
        n = C.nodes[i]
        ins = [vals[p] for p in n.preds]
        if n.kind == VANILLA:
            vals[i] = vanilla_evaluate(n, ins)
        elif n.kind == FFT:
            vals[i] = ntt(ins[0], n.inverse)
        elif n.kind == LASSO:
            vals[i] = [0]                                                          # lasso.rs:53-55
    return vals


# ---- sum-check (C1-C4) -------------------------------------------------------------------------------------------------------
def eq_table(r):
    t = [1]
    for ri in r:
        hi = [v * ri % R for v in t]
        t = [(v - h) % R for v, h in zip(t, hi)] + hi
    return t


def mle_eval(table, point):
    t = list(table)
    for r in point:
        t = [(t[2 * j] + r * (t[2 * j + 1] - t[2 * j])) % R for j in range(len(t) // 2)]
    return t[0]


def prodsum_prove(pairs, claim, rs, proof):
    """prove_sum_check of g = sum_i a_i b_i; appends the 3 coefficients per round; returns (next claim, folded values)."""
    tabs = [list(t) for pr in pairs for t in pr]
    inv2 = pow(2, -1, R)
    for r in rs:
        half = len(tabs[0]) // 2
        e0 = e2 = 0
        for q in range(0, len(tabs), 2):
            A, Bt = tabs[q], tabs[q + 1]
            for j in range(half):
                a0, a1, b0, b1 = A[2 * j], A[2 * j + 1], Bt[2 * j], Bt[2 * j + 1]
                e0 += a0 * b0
                e2 += (2 * a1 - a0) * (2 * b1 - b0)
        e0 %= R
        e2 %= R
        e1 = (claim - e0) % R
        c2 = (e2 - 2 * e1 + e0) * inv2 % R
        c = [e0, (e1 - e0 - c2) % R, c2]
        proof += c
        claim = (c[0] + c[1] * r + c[2] * r * r) % R
        tabs = [[(T[2 * j] + r * (T[2 * j + 1] - T[2 * j])) % R for j in range(half)] for T in tabs]
    return claim, [T[0] for T in tabs]


def prodsum_verify(nv, claim, el, it):
    point = []
    for _ in range(nv):
        c = [next(el) for _ in range(3)]
        if (2 * c[0] + c[1] + c[2]) % R != claim % R:
            raise ValueError("InvalidSumCheck: round polynomial does not match claim")
        r = next(it)
        claim = (c[0] + c[1] * r + c[2] * r * r) % R
        point.append(r)
    return claim, point


# ---- nodes (G2-G4) -------------------------------------------------------------------------------------------------------------
def combined_eq(cl, alpha):
    eqc = [0] * (1 << len(cl[0][0]))
    for (pt, _), a in zip(cl, alpha):
        for i, v in enumerate(eq_table(pt)):
            eqc[i] = (eqc[i] + v * a) % R
    return eqc


def vanilla_use(n):
    left, right = [False] * n.arity, [False] * n.arity
    for t in n.lin:
        left[t[1]] = True
    for t in n.mul:
        left[t[1]] = True
        right[t[3]] = True
    return left, right


def vanilla_prove(n, cl, alpha, ins, it, proof):
    G, S, Rp = 1 << n.log2_sub_out, 1 << n.log2_sub_in, 1 << n.log2_reps
    nin = n.log2_sub_in + n.log2_reps
    eqc = combined_eq(cl, alpha)
    claim = sum(v * a for (_, v), a in zip(cl, alpha)) % R
    for rep in range(Rp):
        for gte, c in n.w0:
            claim -= eqc[rep * G + gte] * c
    claim %= R
    left, right = vanilla_use(n)
    T = [[0] * (S * Rp) if left[i] else None for i in range(n.arity)]
    for rep in range(Rp):
        for gte, i, j, c in n.lin:
            T[i][rep * S + j] += eqc[rep * G + gte] * c
        for gte, i0, j0, i1, j1, c in n.mul:
            T[i0][rep * S + j0] += eqc[rep * G + gte] * c * ins[i1][rep * S + j1]
    li = [i for i in range(n.arity) if left[i]]
    rs = [next(it) for _ in range(nin)]
    after1, ev = prodsum_prove([(ins[i], [v % R for v in T[i]]) for i in li], claim, rs, proof)
    u = [0] * n.arity
    sub = [[] for _ in range(n.arity)]
    for q, i in enumerate(li):
        u[i] = ev[2 * q]
        proof.append(u[i])
        sub[i].append((rs, u[i]))
    if n.mul:
        eqx = eq_table(rs)
        claim2 = after1
        for rep in range(Rp):
            for gte, i, j, c in n.lin:
                claim2 -= u[i] * eqc[rep * G + gte] * c * eqx[rep * S + j]
        claim2 %= R
        Bt = [[0] * (S * Rp) if right[i] else None for i in range(n.arity)]
        for rep in range(Rp):
            for gte, i0, j0, i1, j1, c in n.mul:
                Bt[i1][rep * S + j1] += eqc[rep * G + gte] * c * eqx[rep * S + j0] % R * u[i0]
        ri = [i for i in range(n.arity) if right[i]]
        rs2 = [next(it) for _ in range(nin)]
        _, ev2 = prodsum_prove([(ins[i], [v % R for v in Bt[i]]) for i in ri], claim2, rs2, proof)
        for q, i in enumerate(ri):
            proof.append(ev2[2 * q])
            sub[i].append((rs2, ev2[2 * q]))
    return sub


def vanilla_verify(n, cl, alpha, it, el):
    G, S, Rp = 1 << n.log2_sub_out, 1 << n.log2_sub_in, 1 << n.log2_reps
    nin = n.log2_sub_in + n.log2_reps
    eqc = combined_eq(cl, alpha)
    claim = sum(v * a for (_, v), a in zip(cl, alpha)) % R
    for rep in range(Rp):
        for gte, c in n.w0:
            claim -= eqc[rep * G + gte] * c
    claim %= R
    left, right = vanilla_use(n)
    fin1, rx = prodsum_verify(nin, claim, el, it)
    u = [0] * n.arity
    sub = [[] for _ in range(n.arity)]
    for i in range(n.arity):
        if left[i]:
            u[i] = next(el)
            sub[i].append((rx, u[i]))
    eqx = eq_table(rx)
    lin_part = 0
    for rep in range(Rp):
        for gte, i, j, c in n.lin:
            lin_part += u[i] * eqc[rep * G + gte] % R * c * eqx[rep * S + j]
    lin_part %= R
    if not n.mul:
        if fin1 != lin_part:
            raise ValueError("vanilla node: final evaluation mismatch")
        return sub
    fin2, ry = prodsum_verify(nin, (fin1 - lin_part) % R, el, it)
    w = [0] * n.arity
    for i in range(n.arity):
        if right[i]:
            w[i] = next(el)
            sub[i].append((ry, w[i]))
    eqy = eq_table(ry)
    fin = 0
    for rep in range(Rp):
        for gte, i0, j0, i1, j1, c in n.mul:
            fin += w[i1] * u[i0] % R * (eqc[rep * G + gte] * c % R) * (eqx[rep * S + j0] * eqy[rep * S + j1] % R)
    if fin2 != fin % R:
        raise ValueError("vanilla node: phase-2 final evaluation mismatch")
    return sub


def fft_table(r, L, inverse):
    """F(r, x) = scale * prod_b (1 + r_b (w^(2^b x) - 1)), x < 2^L (G4)."""
    N = 1 << L
    w = root_of_unity(L)
    if inverse:
        w = pow(w, -1, R)
    W = [1] * N
    for i in range(1, N):
        W[i] = W[i - 1] * w % R
    cur = [pow(N, -1, R) if inverse else 1]
    for bb in range(L - 1, -1, -1):
        sz = 1 << (L - bb)
        nxt = [0] * sz
        for x in range(sz):
            f = (1 + r[bb] * (W[(x << bb) & (N - 1)] - 1)) % R
            nxt[x] = cur[x & (sz // 2 - 1)] * f % R
        cur = nxt
    return cur


def fft_combined(n, cl, alpha):
    Fc = [0] * (1 << n.log2_size)
    for (pt, _), a in zip(cl, alpha):
        for i, v in enumerate(fft_table(pt, n.log2_size, n.inverse)):
            Fc[i] = (Fc[i] + v * a) % R
    return Fc


def fft_prove(n, cl, alpha, vin, it, proof):
    Fc = fft_combined(n, cl, alpha)
    claim = sum(v * a for (_, v), a in zip(cl, alpha)) % R
    rs = [next(it) for _ in range(n.log2_size)]
    _, ev = prodsum_prove([(vin, Fc)], claim, rs, proof)
    proof.append(ev[0])
    return [[(rs, ev[0])]]


def fft_verify(n, cl, alpha, it, el):
    claim = sum(v * a for (_, v), a in zip(cl, alpha)) % R
    fin, rx = prodsum_verify(n.log2_size, claim, el, it)
    u = next(el)
    if fin != u * mle_eval(fft_combined(n, cl, alpha), rx) % R:
        raise ValueError("fft node: final evaluation mismatch")
    return [[(rx, u)]]


# ---- drivers (sk_encryption_circuit.rs:417-460, 462-517) ---------------------------------------------------------------------
def prove(c, inputs, ct0is, chal, lasso_prove_fn, trace=None):
    """c: constants; inputs/ct0is from layout_inputs; chal: enough challenges (list); lasso_prove_fn(lasso_in_values, chal_rest)
    -> (proof elements, r, value, challenges used). Returns the proof as a list of Fr integers (wire order)."""
    C, lasso_in, lasso_id, sum_id = build_circuit(c)
    vals = circuit_evaluate(C, inputs)
    pos = [0]

    class It:
        def __iter__(self):
            return self

        def __next__(self):
            pos[0] += 1
            return chal[pos[0] - 1]

    it = It()
    ov = c["n"].bit_length() + c["k"].bit_length() - 1
    point = [next(it) for _ in range(ov)]                                          # :445
    value = mle_eval(ct0is, point)                                                 # :446
    claims = [[] for _ in C.nodes]
    claims[lasso_id].append(([], 0))                                               # :450
    claims[sum_id].append((point, value))
    proof = []
    for i in reversed(C.topo()):
        n = C.nodes[i]
        if n.kind == INPUT:
            continue
        cl = claims[i]
        assert cl, "node without claim"
        alpha = [next(it) for _ in cl] if len(cl) > 1 else [1]                     # G2
        if trace is not None:
            trace.append((len(proof), "node %d kind %d" % (i, n.kind)))
        ins = [vals[p] for p in n.preds]
        if n.kind == VANILLA:
            sub = vanilla_prove(n, cl, alpha, ins, it, proof)
        elif n.kind == FFT:
            sub = fft_prove(n, cl, alpha, ins[0], it, proof)
        else:
            els, r, v, used = lasso_prove_fn(ins[0], chal[pos[0]:])
            pos[0] += used
            proof += els
            sub = [[(r, v)]]
        for p, s in zip(n.preds, sub):
            claims[p] += s
    return proof, vals


def verify(c, inputs, ct0is, proof, chal, lasso_verify_fn):
    """Raises ValueError on the first failed check. lasso_verify_fn(elems_iterator_state...) see tests."""
    C, lasso_in, lasso_id, sum_id = build_circuit(c)
    pos = [0]

    class It:
        def __iter__(self):
            return self

        def __next__(self):
            pos[0] += 1
            return chal[pos[0] - 1]

    it = It()
    epos = [0]

    class El:
        def __iter__(self):
            return self

        def __next__(self):
            if epos[0] >= len(proof):
                raise ValueError("proof too short")
            epos[0] += 1
            return proof[epos[0] - 1]

    el = El()
    ov = c["n"].bit_length() + c["k"].bit_length() - 1
    point = [next(it) for _ in range(ov)]
    value = mle_eval(ct0is, point)
    claims = [[] for _ in C.nodes]
    claims[lasso_id].append(([], 0))
    claims[sum_id].append((point, value))
    for i in reversed(C.topo()):
        n = C.nodes[i]
        if n.kind == INPUT:
            continue
        cl = claims[i]
        alpha = [next(it) for _ in cl] if len(cl) > 1 else [1]
        if n.kind == VANILLA:
            sub = vanilla_verify(n, cl, alpha, it, el)
        elif n.kind == FFT:
            sub = fft_verify(n, cl, alpha, it, el)
        else:
            r, v, used_el, used_ch = lasso_verify_fn(proof[epos[0]:], chal[pos[0]:])
            epos[0] += used_el
            pos[0] += used_ch
            sub = [[(r, v)]]
        for p, s in zip(n.preds, sub):
            claims[p] += s
    if epos[0] != len(proof):
        raise ValueError("trailing proof elements")
    k = 0
    for i, n in enumerate(C.nodes):                                                # :512-516
        if n.kind != INPUT:
            continue
        for pt, v in claims[i]:
            if mle_eval(inputs[k], pt) != v:
                raise ValueError("input claim mismatch at input %d" % k)
        k += 1
    return True
