"""Pins the CPU oracle against everything the reference tree fixes for this path:
Keccak / challenge-chain KATs, field arithmetic vs Python big ints, the Lasso memory maps, the
range.rs sub-table identities, the reference's JSON witness fixtures (layout + circuit relation) and
prove -> verify acceptance. The oracle's sum-check / GKR-engine conventions are [RECALL] (parity
unpinned, see oracle/sumcheck.hpp, oracle/gkr.hpp)."""
import ctypes as C
import hashlib
import json
import os
import random

import numpy as np
import pytest

import orclib
from orclib import P, ptr

L = orclib.lib()


def keccak(b):
    out = (C.c_uint8 * 32)()
    L.orc_keccak256(b, C.c_size_t(len(b)), out)
    return bytes(out)


def test_keccak_kats():
    h1 = keccak(b"")
    assert h1.hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    h2 = keccak(h1)
    assert h2.hex() == "10ca3eff73ebec87d2394fc58560afeab86dac7a21f5e402ea0a55e5c8a6758f"
    # multi-block absorb (rate 136) against a value computed with the same sponge one-shot on split input is
    # not available in hashlib (sha3 != keccak padding); check length-boundary behaviour is at least stable
    assert keccak(b"a" * 135) != keccak(b"a" * 136) != keccak(b"a" * 137)
    # well-known Keccak-256 KAT: "abc"
    assert keccak(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_challenge_chain_kats():
    out = np.zeros(4, dtype=np.uint64)
    L.orc_challenge_chain(C.c_size_t(4), ptr(out))
    assert [int(x) for x in out] == [15017384644633299356, 6854594310142832579, 9149254073876997563, 1396060396769822097]
    # definition: LE integer of the hash mod p
    h = keccak(b"")
    assert int.from_bytes(h, "little") % P == int(out[0])


def _rand_felts(n, rng):
    edge = [0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, (1 << 63), P >> 1]
    v = edge + [rng.randrange(P) for _ in range(n - len(edge))]
    rng.shuffle(v)
    return v


def test_field_ops_vs_bigint():
    rng = random.Random(1)
    a = _rand_felts(4096, rng)
    b = _rand_felts(4096, rng)
    A, B = np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64)
    out = np.zeros_like(A)
    for op, f in [(0, lambda x, y: (x + y) % P), (1, lambda x, y: (x - y) % P), (2, lambda x, y: x * y % P)]:
        L.orc_f_binop(op, C.c_size_t(A.size), ptr(A), ptr(B), ptr(out))
        assert [int(x) for x in out] == [f(x, y) for x, y in zip(a, b)]
    nz = np.array([x for x in a if x], dtype=np.uint64)
    inv = np.zeros_like(nz)
    L.orc_f_binop(3, C.c_size_t(nz.size), ptr(nz), ptr(nz), ptr(inv))
    assert all(int(x) * int(y) % P == 1 for x, y in zip(nz, inv))


def test_ext2_ops_vs_bigint():
    rng = random.Random(2)
    n = 1024
    a = _rand_felts(2 * n, rng)
    b = _rand_felts(2 * n, rng)
    A, B = np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64)
    out = np.zeros_like(A)
    L.orc_e_binop(2, C.c_size_t(n), ptr(A), ptr(B), ptr(out))
    for i in range(n):
        a0, a1, b0, b1 = a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1]
        assert int(out[2 * i]) == (a0 * b0 + 7 * a1 * b1) % P  # X^2 = 7
        assert int(out[2 * i + 1]) == (a0 * b1 + a1 * b0) % P
    L.orc_e_binop(3, C.c_size_t(n), ptr(A), ptr(A), ptr(out))
    chk = np.zeros_like(A)
    L.orc_e_binop(2, C.c_size_t(n), ptr(A), ptr(out), ptr(chk))
    for i in range(n):
        if a[2 * i] or a[2 * i + 1]:
            assert (int(chk[2 * i]), int(chk[2 * i + 1])) == (1, 0)


def test_root_of_unity_and_ntt():
    w32 = L.orc_root_of_unity(C.c_size_t(32))
    assert w32 == pow(7, (P - 1) >> 32, P) == 1753635133440165772  # = plonky2's POWER_OF_TWO_GENERATOR
    w = L.orc_root_of_unity(C.c_size_t(5))
    assert pow(w, 32, P) == 1 and pow(w, 16, P) == P - 1
    rng = random.Random(3)
    x = [rng.randrange(P) for _ in range(32)]
    X = np.array(x, dtype=np.uint64)
    Y = np.zeros_like(X)
    L.orc_ntt(ptr(X), C.c_size_t(5), 0, ptr(Y))
    assert [int(v) for v in Y] == [sum(x[j] * pow(w, j * z, P) for j in range(32)) % P for z in range(32)]
    Z = np.zeros_like(X)
    L.orc_ntt(ptr(Y), C.c_size_t(5), 1, ptr(Z))
    assert (Z == X).all()


def test_fft_table_is_mle_of_dft_matrix():
    # F(r, x) = sum_z eq(r, z) w^(zx)  (zkCNN); check against the definition at L = 4
    rng = random.Random(4)
    Lg = 4
    r = np.array([rng.randrange(P) for _ in range(2 * Lg)], dtype=np.uint64)
    eq = np.zeros(2 << Lg, dtype=np.uint64)
    L.orc_eq_table(ptr(r), C.c_size_t(Lg), ptr(eq))
    for inv in (0, 1):
        tab = np.zeros(2 << Lg, dtype=np.uint64)
        L.orc_fft_table(ptr(r), C.c_size_t(Lg), inv, ptr(tab))
        w = L.orc_root_of_unity(C.c_size_t(Lg))
        if inv:
            w = pow(w, P - 2, P)
        sc = pow(1 << Lg, P - 2, P) if inv else 1
        for x in range(1 << Lg):
            c0 = sum(int(eq[2 * z]) * pow(w, z * x, P) for z in range(1 << Lg)) * sc % P
            c1 = sum(int(eq[2 * z + 1]) * pow(w, z * x, P) for z in range(1 << Lg)) * sc % P
            assert (int(tab[2 * x]), int(tab[2 * x + 1])) == (c0, c1)


def test_subtable_cutoffs_and_mle_identities():
    # b3 -> 5, b39 -> 71, b65537 -> 2
    assert L.orc_subtable_cutoff(C.c_uint64(3)) == 5
    assert L.orc_subtable_cutoff(C.c_uint64(39)) == 71
    assert L.orc_subtable_cutoff(C.c_uint64(65537)) == 2
    rng = random.Random(5)
    c3 = orclib.constants(32768, 16)
    bounds = [0, (1 << 55) + 55, 3, 39, 65537, 2493, 82638181, 27424203952895201]  # [REF range.rs:293-331] + fixtures
    bounds += [2 * b + 1 for b in c3["r1_bounds"]] + sorted(set(2 * b + 1 for b in c3["r2_bounds"]))
    for b in bounds:
        pt = np.array([rng.randrange(P) for _ in range(32)], dtype=np.uint64)
        d, c = np.zeros(2, dtype=np.uint64), np.zeros(2, dtype=np.uint64)
        assert L.orc_subtable_mle_identity(C.c_uint64(b), ptr(pt), ptr(d), ptr(c)) == 1, b
        # base-field point like the reference test (F::from(rng.next_u64()))
        pt[1::2] = 0
        assert L.orc_subtable_mle_identity(C.c_uint64(b), ptr(pt), ptr(d), ptr(c)) == 1, b


def test_lasso_memory_maps():
    # SURVEY.md §8(a) row A2 (computed from constants/*.rs + lasso.rs:527-627 + range.rs:207-250)
    mems, _ = orclib.lasso_layout(orclib.params(1024, 1))
    assert mems == ["bound_2493@0", "bound_3@0", "bound_39@0", "full@0", "bound_65537@1", "bound_82638181@1"]
    mems, _ = orclib.lasso_layout(orclib.params(4096, 2))
    assert mems == ["full@0", "full@1", "full@2", "bound_27424203952895201@3", "bound_3@0", "bound_39@0",
                    "bound_39007@0", "bound_51933@0", "bound_65537@1"]
    mems, lk = orclib.lasso_layout(orclib.params(32768, 16))
    assert len(mems) == 25 and len(lk) == 22
    short = [m.replace("bound_", "b") for m in mems]
    assert short[:10] == ["b3@0", "b34899@0", "b37227@0", "b39@0", "b40683@0", "b42261@0", "b46675@0", "full@0", "full@1", "full@2"]
    assert [m[-2:] for m in short[10:13]] == ["@3"] * 3 and short[10].startswith("b4775")
    assert short[13:] == ["b47943@0", "b50877@0", "b58647@0", "b65537@1", "b72431@1", "b77055@1", "b81125@1",
                          "b82453@1", "b82463@1", "b89479@1", "b94161@1", "b94357@1"]


# every Goldilocks witness the reference holds (bfv-gkr/src/data/goldilocks/); 8192 is its only one with k = 4, i.e. two r2is chunks
# [REF sk_encryption_circuit.rs:149-161]
FIX = [(1024, 1, 27), (2048, 1, 52), (4096, 2, 55), (8192, 4, 55)]
BN_FIX = [(1024, 1, 27), (2048, 1, 52), (4096, 2, 55)]   # bfv-gkr/src/data/bn254/


@pytest.mark.parametrize("n,k,bits", FIX)
def test_fixture_layout_and_circuit_relation(n, k, bits):
    p = orclib.params(n, k)
    inp = orclib.fixture_inputs(n, k, bits)
    lasso_in, sum_out, info = orclib.circuit_eval(p, inp)
    # the circuit's `sum` output reproduces the ct0is layout exactly
    assert (sum_out == inp.d["ct0is"]).all()
    chunks = max(1, k // 2)
    assert info["num_nodes"] == 5 * k + 14 + chunks
    assert info["rows"] == (k + chunks + 3) << p.L
    # every shifted range input lies in [0, 2*bound]
    SZ = 1 << p.L
    c = p.c
    bounds = c["r1_bounds"][:k] + [c["r2_bounds"][0]] * chunks + [c["s_bound"], c["e_bound"], c["k1_bound"]]
    for i, b in enumerate(bounds):
        assert int(lasso_in[i * SZ:(i + 1) * SZ].max()) <= 2 * b


@pytest.mark.parametrize("n,k,bits", FIX)
def test_prove_verify_roundtrip_and_tamper(n, k, bits):
    p = orclib.params(n, k)
    inp = orclib.fixture_inputs(n, k, bits)
    proof, _ = orclib.prove(p, inp, threads=4)
    ok, err = orclib.verify(p, inp, proof)
    assert ok, err
    # determinism and thread-count independence (exact field arithmetic)
    proof1, _ = orclib.prove(p, inp, threads=1)
    assert proof1 == proof
    # golden digest of the oracle's own transcript (regression pin; NOT a reference-produced value)
    gold = json.load(open(os.path.join(orclib.GOLDEN, "oracle_proof_digests.json")))
    assert hashlib.sha256(proof).hexdigest() == gold[f"{n}_{k}"]["sha256"]
    assert len(proof) == gold[f"{n}_{k}"]["len"]
    rng = random.Random(n)
    rejected = 0
    for _ in range(6):
        bad = bytearray(proof)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        ok, _ = orclib.verify(p, inp, bytes(bad))
        rejected += (not ok)
    assert rejected >= 4  # the reference verifier leaves some sections unbound
    # wrong public input
    d2 = dict(inp.d)
    d2["ct0is"] = inp.d["ct0is"].copy()
    d2["ct0is"][5] ^= np.uint64(1)
    ok, _ = orclib.verify(p, orclib.Inputs(d2), proof)
    assert not ok


def test_lasso_node_claim_is_mle_of_inputs():
    # `sanity-check` feature assertion [REF lasso.rs:265-267]: claimed_sum == lookup_output_poly.evaluate(r)
    p = orclib.params(1024, 1)
    inp = orclib.fixture_inputs(1024, 1, 27)
    lasso_in, _, info = orclib.circuit_eval(p, inp)
    proof, claim = orclib.lasso_prove(p, lasso_in)
    nu = info["nu"]
    out = np.zeros(2, dtype=np.uint64)
    L.orc_mle_eval_f(ptr(lasso_in), C.c_size_t(nu), ptr(claim[:2 * nu].copy()), ptr(out))
    assert (out == claim[2 * nu:]).all()
    ok, err = orclib.lasso_verify(p, proof)
    assert ok, err
    # Appendix C element count: 1 + nu*m2 + sum over both grand products + openings, m2 = 3, m3 = 4
    alpha, m2, m3 = 6, 3, 4
    def gp(nv):
        return 2 * alpha + nv * 4 * alpha + m3 * sum(range(1, nv))
    n_e = 1 + nu * m2 + gp(nu) + gp(16) + (3 * 2 + alpha)
    assert len(proof) == 16 * n_e


# ---- BN254 slice (oracle/bn254.py) -----------------------------------------------------------------------------------
def test_bn254_first_challenge_and_chain():
    """fe_mod_from_le_bytes(Keccak256("")) over bn256::Fr [REF transcript.rs:198-203]."""
    bn = orclib.bn254()
    c = bn.challenges(3, orclib.keccak256)
    assert c[0] == 7173236656320612194178997223602979818891828541827642103715116037219761443523
    h1 = bytes.fromhex("c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470")
    assert c[0] == int.from_bytes(h1, "little") % bn.R
    h2 = bytes.fromhex("10ca3eff73ebec87d2394fc58560afeab86dac7a21f5e402ea0a55e5c8a6758f")
    assert c[1] == int.from_bytes(h2, "little") % bn.R
    assert bn.R == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


@pytest.mark.parametrize("kind,ntab,nv", [(0, 3, 4), (1, 4, 3), (2, 6, 4), (1, 2, 1)])
def test_bn254_sumcheck_oracle_round_trip(kind, ntab, nv):
    """prove -> verify with the true claim; a wrong claim or a tampered message is rejected."""
    import random
    bn = orclib.bn254()
    rng = random.Random(17 * kind + ntab + nv)
    tabs = [[rng.randrange(bn.R) for _ in range(1 << nv)] for _ in range(ntab)]
    npw = ntab if kind == 0 else (ntab // 2 if kind == 1 else 0)
    pw = [rng.randrange(bn.R) for _ in range(npw)]
    # true sum over the hypercube of g
    def g(vals):
        if kind == 0: return vals[0] * sum(pw[i] * vals[i] for i in range(ntab))
        if kind == 1: return vals[0] * sum(pw[i] * vals[2 * i] * vals[2 * i + 1] for i in range(ntab // 2))
        return sum(vals[2 * i] * vals[2 * i + 1] for i in range(ntab // 2))
    claim = sum(g([T[x] for T in tabs]) for x in range(1 << nv)) % bn.R
    chal = bn.challenges(nv + 5, orclib.keccak256)[5:]
    msgs, evals, sums = bn.sumcheck(kind, tabs, pw, claim, chal)
    assert bn.verify_sumcheck(kind, msgs, evals, pw, claim, chal)
    assert not bn.verify_sumcheck(kind, msgs, evals, pw, (claim + 1) % bn.R, chal)
    bad = [list(m) for m in msgs]
    bad[-1][1] = (bad[-1][1] + 1) % bn.R
    assert not bn.verify_sumcheck(kind, bad, evals, pw, claim, chal)


def test_bn254_root_of_unity_and_ntt_oracle():
    """halo2curves bn256::Fr::ROOT_OF_UNITY = 7^((r-1)/2^28): the recalled constant, its order, and the O(n^2) NTT oracle."""
    bn = orclib.bn254()
    assert bn.ROOT_OF_UNITY_2_28 == 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
    assert pow(bn.ROOT_OF_UNITY_2_28, 1 << 28, bn.R) == 1 and pow(bn.ROOT_OF_UNITY_2_28, 1 << 27, bn.R) == bn.R - 1
    a = [3, 1, 4, 1, 5, 9, 2, 6]
    assert bn.ntt(bn.ntt(a), inverse=True) == a
    # convolution theorem on a cyclic product: (1 + x) * (1 + x) = 1 + 2x + x^2 mod x^8 - 1
    f = bn.ntt([1, 1, 0, 0, 0, 0, 0, 0])
    assert bn.ntt([x * x % bn.R for x in f], inverse=True) == [1, 2, 1, 0, 0, 0, 0, 0]
    assert bn.mle_eval([5, 7], [3]) == (5 + 3 * 2) % bn.R


def test_bn254_grand_product_oracle_claims_are_mle_evaluations():
    """Each layer reduces the product claim one level down; the final claims must equal the MLEs of the input tables at
    the final point, and the first proof elements are the plain products of the tables."""
    import random
    from functools import reduce
    bn = orclib.bn254()
    rng = random.Random(5)
    tabs = [[rng.randrange(bn.R) for _ in range(16)] for _ in range(3)]
    chal = bn.challenges(40, orclib.keccak256)
    proof, claims, point = bn.grand_product(tabs, chal)
    assert proof[:3] == [reduce(lambda a, b: a * b % bn.R, t, 1) for t in tabs]
    assert len(point) == 4 and claims == [bn.mle_eval(t, point) for t in tabs]


def test_bn254_lasso_oracle_passes_the_verifier_checks():
    """The checks of MemoryCheckingVerifier::verify_memories [REF lasso/src/memory_checking/verifier.rs:61-95] on the oracle's
    output over Fr, for the reference fixture's lookup table: the grand-product claims equal the multiset hash of the
    openings (reads / writes at x, init / final at y with the subtable MLE and the identity polynomial)."""
    bn = orclib.bn254()
    p = orclib.params(1024, 1)
    lasso_in, _, _ = orclib.circuit_eval(p, orclib.fixture_inputs(1024, 1, 27))
    P = orclib.lasso_polys(p, lasso_in)
    assert P["nu"] == 14 and P["A"] == 6 and P["rows"] == 10240
    chal = bn.challenges(bn.lasso_challenge_count(P["nu"]), orclib.keccak256)
    sec = {}
    proof, r, claimed = bn.lasso_prove(P, chal, sec)
    A, R = P["A"], bn.R
    gamma, tau, x, y = sec["gamma"], sec["tau"], sec["x"], sec["y"]
    h = lambda a, v, t: (a + v * gamma + t * gamma * gamma - tau) % R
    id_y = sum((1 << i) * yi for i, yi in enumerate(y)) % R
    pos = sec["openings"]
    i = 0
    for c in sorted(set(P["mem_dim"])):
        mems = [m for m in range(A) if P["mem_dim"][m] == c]
        dim_x, rts_x, fct_y = proof[pos], proof[pos + 1], proof[pos + 2]
        e_xs = proof[pos + 3:pos + 3 + len(mems)]
        pos += 3 + len(mems)
        for q, m in enumerate(mems):
            assert sec["order"][i] == (m, c)
            assert sec["gp1_claims"][i] == h(dim_x, e_xs[q], rts_x)
            assert sec["gp1_claims"][A + i] == h(dim_x, e_xs[q], (rts_x + 1) % R)
            t_y = bn.mle_eval([a if a < P["mem_cutoff"][m] else 0 for a in range(65536)], y)
            assert sec["gp2_claims"][i] == h(id_y, t_y, 0)
            assert sec["gp2_claims"][A + i] == h(id_y, t_y, fct_y)
            i += 1
    assert pos == len(proof) and proof[0] == claimed and len(r) == 14


def test_bn254_lasso_oracle_verifier_accepts_and_rejects():
    bn = orclib.bn254()
    p = orclib.params(1024, 1)
    lasso_in, _, _ = orclib.circuit_eval(p, orclib.fixture_inputs(1024, 1, 27))
    P = orclib.lasso_polys(p, lasso_in)
    chal = bn.challenges(bn.lasso_challenge_count(P["nu"]), orclib.keccak256)
    proof, r, claimed = bn.lasso_prove(P, chal)
    assert bn.lasso_verify(proof, P["nu"], P["mem_dim"], P["mem_cutoff"], chal) == (r, claimed)
    for at in (1, len(proof) // 2, len(proof) - 1):   # a collation coefficient, a grand-product element, the last opening
        bad = list(proof)
        bad[at] = (bad[at] + 1) % bn.R
        with pytest.raises(ValueError):
            bn.lasso_verify(bad, P["nu"], P["mem_dim"], P["mem_cutoff"], chal)


# ---- BN254 whole-proof oracle (oracle/bn254_gkr.py) on the reference's own bn254 fixture ---------------------------------------
def test_bn254_gkr_oracle_on_the_reference_fixture():
    """bfv-gkr/src/data/bn254/sk_enc_1024_1x27_65537.json: layout + circuit relation (sum == ct0is), prove -> verify round trip,
    tamper rejection, and the proof has exactly as many elements as the Goldilocks oracle's proof of the same parameter set
    (same protocol with E = F)."""
    import json
    G, bn = orclib.bn254_gkr(), orclib.bn254()
    n, k = 1024, 1
    c = orclib.constants(n, k)
    w = json.load(open(os.path.join(orclib.GOLDEN, "bn254_sk_enc_1024_1x27_65537.json")))
    inputs, ct0is = G.layout_inputs(n, k, w)
    Cc, lasso_in, lasso_id, sum_id = G.build_circuit(c)
    vals = G.circuit_evaluate(Cc, inputs)
    assert vals[sum_id] == ct0is
    assert max(vals[lasso_in]) < 1 << 28          # range-shifted lookups are small non-negative integers
    # the Goldilocks fixture of this parameter set (another sample) is an integer witness too: lifted into Fr it satisfies the relation
    gl = orclib.fixture_inputs(n, k, 27).d
    lifted = [[G.lift_signed(v) for v in gl[f]] for f in ("s", "e", "k1", "ais", "r1is", "r2is")]
    assert G.circuit_evaluate(Cc, lifted)[sum_id] == [G.lift_signed(v) for v in gl["ct0is"]]
    p = orclib.params(n, k)
    prove_fn, verify_fn = orclib.bn254_lasso_fns(p)
    chal = bn.challenges(3000, orclib.keccak256)
    proof, _ = G.prove(c, inputs, ct0is, chal, prove_fn)
    gl_proof, _ = orclib.prove(p, orclib.fixture_inputs(n, k, 27))
    assert len(proof) == len(gl_proof) // 16
    gold = json.load(open(os.path.join(orclib.GOLDEN, "oracle_proof_digests.json")))["bn254_1024_1"]   # regression pin of the oracle itself
    wire = b"".join(int(v).to_bytes(32, "big") for v in proof)
    assert len(wire) == gold["bytes"] and hashlib.sha256(wire).hexdigest() == gold["sha256"]
    assert G.verify(c, inputs, ct0is, proof, chal, verify_fn)
    for at in (0, len(proof) // 2, len(proof) - 1):
        bad = list(proof)
        bad[at] = (bad[at] + 1) % G.R
        with pytest.raises(ValueError):
            G.verify(c, inputs, ct0is, bad, chal, verify_fn)
    wrong = list(ct0is)
    wrong[3] = (wrong[3] + 1) % G.R
    with pytest.raises(ValueError):
        G.verify(c, inputs, wrong, proof, chal, verify_fn)


def test_bn254_gkr_oracle_fft_and_circuit_pieces():
    G = orclib.bn254_gkr()
    bn = orclib.bn254()
    rng = random.Random(5)
    a = [rng.randrange(G.R) for _ in range(16)]
    assert G.ntt(a) == bn.ntt(a) and G.ntt(G.ntt(a), True) == a      # radix-2 vs the O(n^2) definition
    r = [rng.randrange(G.R) for _ in range(4)]
    # F(r, x) is the multilinear extension in the output index of the DFT matrix: sum_x F(r,x) a[x] = MLE(ntt(a))(r)
    F = G.fft_table(r, 4, False)
    assert sum(f * x for f, x in zip(F, a)) % G.R == G.mle_eval(G.ntt(a), r)
    Fi = G.fft_table(r, 4, True)
    assert sum(f * x for f, x in zip(Fi, a)) % G.R == G.mle_eval(G.ntt(a, True), r)
    assert G.lift_signed(5) == 5 and G.lift_signed(G.GL_P - 7) == G.R - 7


# ---- the C++ oracle over bn256::Fr (oracle/fr.hpp) and the protocol modes (transcript.hpp ProtocolMode) -----------------------
def test_fr_field_against_python_integers():
    R = orclib.R_BN
    rng = random.Random(11)
    a = [rng.randrange(R) for _ in range(200)] + [0, 1, R - 1, R - 1, (1 << 253), (1 << 64) - 1]
    b = [rng.randrange(R) for _ in range(200)] + [R - 1, R - 1, R - 1, 1, (1 << 253) + 5, (1 << 64) - 1]
    assert orclib.f_binop("bn254", 0, a, b) == [(x + y) % R for x, y in zip(a, b)]
    assert orclib.f_binop("bn254", 1, a, b) == [(x - y) % R for x, y in zip(a, b)]
    assert orclib.f_binop("bn254", 2, a, b) == [(x * y) % R for x, y in zip(a, b)]
    nz = [x for x in a if x][:20]
    assert orclib.f_binop("bn254", 3, nz, nz) == [pow(x, -1, R) for x in nz]
    # halo2curves bn256::Fr ROOT_OF_UNITY (S = 28): 7^((r-1)/2^28), of exact order 2^28
    w = orclib.root_of_unity_f("bn254", 28)
    assert w == pow(7, (R - 1) >> 28, R) == 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
    assert pow(w, 1 << 27, R) == R - 1
    assert orclib.root_of_unity_f("bn254", 11) == pow(w, 1 << 17, R)


def test_fr_wire_format_and_challenge_chain():
    R = orclib.R_BN
    vals = [0, 1, R - 1, 0x0102030405060708090a0b0c0d0e0f101112131415161718191a1b1c1d1e1f20 % R]
    by, back = orclib.wire_roundtrip("bn254", vals)
    assert back == vals and by == b"".join(v.to_bytes(32, "big") for v in vals)   # transcript.rs:183-189: repr reversed to BE
    by, back = orclib.wire_roundtrip("goldilocks", [5, P - 1])
    assert by == (5).to_bytes(8, "big") + (P - 1).to_bytes(8, "big") and back == [5, P - 1]
    # squeeze_challenge over Fr (transcript.rs:198-203): c_j = LE(Keccak^j("")) mod r; first value pinned in SURVEY 8(c)(5)
    ch = orclib.challenge_chain("bn254", 8)
    h, exp = b"", []
    for _ in range(8):
        h = orclib.keccak256(h)
        exp.append(int.from_bytes(h, "little") % R)
    assert ch == exp
    assert ch[0] == 7173236656320612194178997223602979818891828541827642103715116037219761443523
    assert orclib.challenge_chain("bn254", 8) == orclib.bn254().challenges(8, orclib.keccak256)


@pytest.mark.parametrize("n,k,bits", BN_FIX)
def test_fr_oracle_on_every_reference_bn254_fixture(n, k, bits):
    """Every bn254 witness the reference holds: get_inputs layout, the circuit relation sum == ct0is and the range bounds on the
    integer witness, orcbn_prove -> orcbn_verify round trip, tamper and wrong-public-input rejection, the regression digest, and
    the element count of the Goldilocks proof of the same parameter set (same protocol with E = F)."""
    p = orclib.params(n, k)
    inp = orclib.bn254_fixture_inputs(n, k, bits)
    lasso_in, sum_out, info = orclib.circuit_eval(p, inp)
    assert (sum_out == inp.d["ct0is"]).all()
    chunks = max(1, k // 2)
    SZ = 1 << p.L
    c = p.c
    bounds = c["r1_bounds"][:k] + [c["r2_bounds"][0]] * chunks + [c["s_bound"], c["e_bound"], c["k1_bound"]]
    for i, b in enumerate(bounds):
        assert int(lasso_in[i * SZ:(i + 1) * SZ].max()) <= 2 * b
    proof, _ = orclib.prove_f("bn254", p, inp, threads=8)
    gold = json.load(open(os.path.join(orclib.GOLDEN, "oracle_proof_digests.json")))[f"bn254_{n}_{k}"]
    assert len(proof) == gold["bytes"] and hashlib.sha256(proof).hexdigest() == gold["sha256"]
    gl_proof, _ = orclib.prove(p, orclib.fixture_inputs(n, k, bits), threads=8)
    assert len(proof) // 32 == len(gl_proof) // 16
    ok, err = orclib.verify_f("bn254", p, inp, proof, threads=8)
    assert ok, err
    bad = bytearray(proof)
    bad[len(bad) // 2 + 31] ^= 1
    assert not orclib.verify_f("bn254", p, inp, bytes(bad), threads=8)[0]
    d2 = dict(inp.d)
    d2["ct0is"] = inp.d["ct0is"].copy()
    d2["ct0is"][5] ^= np.uint64(1)
    assert not orclib.verify_f("bn254", p, orclib.Inputs(d2), proof, threads=8)[0]


def test_fr_oracle_reproduces_the_python_oracle_on_the_reference_bn254_fixture():
    """orcbn_prove (C++ over Fr) on the reference's own bn254 witness: byte-identical to the Python-integer oracle's proof
    (oracle/bn254_gkr.py, pinned by digest) - two independent restatements of the same protocol agree on all 2121 elements."""
    import hashlib
    p = orclib.params(1024, 1)
    inp = orclib.bn254_fixture_inputs()
    proof, _ = orclib.prove_f("bn254", p, inp, threads=4)
    gold = json.load(open(os.path.join(orclib.GOLDEN, "oracle_proof_digests.json")))["bn254_1024_1"]
    assert len(proof) == gold["bytes"] and hashlib.sha256(proof).hexdigest() == gold["sha256"]
    ok, err = orclib.verify_f("bn254", p, inp, proof, threads=4)
    assert ok, err
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    assert not orclib.verify_f("bn254", p, inp, bytes(bad), threads=4)[0]
    # the same integer witness in its Goldilocks fixture proves to the same Fr transcript (the witness is field-independent)
    proof_gl_fixture, _ = orclib.prove_f("bn254", p, orclib.fixture_inputs(1024, 1, 27), threads=4)
    assert len(proof_gl_fixture) == len(proof)


@pytest.mark.parametrize("field", ["goldilocks", "bn254"])
def test_protocol_modes_round_trip_and_are_distinguished(field):
    """f-4 modes of the oracle: absorbing transcript (bit 0) and extension-field memory checking (bit 1). Each mode's proof
    verifies under that mode only; a flipped byte is rejected; mode 0 is the reference as it is."""
    p = orclib.params(1024, 1)
    inp = orclib.fixture_inputs(1024, 1, 27) if field == "goldilocks" else orclib.bn254_fixture_inputs()
    base, _ = orclib.prove_f(field, p, inp, threads=4)
    assert base == (orclib.prove(p, inp, threads=4)[0] if field == "goldilocks" else base)
    for mode in (orclib.MODE_ABSORB, orclib.MODE_EXT_MEMCHECK, orclib.MODE_ABSORB | orclib.MODE_EXT_MEMCHECK):
        proof, _ = orclib.prove_f(field, p, inp, threads=4, mode=mode)
        assert len(proof) == len(base)
        same_as_base = field == "bn254" and mode == orclib.MODE_EXT_MEMCHECK   # E = F over Fr: gamma, tau are never truncated
        assert (proof == base) == same_as_base
        ok, err = orclib.verify_f(field, p, inp, proof, threads=4, mode=mode)
        assert ok, err
        assert orclib.verify_f(field, p, inp, proof, threads=4, mode=0)[0] == same_as_base
        bad = bytearray(proof)
        bad[len(bad) // 3] ^= 2
        assert not orclib.verify_f(field, p, inp, bytes(bad), threads=4, mode=mode)[0]


def test_absorbing_transcript_depends_on_every_message():
    """With ProtocolMode::absorb the challenge after a write is Keccak(previous hash || to_repr(message)) - the in-tree
    plonkish-trait writer's rule (transcript.rs:205-208, 224-233) - so the Lasso node's proof changes with the point it enters at
    and with its input; without it the challenges are the fixed chain."""
    p = orclib.params(1024, 1)
    lasso_in, _, _ = orclib.circuit_eval(p, orclib.fixture_inputs(1024, 1, 27))
    a, _ = orclib.lasso_prove_f("goldilocks", p, lasso_in, threads=4, mode=orclib.MODE_ABSORB)
    ok, err = orclib.lasso_verify_f("goldilocks", p, a, mode=orclib.MODE_ABSORB)
    assert ok, err
    assert not orclib.lasso_verify_f("goldilocks", p, a, mode=0)[0]
    b, _ = orclib.lasso_prove_f("goldilocks", p, lasso_in, threads=4, mode=0)
    assert a != b and b == orclib.lasso_prove(p, lasso_in, threads=4)[0]


def test_unused_memories_of_a_chunk_have_identical_hash_rows_inside_a_lookup_segment():
    """The structural fact behind the next optimisation of the read / write grand product: the hash of memory m
    at row j is h(dim_c[j], E_m[j], read_ts_c[j]) with chunk value and read counter taken per CHUNK position c = mem_dim[m]
    (lasso.rs:317-319) and E_m[j] = 0 on every row whose lookup does not use m - so inside a lookup segment all unused memories of a
    chunk have the same (address, value, timestamp) triples, and a lookup uses at most one memory per chunk position. Checked on the
    reference's own witnesses (k = 1, 2, 4)."""
    for n, k, bits in ((1024, 1, 27), (4096, 2, 55), (8192, 4, 55)):
        p = orclib.params(n, k)
        w = json.load(open(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json")))
        lasso_in, _, _ = orclib.circuit_eval(p, orclib.Inputs(orclib.layout_inputs(n, k, w)))
        P = orclib.lasso_polys(p, lasso_in)
        A, rows, md, lm = P["A"], P["rows"], P["mem_dim"], P["lookup_mems"]
        rl = np.array(P["row_lookup"])[:rows]
        ep = np.array(P["e_polys"], dtype=np.uint64)[:, :rows]
        for l, mems in enumerate(lm):
            assert len(set(md[m] for m in mems)) == len(mems) <= 4, (l, mems)   # one memory per chunk position
        for m in range(A):
            unused = ~np.isin(rl, [l for l in range(len(lm)) if m in lm[l]])
            assert not ep[m][unused].any(), (n, m)                               # E_m vanishes where m is not looked up
        # rows come in whole segments per lookup
        change = np.flatnonzero(np.diff(rl)) + 1
        assert all(int(c) % n == 0 for c in change), change[:8]   # segments of 2^n_log2 = n rows
        # distinct (chunk, used-memory-or-class) tables per segment: at most 2 per chunk position
        per_chunk = [sum(1 for m in range(A) if md[m] == c) for c in range(4)]
        print(f"n={n} k={k}: alpha={A}, memories per chunk position {per_chunk}, {len(change) + 1} segments: "
              f"<= {sum(min(2, c) for c in per_chunk)} distinct read tables per segment instead of {A}")


def test_joint_classes_of_the_product_tree_layers_share_their_rows():
    """What the slot form of grand product #1 rests on (prover.hip: lasso_node / GpSlots), checked on the reference's
    own witnesses with the class rule restated here: layer d of the product tree multiplies 2^(d+1) row segments that lie
    (segments >> (d+1)) apart; two memories whose classes agree on all of them (class of a memory in a segment = itself where the
    segment's lookup uses it, else its chunk position) have identical level-d product rows inside that segment group - for the read
    hashes and for the write hashes (read + gamma^2). gamma, tau: arbitrary field elements."""
    P = (1 << 64) - (1 << 32) + 1
    gamma, tau = 0x1234567887654321 % P, 0x0FEDCBA998765432 % P
    for n, k, bits in ((1024, 1, 27), (4096, 2, 55)):
        p = orclib.params(n, k)
        w = json.load(open(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json")))
        lasso_in, _, _ = orclib.circuit_eval(p, orclib.Inputs(orclib.layout_inputs(n, k, w)))
        L = orclib.lasso_polys(p, lasso_in)
        A, nu, rows, md, lm = L["A"], L["nu"], L["rows"], L["mem_dim"], L["lookup_mems"]
        N = 1 << nu
        nseg = N // n
        seg_lookup = [L["row_lookup"][s * n] if s * n < rows else None for s in range(nseg)]
        def cls(m, s):
            return 1000 + m if seg_lookup[s] is not None and m in lm[seg_lookup[s]] else md[m]
        dims = [np.array(d, dtype=object) for d in L["dims"]]
        ts = [np.array(t, dtype=object) for t in L["read_cts"]]   # (per memory; equal for the memories of a chunk position)
        for add in (0, gamma * gamma % P):   # read rows, write rows
            lev = [(dims[md[m]] + np.array(L["e_polys"][m], dtype=object) * gamma + ts[m] * (gamma * gamma % P) - tau + add) % P for m in range(A)]
            for d in range(3):
                half = len(lev[0]) // 2
                lev = [(r[:half] * r[half:]) % P for r in lev]          # level d + 1: v_l * v_r on the MSB split (prover.rs:308-313)
                ng = nseg >> (d + 1)
                if ng < 1: break
                found = 0
                for g in range(ng):
                    groups = {}
                    for m in range(A):
                        groups.setdefault(tuple(cls(m, g + q * ng) for q in range(2 << d)), []).append(m)
                    for members in groups.values():
                        for m in members[1:]:
                            assert (lev[m][g * n:(g + 1) * n] == lev[members[0]][g * n:(g + 1) * n]).all(), (n, d, g, members)
                    found = max(found, len(groups))
                assert found < A or d > 0, (n, d, found)   # (the top layer always has something to share)
