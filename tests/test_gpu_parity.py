"""GPU parity tests (-m gpu): every check calls the HIP path through the C ABI and compares with the
CPU oracle bit for bit (integer work: exact equality)."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import orclib
from orclib import P, ptr
from hglib import hg

pytestmark = pytest.mark.gpu

OL = orclib.lib()


@pytest.fixture(scope="module")
def ctx():
    c = hg.Context(0)
    yield c
    c.close()


def _first_diff(a, b, nbytes=32):
    if len(a) != len(b):
        return "lengths %d / %d bytes" % (len(a), len(b))
    for i in range(0, len(a), nbytes):
        if a[i:i + nbytes] != b[i:i + nbytes]:
            return "first differing element %d of %d" % (i // nbytes, len(a) // nbytes)
    return "identical"


def rand_f(rng, n):
    edge = [0, 1, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000]
    v = [rng.randrange(P) for _ in range(n)]
    for i, e in enumerate(edge[:n]):
        v[rng.randrange(n)] = e
    return np.array(v, dtype=np.uint64)


def oracle_sumcheck(kind, tables, is_base, pw, claim, chain_skip=0, threads=4):
    ntab = len(tables)
    nv = int(np.log2(tables[0].size if is_base[0] else tables[0].size // 2))
    d = 3 if kind == 1 else 2
    tabs = [np.ascontiguousarray(t, dtype=np.uint64) for t in tables]
    ptrs = (orclib.u64p * ntab)(*[ptr(t) for t in tabs])
    flags = (C.c_int * ntab)(*[int(b) for b in is_base])
    pw = np.ascontiguousarray(pw, dtype=np.uint64).reshape(-1)
    claim = np.ascontiguousarray(claim, dtype=np.uint64)
    msgs = np.zeros(nv * (d + 1) * 2, dtype=np.uint64)
    point = np.zeros(nv * 2, dtype=np.uint64)
    evals = np.zeros(ntab * 2, dtype=np.uint64)
    sums = np.zeros(nv * d * 2, dtype=np.uint64)
    OL.orc_sumcheck(kind, C.c_size_t(nv), C.c_size_t(ntab), ptrs, flags, ptr(pw), C.c_size_t(pw.size // 2), ptr(claim),
                    C.c_size_t(chain_skip), threads, ptr(msgs), ptr(point), ptr(evals), ptr(sums))
    return msgs, point, evals, sums


@pytest.mark.parametrize("kind,ntab,nv,base", [
    (0, 6, 10, True), (0, 25, 12, True), (0, 3, 1, True), (0, 9, 7, False),
    (1, 12, 9, True), (1, 100, 11, True), (1, 4, 1, True), (1, 2, 14, True), (1, 8, 6, False),
    (2, 2, 11, None), (2, 10, 9, None), (2, 54, 8, None), (2, 2, 1, None),
    # long tables: fused two-round grand-product launches (half >= 2^13), mixed-size launches, chunked last rounds
    (1, 6, 15, True), (1, 4, 16, True), (1, 4, 15, False), (0, 5, 15, True), (2, 4, 14, None),
])
def test_sumcheck_kernels_bit_exact(ctx, kind, ntab, nv, base):
    rng = random.Random(kind * 1000 + ntab * 10 + nv)
    N = 1 << nv
    if kind == 2:
        is_base = [i % 2 == 0 for i in range(ntab)]
    else:
        is_base = [base] * ntab
    tables = [rand_f(rng, N if b else 2 * N) for b in is_base]
    npw = ntab if kind == 0 else ntab // 2
    if kind == 0:
        pw = np.array([[pow(65536, i, P), 0] for i in range(npw)], dtype=np.uint64)
    else:
        if kind == 1:  # grand product: pw[i] = gamma^i in GoldilocksExt2 (X^2 = 7), pw[0] = 1
            g = (rng.randrange(P), rng.randrange(P))
            cur, pws = (1, 0), []
            for _ in range(npw):
                pws.append(cur)
                cur = ((cur[0] * g[0] + 7 * cur[1] * g[1]) % P, (cur[0] * g[1] + cur[1] * g[0]) % P)
            pw = np.array(pws, dtype=np.uint64)
        else:
            pw = np.zeros((0, 2), dtype=np.uint64)
    claim = np.array([rng.randrange(P), rng.randrange(P)], dtype=np.uint64)
    skip = rng.randrange(50)
    got = ctx.sumcheck(kind, tables, is_base, pw, claim, skip)
    exp = oracle_sumcheck(kind, tables, is_base, pw, claim, skip)
    for name, g_, e_ in zip(("msgs", "point", "evals", "sums"), got, exp):
        assert (g_ == e_).all(), name


def test_mle_eval_and_ntt(ctx):
    rng = random.Random(7)
    for nv in (0, 1, 5, 13, 16):
        tab = rand_f(rng, 1 << nv)
        pt = rand_f(rng, max(2 * nv, 2))[:2 * nv]
        exp = np.zeros(2, dtype=np.uint64)
        OL.orc_mle_eval_f(ptr(tab), C.c_size_t(nv), ptr(np.ascontiguousarray(pt)), ptr(exp))
        got = ctx.mle_eval(tab, pt)
        assert (got == exp).all(), nv
    for log2n in (1, 4, 11, 13, 16):
        for inv in (False, True):
            batch = 3
            x = rand_f(rng, batch << log2n)
            got = ctx.ntt(x, log2n, inv, batch)
            for b in range(batch):
                exp = np.zeros(1 << log2n, dtype=np.uint64)
                OL.orc_ntt(ptr(np.ascontiguousarray(x[b << log2n:(b + 1) << log2n])), C.c_size_t(log2n), int(inv), ptr(exp))
                assert (got[b << log2n:(b + 1) << log2n] == exp).all(), (log2n, inv, b)


# every Goldilocks witness the reference holds (bfv-gkr/src/data/goldilocks/, copied as data into tests/golden/); 8192 is its only
# witness with k = 4 (two r2is chunks, sk_encryption_circuit.rs:149-161)
FIX = [(1024, 1, 27), (2048, 1, 52), (4096, 2, 55), (8192, 4, 55)]
BN_FIX = [(1024, 1, 27), (2048, 1, 52), (4096, 2, 55)]   # bfv-gkr/src/data/bn254/


@pytest.mark.parametrize("n,k", [(1024, 1), (4096, 2), (32768, 16)])
def test_device_witness_generation_matches_host(ctx, n, k):
    """Circuit::evaluate on the device (batched NTTs + gate kernels) vs the host evaluation and the oracle."""
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    vals = hg.witness_gen(ctx, pk, w)
    lasso_in, sum_out = pk.circuit_eval(w)  # host
    assert (vals.node(ctx, pk.lasso_in_id) == lasso_in).all()
    assert (vals.node(ctx, pk.sum_id) == sum_out).all()
    assert (sum_out == w.arrays()["ct0is"]).all()
    o_lasso, o_sum, _ = orclib.circuit_eval(orclib.params(n, k), orclib.Inputs(w.arrays()))  # the oracle, at every size
    assert (o_lasso == lasso_in).all() and (o_sum == sum_out).all()
    vals.free()
    pk.free()


@pytest.mark.parametrize("n,k,bits", FIX)
def test_lasso_node_transcript_bit_exact(ctx, n, k, bits):
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = bfv.get_inputs(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    lasso_in, _ = pk.circuit_eval(w)
    proof, claim = hg.LassoNode(pk).prove_claim_reduction(ctx, lasso_in)
    p = orclib.params(n, k)
    ref, ref_claim = orclib.lasso_prove(p, lasso_in, threads=4)
    assert (claim == ref_claim).all()
    assert proof == ref
    ok, err = orclib.lasso_verify(p, proof)
    assert ok, err
    # adversarial table: random field elements (out of range for every lookup) still give the same transcript
    rng = random.Random(n)
    junk = rand_f(rng, lasso_in.size)
    proof2, claim2 = hg.LassoNode(pk).prove_claim_reduction(ctx, junk)
    ref2, ref_claim2 = orclib.lasso_prove(p, junk, threads=4)
    assert proof2 == ref2 and (claim2 == ref_claim2).all()
    # all-zero table (every row hits address 0: worst-case counter collisions)
    zeros = np.zeros_like(lasso_in)
    proof3, _ = hg.LassoNode(pk).prove_claim_reduction(ctx, zeros)
    ref3, _ = orclib.lasso_prove(p, zeros, threads=4)
    assert proof3 == ref3
    pk.free()


@pytest.mark.parametrize("n,k,bits", FIX)
def test_full_prove_bit_exact_on_reference_fixtures(ctx, n, k, bits):
    import hashlib, json
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = bfv.get_inputs(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    proof, tm = bfv.prove(ctx, pk, w)
    p = orclib.params(n, k)
    inp = orclib.Inputs(w.arrays())
    ref, _ = orclib.prove(p, inp, threads=4)
    assert proof == ref
    gold = json.load(open(os.path.join(orclib.GOLDEN, "oracle_proof_digests.json")))[f"{n}_{k}"]
    assert hashlib.sha256(proof).hexdigest() == gold["sha256"]
    ok, err = orclib.verify(p, inp, proof)
    assert ok, err
    ok, err = hg.verify(pk, w, proof)  # product-side verifier (BfvEncrypt::verify)
    assert ok, err
    ok, err = hg.verify_device(ctx, pk, w, proof)  # the same decisions with the table-sized checks on the device
    assert ok, err
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 4
    assert not hg.verify(pk, w, bytes(bad))[0] and not hg.verify_device(ctx, pk, w, bytes(bad))[0]
    proof2, _ = bfv.prove(ctx, pk, w)  # determinism + arena reuse
    assert proof2 == proof
    pk.free()


@pytest.mark.parametrize("n,k", [(2048, 1), (8192, 4), (16384, 8)])
def test_full_prove_bit_exact_synthetic(ctx, n, k):
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    proof, _ = bfv.prove(ctx, pk, w)
    p = orclib.params(n, k)
    inp = orclib.Inputs(w.arrays())
    ref, _ = orclib.prove(p, inp, threads=8)
    assert proof == ref
    pk.free()


def test_headline_config_properties(ctx):
    """n=32768 k=16 (BASELINE configs[2]): bit-exact against the oracle prover (about 6 s on the GPU box's host cores),
    plus size-independent properties: both verifier implementations accept the HIP proof, the proof is
    deterministic, a tampered proof is rejected."""
    n, k = 32768, 16
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    proof, tm = bfv.prove(ctx, pk, w)
    proof2, tm2 = bfv.prove(ctx, pk, w)
    assert proof == proof2
    p = orclib.params(n, k)
    inp = orclib.Inputs(w.arrays())
    threads = min(64, os.cpu_count() or 8)
    ok, err = orclib.verify(p, inp, proof, threads=threads)
    assert ok, err
    ok, err = hg.verify(pk, w, proof)
    assert ok, err
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 4
    ok, _ = orclib.verify(p, inp, bytes(bad), threads=threads)
    assert not ok
    ref, _ = orclib.prove(p, inp, threads=threads)
    assert proof == ref  # the whole 149,488-byte transcript, bit for bit
    print("c3 timings", tm2)
    pk.free()


@pytest.mark.parametrize("n,k,world", [(1024, 1, 2), (4096, 2, 3), (8192, 4, 8), (32768, 16, 8), (32768, 16, 4), (32768, 16, 2)])
def test_sharded_single_proof_reassembles_bit_exact(ctx, n, k, world):
    """Single-proof sharding (hg_prove_shard_*): run every virtual rank of a `world`-GPU job on this one GPU, sum the
    partial result buffers the way the all-gather + hg_prove_shard_combine does (lane-wise sum mod p: grand product
    #1 is split by memory, its round sums are partial sums), replay -> must give exactly the unsharded proof."""
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    vals = hg.witness_gen(ctx, pk, w)
    out = hg.ProofBuffer()
    ref = hg.prove_resident(ctx, pk, vals, out).bytes()
    parts = []
    for r in range(world):
        parts.append(hg.prove_shard_begin(ctx, pk, vals, r, world).copy())
    # every lane outside the shared round-sum slots has at most one contributor; the combine is a lane-wise sum mod p
    hg.prove_shard_combine(ctx, np.stack(parts), world)
    got = hg.prove_shard_finish(ctx, out).bytes()
    assert got == ref
    vals.free()
    pk.free()


@pytest.mark.parametrize("n,k,world", [(4096, 2, 2), (32768, 16, 8)])
def test_sharded_proof_from_per_rank_resident_tables(ctx, n, k, world):
    """BASELINE config 4 without replicating the witness: every virtual rank gets its own values object from hg_witness_gen_shard -
    only the node tables its share reads (the others are not resident: a kernel touching them would fault) - runs its share, the
    partial buffers are combined and replayed: the proof must equal the oracle's. A second witness goes through the same objects
    (hg_witness_gen_into) and, from the third prove on, through each rank's launch graph."""
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    p = orclib.params(n, k)
    threads = min(16, os.cpu_count() or 8)
    ws = [hg.Witness.synthetic(bfv.params, 0x91 + i) for i in range(2)]
    refs = [orclib.prove(p, orclib.Inputs(w.arrays()), threads=threads)[0] for w in ws]
    vals = [hg.witness_gen_shard(ctx, pk, ws[0], r, world) for r in range(world)]
    infos = [v.info() for v in vals]
    full = infos[0]["full_bytes"]
    assert all(i["resident_bytes"] < full for i in infos) and all(i["resident_tables"] < i["tables"] for i in infos)
    if world == 8:
        assert max(i["resident_bytes"] for i in infos) < full // 2, infos
    print("n=%d world=%d: resident MB per rank %s of %.1f MB" % (n, world, ["%.1f" % (i["resident_bytes"] / 1e6) for i in infos], full / 1e6))
    out = hg.ProofBuffer()
    with pytest.raises(hg.HgError, match="of rank"):
        hg.prove_shard_begin(ctx, pk, vals[0], 1, world)                   # another rank's share cannot run on these tables
    with pytest.raises(hg.HgError, match="of rank"):
        hg.prove_resident(ctx, pk, vals[0], out)
    for it in range(5):                                                    # walk, walk, capture / replay; the second witness from it == 2 on
        j = 0 if it < 2 else 1
        if it == 2:
            for r in range(world):
                hg.witness_gen_into(ctx, pk, ws[1], vals[r])
        parts = [hg.prove_shard_begin(ctx, pk, vals[r], r, world).copy() for r in range(world)]
        hg.prove_shard_combine(ctx, np.stack(parts), world)
        assert hg.prove_shard_finish(ctx, out).bytes() == refs[j], (it, _first_diff(out.bytes(), refs[j], 16))
    for v in vals:
        v.free()
    pk.free()


def test_a_ranks_share_replays_from_its_launch_graph(ctx):
    """The share of one rank of an 8-rank proof, five times in a row: two walks, the capture, two graph replays - the partial
    result buffer must be the same every time, and the last one must still reassemble to the unsharded proof."""
    n, k, world, r = 8192, 4, 8, 3
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    vals = hg.witness_gen(ctx, pk, w)
    out = hg.ProofBuffer()
    ref = hg.prove_resident(ctx, pk, vals, out).bytes()
    others = {q: hg.prove_shard_begin(ctx, pk, vals, q, world).copy() for q in range(world) if q != r}
    mine = [hg.prove_shard_begin(ctx, pk, vals, r, world).copy() for _ in range(5)]
    for i in range(1, 5):
        assert np.array_equal(mine[i], mine[0]), i
    parts = [others[q] if q != r else mine[4] for q in range(world)]
    hg.prove_shard_combine(ctx, np.stack(parts), world)
    assert hg.prove_shard_finish(ctx, out).bytes() == ref
    vals.free()
    pk.free()


@pytest.mark.parametrize("n,k,bits", FIX)
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_protocol_modes_bit_exact_against_the_oracle(ctx, n, k, bits, mode):
    """absorbing transcript (bit 0) and extension-field memory checking (bit 1). The round-by-round HIP prover
    (hg_prove_mode) must produce the oracle's transcript in the same mode byte for byte - every challenge now depends on every
    earlier message, so one wrong round sum anywhere changes everything after it - and both verifiers accept it in that mode only."""
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = bfv.get_inputs(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    proof, tm = bfv.prove(ctx, pk, w, mode=mode)
    p = orclib.params(n, k)
    inp = orclib.Inputs(w.arrays())
    ref, _ = orclib.prove_f("goldilocks", p, inp, threads=8, mode=mode)
    assert proof == ref, _first_diff(proof, ref, 16)
    base, _ = bfv.prove(ctx, pk, w)
    assert proof != base and len(proof) == len(base)
    assert orclib.verify_f("goldilocks", p, inp, proof, threads=8, mode=mode)[0]
    assert hg.verify(pk, w, proof, mode=mode) == (True, "")
    assert not hg.verify(pk, w, proof, mode=0)[0]
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    assert not hg.verify(pk, w, bytes(bad), mode=mode)[0]
    print("mode %d n=%d: %.1f ms, %d synchronisations" % (mode, n, tm["prove_ms"], int(tm["sync_ms"])))
    pk.free()


_SHARDED_SEQ_CODE = r"""
import sys, threading
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import __graft_entry__ as entry
import orclib
hg = entry.load_package()
n, k, world, mode = %(n)d, %(k)d, %(world)d, %(mode)d
bfv = hg.BfvEncrypt.new(n, k)
w = hg.Witness.synthetic(bfv.params, 400 + n + mode)
ref, _ = orclib.prove_f("goldilocks", orclib.params(n, k), orclib.Inputs(w.arrays()), threads=8, mode=mode)
group = hg.Group.local(world)
results, errs = [None] * world, []
def run(r):
    try:
        c = hg.Context(0)
        pk = bfv.setup(c)
        v = hg.witness_gen(c, pk, w)
        out = hg.ProofBuffer()
        hg.prove_resident_mode_sharded(c, pk, v, out, mode, r, group)
        results[r] = (out.bytes(), out.timings())
        v.free(); pk.free()
    except Exception as e:   # (a failed rank breaks the group: the others return an error instead of waiting for ever)
        errs.append((r, repr(e)))
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
for t in ts: t.start()
for t in ts: t.join(300)
assert not errs, errs
for r in range(world):
    assert results[r] is not None and results[r][0] == ref, r
n_red = [int(results[r][1]["replay_ms"]) for r in range(world)]
trips = [int(results[r][1]["enqueue_ms"]) for r in range(world)]
assert len(set(n_red)) == 1 and n_red[0] > 100, n_red   # the same rounds on every rank
assert all(nr <= t for nr, t in zip(n_red, trips)), (n_red, trips)   # every all-reduce rides on a mailbox round trip
# world = 1 through the same entry point is the plain prover; mode 0 shards per proof, not per round
c = hg.Context(0); pk = bfv.setup(c); v = hg.witness_gen(c, pk, w); out = hg.ProofBuffer()
assert hg.prove_resident_mode_sharded(c, pk, v, out, mode, 0, hg.Group.local(1)).bytes() == ref
assert int(out.timings()["replay_ms"]) == 0
try:
    hg.prove_resident_mode_sharded(c, pk, v, out, 0, 0, hg.Group.local(2))
    raise SystemExit("mode 0 accepted")
except hg.HgError:
    pass
print("SHARDED SEQ OK mode %%d n=%%d world=%%d: %%d all-reduces, %%d mailbox round trips per rank" %% (mode, n, world, n_red[0], trips[0]))
"""


@pytest.mark.parametrize("n,k,world,mode", [(1024, 1, 2, 3), (4096, 2, 3, 3), (4096, 2, 4, 1), (4096, 2, 2, 2)])
def test_sharded_round_by_round_prover_one_allreduce_per_round(n, k, world, mode):
    """SURVEY 8(e), the form that stays available for an absorbing transcript: `world` ranks (threads of one process, one context
    each, all on device 0: hg_group_local) run hg_prove_resident_mode_sharded; inside every round kernel a rank evaluates the sums of
    its tiles only, the group adds the partial sums once per round, every transcript absorbs the same message. Every rank's proof must
    be the oracle's proof of that mode byte for byte - one wrong partial sum changes every later challenge - and the number of
    all-reduces is the number of rounds the device ran. (The ranks share the device and, as threads of one process, its hardware
    queues: the sharded form launches a round only when its challenge is known, so no kernel waits on the device.)"""
    import subprocess, sys
    from hglib import ROOT
    code = _SHARDED_SEQ_CODE % dict(root=ROOT, tests=os.path.join(ROOT, "tests"), n=n, k=k, world=world, mode=mode)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ), cwd=ROOT)
    assert r.returncode == 0 and "SHARDED SEQ OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    print(r.stdout.strip().splitlines()[-1])


_OWNED_SEQ_CODE = r"""
import sys, threading
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import __graft_entry__ as entry
import orclib
hg = entry.load_package()
n, k, world, mode = %(n)d, %(k)d, %(world)d, %(mode)d
bfv = hg.BfvEncrypt.new(n, k)
w = hg.Witness.synthetic(bfv.params, 700 + n + mode)
ref, _ = orclib.prove_f("goldilocks", orclib.params(n, k), orclib.Inputs(w.arrays()), threads=8, mode=mode)
group = hg.Group.local(world)
results, errs = [None] * world, []
def run(r):
    try:
        c = hg.Context(0)
        pk = bfv.setup(c)
        v = hg.witness_gen_shard(c, pk, w, r, world)      # this rank's tables only
        info = v.info()
        assert info["resident_bytes"] < info["full_bytes"] and info["resident_tables"] < info["tables"], info
        out = hg.ProofBuffer()
        hg.prove_resident_mode_sharded(c, pk, v, out, mode, r, group)
        results[r] = (out.bytes(), out.timings(), info)
        v.free(); pk.free()
    except Exception as e:
        errs.append((r, repr(e)))
ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
for t in ts: t.start()
for t in ts: t.join(600)
assert not errs, errs
for r in range(world):
    assert results[r] is not None and results[r][0] == ref, r
n_red = [int(results[r][1]["replay_ms"]) for r in range(world)]
assert len(set(n_red)) == 1 and n_red[0] > 100, n_red      # every rank took part in the same all-reduces
print("OWNED SEQ OK mode %%d n=%%d world=%%d: %%d all-reduces per rank, resident %%s of %%.1f MB" %% (mode, n, world, n_red[0],
      " / ".join("%%.1f" %% (results[r][2]["resident_bytes"] / 1e6) for r in range(world)), results[0][2]["full_bytes"] / 1e6))
"""


@pytest.mark.parametrize("n,k,world,mode", [(4096, 2, 2, 3), (4096, 2, 3, 1), (32768, 16, 2, 3)])
def test_sharded_round_by_round_prover_with_node_ownership(n, k, world, mode):
    """SURVEY 8(e) "one CRT modulus per GPU, allreduce per round" for the absorbing transcript WITHOUT replicating the witness (round 6):
    every rank holds its share of the node tables only (hg_witness_gen_shard: the Lasso input + the inputs of the node reductions it
    owns, chains dealt by modulus) and hg_prove_resident_mode_sharded runs a Vanilla / FFT node's reduction on its owner alone; the
    other ranks launch nothing for it and join the same all-reduces with zeros. Every rank's proof must be the oracle's proof of the
    mode byte for byte, with less than the full set of tables resident on each (ranks as threads of one process on device 0)."""
    import subprocess, sys
    from hglib import ROOT
    code = _OWNED_SEQ_CODE % dict(root=ROOT, tests=os.path.join(ROOT, "tests"), n=n, k=k, world=world, mode=mode)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500, env=dict(os.environ), cwd=ROOT)
    assert r.returncode == 0 and "OWNED SEQ OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    print(r.stdout.strip().splitlines()[-1])


def test_sharded_round_by_round_prover_at_the_headline_size():
    """The same at n=32768 k=16, mode 3, two ranks: bit-exact against the single-rank prove."""
    import threading
    n, k, world, mode = 32768, 16, 2, 3
    bfv = hg.BfvEncrypt.new(n, k)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    c0 = hg.Context(0); pk0 = bfv.setup(c0); v0 = hg.witness_gen(c0, pk0, w); out0 = hg.ProofBuffer()
    ref = hg.prove_resident_mode(c0, pk0, v0, out0, mode).bytes()
    group = hg.Group.local(world)
    results, errs = [None] * world, []

    def run(r):
        try:
            c = c0 if r == 0 else hg.Context(0)
            pk = pk0 if r == 0 else bfv.setup(c)
            v = v0 if r == 0 else hg.witness_gen(c, pk, w)
            out = hg.ProofBuffer()
            hg.prove_resident_mode_sharded(c, pk, v, out, mode, r, group)
            results[r] = (out.bytes(), out.timings())
        except Exception as e:
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts: t.start()
    for t in ts: t.join(600)
    assert not errs, errs
    assert results[0][0] == ref and results[1][0] == ref
    print("mode 3 n=32768 k=16, 2 ranks on one GPU: %.1f / %.1f ms, %d all-reduces" % (results[0][1]["prove_ms"], results[1][1]["prove_ms"], int(results[0][1]["replay_ms"])))
    pk0.free()


def test_strict_memory_model_build_is_bit_exact():
    """build/strict/libhypergreco.so (-DHG_STRICT_TICKETS: acq_rel tickets, the form every architecture other than gfx942 / gfx950 must
    use; built by __graft_entry__.build()) proves the same bytes as the oracle - plain launches, graph replays, a sharded proof and
    the sequential prover. The same build caps the stride-layout launches at 48 workgroups: at n=32768 k=16 every round
    kernel then walks long tile loops (the pipelined item streams cross many tile boundaries) and the slot-form hash kernel's
    workgroups straddle segment pairs (its restaging path)."""
    import subprocess, sys
    from hglib import ROOT
    lib = os.path.join(ROOT, "build", "strict", "libhypergreco.so")
    if not os.path.exists(lib):
        pytest.skip("build/strict/libhypergreco.so not built (run __graft_entry__.build())")
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import __graft_entry__ as entry\n"
        "import orclib\n"
        "hg = entry.load_package()\n"
        "assert 'strict' in hg.build.__globals__['_LIB_PATH']\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(4096, 2); pk = bfv.setup(ctx)\n"
        "w = hg.Witness.synthetic(bfv.params, 29); v = hg.witness_gen(ctx, pk, w); out = hg.ProofBuffer()\n"
        "inp = orclib.Inputs(w.arrays())\n"
        "ref, _ = orclib.prove(orclib.params(4096, 2), inp, threads=4)\n"
        "for i in range(5): assert hg.prove_resident(ctx, pk, v, out).bytes() == ref, i\n"
        "parts = [np.array(hg.prove_shard_begin(ctx, pk, v, r, 3), copy=True) for r in range(3)]\n"
        "hg.prove_shard_combine(ctx, np.stack(parts), 3)\n"
        "assert hg.prove_shard_finish(ctx, out).bytes() == ref\n"
        "ref3, _ = orclib.prove_f('goldilocks', orclib.params(4096, 2), inp, threads=4, mode=3)\n"
        "assert bfv.prove(ctx, pk, w, mode=3)[0] == ref3\n"
        "pk.free(); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)\n"   # (48 workgroups per launch: the slot-form hash kernel restages)
        "w = hg.Witness.synthetic(bfv.params, 31); v = hg.witness_gen(ctx, pk, w)\n"
        "ref, _ = orclib.prove(orclib.params(32768, 16), orclib.Inputs(w.arrays()), threads=16)\n"
        "for i in range(4): assert hg.prove_resident(ctx, pk, v, out).bytes() == ref, ('c3', i)\n"
        "print('STRICT OK')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HG_LIB=lib), cwd=ROOT)
    assert r.returncode == 0 and "STRICT OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("switch", ["HG_SEQ_CLASSIC=1", "HG_SEQ_NO_MAIL=1", "HG_SEQ_SYNC_EVERY=1", "HG_SEQ_SYNC_EVERY=0", "HG_SEQ_HOST_TAIL=0", "HG_SEQ_HOST_TAIL=64", "HG_SEQ_HOST_TAIL=8192"])
def test_sequential_prover_switches_stay_bit_exact(switch):
    """The sequential prover's alternatives - the fast path's round kernels run twice per round through the mailbox (CLASSIC), one
    stream synchronisation per round instead of the mailbox (NO_MAIL), a real synchronisation at every / no drain point, the last
    rounds of every sum-check on the device (HOST_TAIL=0) or on the host from 64 / 8192 table entries on - give the
    oracle's bytes in modes 1 and 3 (child process: the library reads the switches once)."""
    import subprocess, sys
    from hglib import ROOT
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as entry\n"
        "import orclib\n"
        "hg = entry.load_package()\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(4096, 2); pk = bfv.setup(ctx)\n"
        "w = hg.Witness.synthetic(bfv.params, 23)\n"
        "inp = orclib.Inputs(w.arrays())\n"
        "for mode in (1, 3):\n"
        "    ref, _ = orclib.prove_f('goldilocks', orclib.params(4096, 2), inp, threads=4, mode=mode)\n"
        "    for i in range(2):\n"
        "        proof, tm = bfv.prove(ctx, pk, w, mode=mode)\n"
        "        assert proof == ref, (mode, i)\n"
        "print('SEQ OK', int(tm['sync_ms']), int(tm['enqueue_ms']))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    name, value = switch.split("=")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **{name: value}), cwd=ROOT)
    assert r.returncode == 0 and "SEQ OK" in r.stdout, (switch, r.stdout[-500:], r.stderr[-2000:])
    syncs, trips = (int(x) for x in r.stdout.split("SEQ OK")[1].split()[:2])
    if switch == "HG_SEQ_NO_MAIL=1":
        assert trips == 0 and syncs > 100
    else:
        assert trips > 100


def test_protocol_modes_at_the_headline_size(ctx):
    """The same at n=32768 k=16 (mode 3 = both fixes): bit-exact against the oracle, accepted by both verifiers."""
    n, k = 32768, 16
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    proof, tm = bfv.prove(ctx, pk, w, mode=3)
    p = orclib.params(n, k)
    inp = orclib.Inputs(w.arrays())
    threads = min(64, os.cpu_count() or 8)
    ref, _ = orclib.prove_f("goldilocks", p, inp, threads=threads, mode=3)
    assert proof == ref, _first_diff(proof, ref, 16)
    assert hg.verify(pk, w, proof, mode=3) == (True, "")
    proof2, tm = bfv.prove(ctx, pk, w, mode=3)   # (second call: warm arena)
    assert proof2 == ref
    print("mode 3 n=32768 k=16: prove %.1f ms, %d stream synchronisations, %d mailbox round trips" % (tm["prove_ms"], int(tm["sync_ms"]), int(tm["enqueue_ms"])))
    # the device's rounds go through the mailbox (the last rounds of a sum-check run on the host: of the 1800 rounds about 1000 are
    # round trips); what is left of synchronisations: a real one at every fourth drain point
    assert int(tm["sync_ms"]) <= 128 and 600 <= int(tm["enqueue_ms"]) <= 1800
    pk.free()


@pytest.mark.parametrize("nb,nv", [(2, 5), (6, 10), (50, 12), (3, 15), (2, 17)])  # (the last two: three tree levels per launch, once and twice)
def test_grand_product_entry_bit_exact(ctx, nb, nv):
    """hg_grand_product (the Goldilocks counterpart of hg_grand_product_bn254) against the oracle's
    prove_grand_product on random tables: proof bytes, final claims and the point."""
    rng = random.Random(100 * nb + nv)
    tabs = [rand_f(rng, 1 << nv) for _ in range(nb)]
    skip = rng.randrange(40)
    proof, claims, point = hg.grand_product(ctx, tabs, skip)
    ref, rclaims, rpoint = orclib.grand_product_f("goldilocks", [[int(v) for v in t] for t in tabs], skip, threads=4)
    assert proof == ref
    assert [int(a) | (int(b) << 64) for a, b in claims] == rclaims
    assert [int(a) | (int(b) << 64) for a, b in point] == rpoint


def test_fold_entry_matches_the_definition(ctx):
    """hg_fold = fix_var on the LOWEST variable (convention C3): out[j] = T[2j] + r (T[2j+1] - T[2j]) over GoldilocksExt2."""
    rng = random.Random(5)

    def emul(a, b):
        return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    r = (rng.randrange(P), rng.randrange(P))
    for nv, base in ((1, True), (9, True), (12, False), (3, False)):
        t = rand_f(rng, (1 << nv) * (1 if base else 2))
        got = hg.fold(ctx, t, base, r)
        ent = [(int(v), 0) for v in t] if base else [(int(t[2 * i]), int(t[2 * i + 1])) for i in range(1 << nv)]
        for j in range(1 << (nv - 1)):
            x, y = ent[2 * j], ent[2 * j + 1]
            d = ((y[0] - x[0]) % P, (y[1] - x[1]) % P)
            rd = emul(r, d)
            assert (int(got[j][0]), int(got[j][1])) == ((x[0] + rd[0]) % P, (x[1] + rd[1]) % P), (nv, base, j)


def test_derived_parameter_set_proves_and_verifies(ctx):
    """A parameter set that is NOT one of the six shipped ones (hg_params_derive: n = 2048 with two 55-bit moduli): setup, synthetic
    witness, prove, bit-exact against the oracle on the same derived constants, accepted by both verifiers."""
    c = orclib.constants(4096, 2)
    params = hg.params_derive(2048, 2, c["qis"], 65537)
    bfv = hg.BfvEncrypt(params)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(params, 99)
    proof, _ = bfv.prove(ctx, pk, w)
    assert (pk.circuit_eval(w)[1] == w.arrays()["ct0is"]).all()   # the synthetic witness satisfies the circuit relation for the derived constants
    oc = dict(n=2048, k=2, s_bound=params.s_bound, e_bound=params.e_bound, k1_bound=params.k1_bound,
              r1_bounds=list(params.r1_bounds)[:2], r2_bounds=list(params.r2_bounds)[:2], qis=list(params.qis)[:2], k0is=list(params.k0is)[:2])
    p = orclib.Params(oc)
    inp = orclib.Inputs(w.arrays())
    ref, _ = orclib.prove(p, inp, threads=4)
    assert proof == ref
    assert orclib.verify(p, inp, proof)[0] and hg.verify(pk, w, proof) == (True, "")
    pk.free()


def test_cached_launch_graph_replays_and_invalidates(ctx):
    """Resident proves of one (key, values object) pair: the third is captured into a hipGraph, later ones replay it. The bytes must
    not change; other users of the context's arena in between must NOT disturb it (the graph works in a private arena); another
    values object gets its own graph and both replay in alternation; the oracle agrees throughout."""
    n, k = 4096, 2
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w1, w2 = hg.Witness.synthetic(bfv.params, 11), hg.Witness.synthetic(bfv.params, 12)
    v1, v2 = hg.witness_gen(ctx, pk, w1), hg.witness_gen(ctx, pk, w2)
    out = hg.ProofBuffer()
    p = orclib.params(n, k)
    ref1, _ = orclib.prove(p, orclib.Inputs(w1.arrays()), threads=4)
    ref2, _ = orclib.prove(p, orclib.Inputs(w2.arrays()), threads=4)
    for i in range(6):                                                     # walk, walk, capture, replay x3
        assert hg.prove_resident(ctx, pk, v1, out).bytes() == ref1, i
    rng = random.Random(3)
    tab = rand_f(rng, 1 << 10)
    ctx.mle_eval(tab, rand_f(rng, 20))                                     # another user of the context's arena
    for i in range(4):
        assert hg.prove_resident(ctx, pk, v1, out).bytes() == ref1, i      # still the same graph
    assert hg.prove_resident(ctx, pk, v2, out).bytes() == ref2             # other values: never v1's graph
    for i in range(5):
        assert hg.prove_resident(ctx, pk, v2, out).bytes() == ref2, i      # (its own graph from its third prove on)
        assert hg.prove_resident(ctx, pk, v1, out).bytes() == ref1, i
        ctx.mle_eval(tab, rand_f(rng, 20))
    ctx.set_option("graph", 0)
    for i in range(4):
        assert hg.prove_resident(ctx, pk, v1, out).bytes() == ref1, i
    ctx.set_option("graph", 1)
    v1.free(); v2.free(); pk.free()


def test_warmup_makes_the_first_prove_a_graph_replay(capfd, monkeypatch):
    """hg_warmup (what a drop-in BfvEncrypt::setup calls behind hg_setup; the reference's caller proves ONCE per witness, test.rs:37-38):
    the context-owned tables, the witness staging and the launch graph exist before the first witness arrives - the first hg_prove of
    a fresh context replays (HG_DEBUG=launch reports every hipGraphLaunch) and its bytes are the oracle's, for every later witness too."""
    n, k = 4096, 2
    c = hg.Context(0)
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(c)
    assert bfv.warmup(c, pk) > 0
    p = orclib.params(n, k)
    monkeypatch.setenv("HG_DEBUG", "launch")
    for seed in (31, 32, 33):
        w = hg.Witness.synthetic(bfv.params, seed)
        ref, _ = orclib.prove(p, orclib.Inputs(w.arrays()), threads=4)
        capfd.readouterr()
        proof, _ = bfv.prove(c, pk, w)
        assert "hipGraphLaunch" in capfd.readouterr().err, f"witness {seed}: hg_prove after hg_warmup walked the protocol"
        assert proof == ref
    monkeypatch.delenv("HG_DEBUG")
    # without the warm-up the first prove of a fresh context walks (and is the same proof)
    c2 = hg.Context(0)
    pk2 = bfv.setup(c2)
    monkeypatch.setenv("HG_DEBUG", "launch")
    capfd.readouterr()
    proof2, _ = bfv.prove(c2, pk2, w)
    assert "hipGraphLaunch" not in capfd.readouterr().err
    assert proof2 == ref
    pk.free(); pk2.free(); c.close(); c2.close()


@pytest.mark.parametrize("n,k,seeds", [(4096, 2, 5), (32768, 16, 3)])
def test_graph_replay_across_witnesses(ctx, n, k, seeds):
    """The steady state of a prover that gets a NEW witness per proof (the reference proves each witness once, test.rs:37-38):
    one values object refilled in place by hg_witness_gen_into, proven through the launch graph recorded for its addresses.
    Every proof - walked, captured, replayed - must equal the oracle's proof of THAT witness (BASELINE configs[1] and [2])."""
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    p = orclib.params(n, k)
    threads = min(64, os.cpu_count() or 8)
    ws = [hg.Witness.synthetic(bfv.params, 0x77 + 13 * i) for i in range(seeds)]
    refs = [orclib.prove(p, orclib.Inputs(w.arrays()), threads=threads)[0] for w in ws]
    assert len(set(refs)) == seeds
    out = hg.ProofBuffer()
    vals = hg.witness_gen(ctx, pk, ws[0])
    for i in range(3):                                                     # walk, walk, capture + first replay
        assert hg.prove_resident(ctx, pk, vals, out).bytes() == refs[0], i
    launches = []
    for rnd in range(2):
        for i in range(seeds):                                             # every later prove: another witness, the same graph
            hg.witness_gen_into(ctx, pk, ws[i], vals)
            assert hg.prove_resident(ctx, pk, vals, out).bytes() == refs[i], _first_diff(out.bytes(), refs[i], 16)
            launches.append(out.timings()["enqueue_ms"])
    # hg_prove (the drop-in for BfvEncrypt::prove) does the same with tables owned by the context
    for rnd in range(2):
        for i in range(seeds):
            proof, tm = bfv.prove(ctx, pk, ws[i])
            assert proof == refs[i], (rnd, i)
    print("n=%d: enqueue ms per replayed prove %s; hg_prove end to end %s" % (n, ["%.2f" % x for x in launches[-3:]], {q: round(tm[q], 3) for q in ("upload_ms", "witness_ms", "prove_ms", "total_ms")}))
    vals.free()
    pk.free()


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(1024, 1), (4096, 2)])
def test_prove_stream_equals_one_by_one(ctx, n, k):
    """hg_prove_stream (witness i+1 uploaded and evaluated on a third stream into a second table set while witness i is proven):
    every proof of a run - walked ones at its start, replayed ones with the overlap later, runs of length 0 / 1 / odd / even, a
    second run on warm graphs - equals the oracle's proof of that witness, in order."""
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    p = orclib.params(n, k)
    ws = [hg.Witness.synthetic(bfv.params, 0x1234 + 7 * i) for i in range(5)]
    refs = [orclib.prove(p, orclib.Inputs(w.arrays()), threads=min(16, os.cpu_count() or 8))[0] for w in ws]
    assert len(set(refs)) == len(ws)
    assert bfv.prove_stream(ctx, pk, [])[0] == []
    assert bfv.prove_stream(ctx, pk, ws[:1])[0] == refs[:1]
    order = [0, 1, 2, 3, 4, 4, 3, 0, 2, 1, 1]                               # 11 proofs: both table sets get past their capture
    for rnd in range(3):
        proofs, tm = bfv.prove_stream(ctx, pk, [ws[i] for i in order])
        for j, i in enumerate(order):
            assert proofs[j] == refs[i], (rnd, j, _first_diff(proofs[j], refs[i], 16))
    print("n=%d: %d proofs in %.2f ms (%.3f ms each; sum of prove_ms %.2f, gpu_ms %.2f)" % (n, len(order), tm["total_ms"], tm["total_ms"] / len(order), tm["prove_ms"], tm["gpu_ms"]))
    # hg_prove afterwards (tables of its own) is unaffected, and so is a stream under another key
    assert bfv.prove(ctx, pk, ws[2])[0] == refs[2]
    n2, k2 = (4096, 2) if n == 1024 else (1024, 1)
    bfv2 = hg.BfvEncrypt.new(n2, k2)
    pk2 = bfv2.setup(ctx)
    w2 = hg.Witness.synthetic(bfv2.params, 5)
    ref2 = orclib.prove(orclib.params(n2, k2), orclib.Inputs(w2.arrays()), threads=4)[0]
    assert bfv2.prove_stream(ctx, pk2, [w2, w2, w2, w2])[0] == [ref2] * 4
    assert bfv.prove_stream(ctx, pk, [ws[3], ws[0]])[0] == [refs[3], refs[0]]
    with pytest.raises(hg.HgError):
        bfv2.prove_stream(ctx, pk2, [ws[0]])                                 # a witness of another parameter set
    pk2.free()
    pk.free()


def test_prove_stream_at_the_headline_size(ctx):
    """hg_prove_stream at n=32768 k=16: a run that takes both table sets past their capture gives, proof by proof, the ORACLE's
    transcript of that witness (three oracle proves, about 3 s each on the GPU box), and hg_prove gives the same bytes."""
    n, k = 32768, 16
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    p = orclib.params(n, k)
    threads = min(64, os.cpu_count() or 8)
    ws = [hg.Witness.synthetic(bfv.params, 0x77 + 13 * i) for i in range(3)]
    refs = [orclib.prove(p, orclib.Inputs(w.arrays()), threads=threads)[0] for w in ws]
    assert len(set(refs)) == 3
    order = [0, 1, 2, 2, 1, 0, 0, 2, 1]
    for rnd in range(2):
        proofs, tm = bfv.prove_stream(ctx, pk, [ws[i] for i in order])
        for j, i in enumerate(order):
            assert proofs[j] == refs[i], (rnd, j, _first_diff(proofs[j], refs[i], 16))
    assert [bfv.prove(ctx, pk, w)[0] for w in ws] == refs
    print("n=32768: %d proofs in %.2f ms (%.3f ms each)" % (len(order), tm["total_ms"], tm["total_ms"] / len(order)))
    pk.free()


def test_graph_cache_eviction_and_refill_guards(ctx):
    """One graph slot (HG_GRAPH_ENTRIES=1, child process): two values objects proven in alternation evict each other's graph over
    and over - the bytes never change. In this process: a values object cannot be refilled for another key, and reports its size."""
    import subprocess, sys
    from hglib import ROOT
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as entry\n"
        "import orclib\n"
        "hg = entry.load_package()\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(4096, 2); pk = bfv.setup(ctx); out = hg.ProofBuffer()\n"
        "ws = [hg.Witness.synthetic(bfv.params, 61 + i) for i in range(2)]\n"
        "vs = [hg.witness_gen(ctx, pk, w) for w in ws]\n"
        "refs = [orclib.prove(orclib.params(4096, 2), orclib.Inputs(w.arrays()), threads=4)[0] for w in ws]\n"
        "for i in range(10):\n"
        "    for j in range(2): assert hg.prove_resident(ctx, pk, vs[j], out).bytes() == refs[j], (i, j)\n"
        "print('EVICT OK')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HG_GRAPH_ENTRIES="1"), cwd=ROOT)
    assert r.returncode == 0 and "EVICT OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    bfv = hg.BfvEncrypt.new(4096, 2)
    pk, pk2 = bfv.setup(ctx), bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 63)
    vals = hg.witness_gen(ctx, pk, w)
    info = vals.info()
    assert info["resident_bytes"] == info["full_bytes"] > 0 and info["resident_tables"] == info["tables"] == pk.num_nodes
    with pytest.raises(hg.HgError, match="another prover key"):
        hg.witness_gen_into(ctx, pk2, w, vals)
    with pytest.raises(hg.HgError, match="another prover key"):
        hg.prove_resident(ctx, pk2, vals, hg.ProofBuffer())
    vals.free(); pk.free(); pk2.free()


def test_graph_capture_failure_falls_back_to_plain_launches(ctx, monkeypatch):
    """A launch-graph capture that fails (forced) must not fail the prove nor be retried on every call: the key walks from then on."""
    n, k = 4096, 2
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 31)
    vals = hg.witness_gen(ctx, pk, w)
    out = hg.ProofBuffer()
    ref = hg.prove_resident(ctx, pk, vals, out).bytes()
    monkeypatch.setenv("HG_DEBUG", "fail_capture")
    for i in range(4):                                                     # walk, failed capture -> walk, walk, walk
        assert hg.prove_resident(ctx, pk, vals, out).bytes() == ref, i
    monkeypatch.delenv("HG_DEBUG")
    for i in range(3):                                                     # the key stays on plain launches (no retry), still correct
        assert hg.prove_resident(ctx, pk, vals, out).bytes() == ref, i
        assert out.timings()["enqueue_ms"] > 0.05
    vals.free()
    pk.free()


def test_sharded_prove_in_flight_survives_other_proves(ctx):
    """hg_prove_shard_begin .. _finish with other calls in between that drop or replace launch graphs (the pending
    shard used to hold a raw pointer into the cache): the finish must still replay the right transcript; a finish without a begin
    is an error, and a context destroyed with a shard in flight does not crash."""
    n, k, world = 4096, 2, 2
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w, w2 = hg.Witness.synthetic(bfv.params, 41), hg.Witness.synthetic(bfv.params, 42)
    vals, vals2 = hg.witness_gen(ctx, pk, w), hg.witness_gen(ctx, pk, w2)
    out, out2 = hg.ProofBuffer(), hg.ProofBuffer()
    ref = hg.prove_resident(ctx, pk, vals, out).bytes()
    for i in range(4):                                                     # rank 0's share: walk, walk, capture, replay
        p0 = hg.prove_shard_begin(ctx, pk, vals, 0, world).copy()
    p1 = hg.prove_shard_begin(ctx, pk, vals, 1, world).copy()
    hg.prove_shard_begin(ctx, pk, vals, 0, world)                          # in flight, replayed from rank 0's graph
    ctx.set_option("graph", 0)                                             # drops every cached graph
    ctx.set_option("graph", 1)
    hg.prove_shard_combine(ctx, np.stack([p0, p1]), world)
    assert hg.prove_shard_finish(ctx, out).bytes() == ref
    with pytest.raises(hg.HgError, match="no sharded prove in flight"):
        hg.prove_shard_finish(ctx, out)
    # a prove of other values between begin and finish overwrites the shared result buffer; combine reinstalls it, finish is right
    hg.prove_shard_begin(ctx, pk, vals, 0, world)
    hg.prove_resident(ctx, pk, vals2, out2)
    hg.prove_shard_combine(ctx, np.stack([p0, p1]), world)
    assert hg.prove_shard_finish(ctx, out).bytes() == ref
    c2 = hg.Context(0)
    pk2 = bfv.setup(c2)
    v2 = hg.witness_gen(c2, pk2, w)
    hg.prove_shard_begin(c2, pk2, v2, 0, world)
    v2.free(); pk2.free(); c2.close()                                      # shard in flight: dropped with the context
    vals.free(); vals2.free(); pk.free()


def test_launch_graph_that_replays_slower_than_plain_launches_is_given_up():
    """The library compares the first replays of a captured launch graph with the walked prove it recorded (the runtime decides which
    stream a graph's branches run on) and falls back to plain launches for that key when the graph is clearly slower. Forced here with
    HG_GRAPH_GUARD_FACTOR=0 in a child process: proofs stay identical and only the four checked replays go through the graph."""
    import subprocess, sys
    from hglib import ROOT
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as entry\n"
        "hg = entry.load_package()\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(4096, 2); pk = bfv.setup(ctx)\n"
        "w = hg.Witness.synthetic(bfv.params, 5); v = hg.witness_gen(ctx, pk, w); out = hg.ProofBuffer()\n"
        "first = None\n"
        "for i in range(12):\n"
        "    hg.prove_resident(ctx, pk, v, out); first = first or out.bytes(); assert out.bytes() == first, i\n"
    ) % ROOT
    env = dict(os.environ, HG_GRAPH_GUARD_FACTOR="0", HG_DEBUG="launch")   # (the second one logs every graph launch to stderr)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    # proves 0, 1: walks; 2: capture + first replay; 3 .. 5: replays - all four "slower" than the walk; 6 ..: walks again
    assert r.stderr.count("hipGraphLaunch:") == 4, r.stderr[-2000:]
    env = dict(os.environ, HG_DEBUG="launch")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stderr.count("hipGraphLaunch:") == 10, r.stderr[-2000:]   # the guard leaves a healthy graph alone


@pytest.mark.parametrize("switch", ["HG_ONE_STREAM=1", "HG_NO_GRAPH=1", "HG_GATHER_CSR=1", "HG_NO_PS_EQ=1", "HG_SLOT_DEPTH=0", "HG_SLOT_DEPTH=1", "HG_SLOT_DEPTH=2", "HG_NO_SPLIT=1", "HG_PHASE2_LATE=1"])
def test_alternative_paths_behind_the_environment_switches_stay_bit_exact(switch):
    """The switches that are left select a supported configuration (one stream, plain launches) or force the GENERAL form of a
    path at a size where the specialised one would run (per-term Libra gathers instead of run-length segments, every Libra table
    materialised instead of the eq-factored rounds, fewer or no slot-form layers in the read / write product): each must produce the
    same bytes as the oracle. The library reads them once per process: child process,
    five resident proves (walks, capture, replays) plus a four-rank sharded proof at n=4096 k=2."""
    import subprocess, sys
    from hglib import ROOT
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import __graft_entry__ as entry\n"
        "import orclib\n"
        "hg = entry.load_package()\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(4096, 2); pk = bfv.setup(ctx)\n"
        "w = hg.Witness.synthetic(bfv.params, 21); v = hg.witness_gen(ctx, pk, w); out = hg.ProofBuffer()\n"
        "ref, _ = orclib.prove(orclib.params(4096, 2), orclib.Inputs(w.arrays()), threads=4)\n"
        "for i in range(5): assert hg.prove_resident(ctx, pk, v, out).bytes() == ref, i\n"
        "parts = [np.array(hg.prove_shard_begin(ctx, pk, v, r, 4), copy=True) for r in range(4)]\n"
        "hg.prove_shard_combine(ctx, np.stack(parts), 4)\n"
        "assert hg.prove_shard_finish(ctx, out).bytes() == ref\n"
        "print('SWITCH OK')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    name, value = switch.split("=")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **{name: value}), cwd=ROOT)
    assert r.returncode == 0 and "SWITCH OK" in r.stdout, (switch, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("switch", ["HG_GATHER_CSR=1", "HG_NO_PS_EQ=1", "HG_SLOT_DEPTH=0", "HG_SLOT_DEPTH=1", "HG_SLOT_DEPTH=3", "HG_NO_SPLIT=1", "HG_PHASE2_LATE=1"])
def test_environment_switches_at_the_headline_size(switch):
    """Two of the switches above at n=32768 k=16 (BASELINE configs[2]), where every production shortcut is active: walks, the capture and
    two graph replays must all give the oracle's bytes."""
    import subprocess, sys
    from hglib import ROOT
    code = (
        "import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as entry\n"
        "import orclib\n"
        "hg = entry.load_package()\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)\n"
        "w = hg.Witness.synthetic(bfv.params, 77); v = hg.witness_gen(ctx, pk, w); out = hg.ProofBuffer()\n"
        "ref, _ = orclib.prove(orclib.params(32768, 16), orclib.Inputs(w.arrays()), threads=min(16, os.cpu_count() or 8))\n"
        "for i in range(5): assert hg.prove_resident(ctx, pk, v, out).bytes() == ref, i\n"
        "print('SWITCH OK')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    name, value = switch.split("=")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **{name: value}), cwd=ROOT)
    assert r.returncode == 0 and "SWITCH OK" in r.stdout, (switch, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("switch", ["HG_GATHER_CSR=1", "HG_BN_SLOT_DEPTH=0", "HG_BN_SLOT_DEPTH=1", "HG_BN_SLOT_DEPTH=2", "HG_BN_NO_FUSED0=1"])
def test_bn254_environment_switches_stay_bit_exact(switch):
    """The BN254 prove with its alternative paths (every pair of a PRODSUM job multiplied separately although the pairs share one b
    table; per-term Libra gathers instead of aliased eq slices; the node launches ahead of the Lasso node) gives the C++ Fr oracle's
    bytes at n=4096 k=2. Child process: the switches are read once."""
    import subprocess, sys
    from hglib import ROOT
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import __graft_entry__ as entry\n"
        "import orclib\n"
        "hg = entry.load_package()\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(4096, 2); pk = bfv.setup(ctx)\n"
        "w = hg.Witness.synthetic(bfv.params, 31)\n"
        "ref = orclib.prove_f('bn254', orclib.params(4096, 2), orclib.Inputs(w.arrays()), threads=4)[0]\n"
        "for i in range(2): assert ctx.prove_bn254(pk, w, cap=1 << 24)[0] == ref, i\n"
        "print('SWITCH OK')\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    name, value = switch.split("=")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **{name: value}), cwd=ROOT)
    assert r.returncode == 0 and "SWITCH OK" in r.stdout, (switch, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("n,k", [(1024, 1), (4096, 2), (32768, 16)])
def test_device_verifier_agrees_with_the_host_verifier(ctx, n, k):
    """hg_verify_device (table-sized checks as kernels) against hg_verify (host) and the oracle's verifier: the HIP proof is accepted by all
    three; a byte flipped anywhere in the proof gets the same accept / reject decision from the device and the host verifier (the
    reference's verifier does not bind every byte - the collation sum-check's final evaluation, trailing bytes - so some flips are
    accepted by both); a proof for another witness is rejected."""
    import time
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w, w2 = hg.Witness.synthetic(bfv.params, 0xabc + n), hg.Witness.synthetic(bfv.params, 0xabd + n)
    proof, _ = bfv.prove(ctx, pk, w)
    ok, why = hg.verify_device(ctx, pk, w, proof)
    assert ok, why
    assert hg.verify(pk, w, proof) == (True, "")
    t0 = time.perf_counter(); hg.verify_device(ctx, pk, w, proof); t_dev = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); hg.verify(pk, w, proof); t_host = (time.perf_counter() - t0) * 1e3
    print("n=%d k=%d: hg_verify_device %.1f ms, hg_verify (host) %.1f ms" % (n, k, t_dev, t_host))
    ok2, _ = hg.verify_device(ctx, pk, w2, proof)
    assert not ok2 and not hg.verify(pk, w2, proof)[0]
    rng = random.Random(n)
    positions = [0, 8, len(proof) // 5, len(proof) // 3, len(proof) // 2, 2 * len(proof) // 3, len(proof) - 40, len(proof) - 1]
    positions += [rng.randrange(len(proof)) for _ in range(6 if n < 32768 else 2)]
    rejected = 0
    for pos in positions:
        bad = bytearray(proof)
        bad[pos] ^= 1 << rng.randrange(8)
        dh, dd = hg.verify(pk, w, bytes(bad))[0], hg.verify_device(ctx, pk, w, bytes(bad))[0]
        assert dh == dd, (pos, dh, dd)
        rejected += not dd
    assert rejected >= len(positions) // 2
    assert not hg.verify_device(ctx, pk, w, proof[:len(proof) // 2])[0]      # truncated
    pk.free()


def test_library_collective_single_rank_communicator(ctx):
    """hg_comm_init / hg_prove_sharded with a one-rank RCCL communicator (the only size a one-GPU box offers): the limb-split
    kernel, ncclAllReduce on the prover stream and the fold-back kernel run for real and must leave the proof unchanged."""
    n, k = 4096, 2
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    vals = hg.witness_gen(ctx, pk, w)
    out = hg.ProofBuffer()
    ref = hg.prove_resident(ctx, pk, vals, out).bytes()
    with pytest.raises(hg.HgError, match="no communicator"):
        hg.prove_sharded(ctx, pk, vals, out)
    hg.comm_init(ctx, hg.comm_unique_id(), 0, 1)
    with pytest.raises(hg.HgError, match="already has a communicator"):
        hg.comm_init(ctx, hg.comm_unique_id(), 0, 1)
    for i in range(6):   # walk, walk, capture (the share's launch graph; the all-reduce follows it on the stream), replay x3
        assert hg.prove_sharded(ctx, pk, vals, out).bytes() == ref, i
    hg.comm_destroy(ctx)
    assert hg.prove_resident(ctx, pk, vals, out).bytes() == ref
    vals.free()
    pk.free()


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_exchange_limb_kernels_for_more_than_one_rank(ctx, world):
    """k_split_limbs -> (what ncclSum does to the limb lanes of `world` ranks) -> k_combine_limbs against the lane-wise sum mod p
    in Python integers. A one-GPU box only ever forms a one-rank communicator, where both limb sums stay below 2^32; here the
    fold-back sees sums up to world (2^32 - 1), with lanes at p - 1 and at the limb boundaries on EVERY rank."""
    rng = np.random.default_rng(world)
    n = 20000 + world
    bufs = rng.integers(0, P, size=(world, n), dtype=np.uint64)
    edge = [P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF, 0, 1]
    for i, e in enumerate(edge):
        bufs[:, i] = e                       # the same extreme lane on every rank
        bufs[:, 100 + i] = 0
        bufs[i % world, 100 + i] = e         # ... and on one rank only
    got = hg.comm_selftest(ctx, bufs)
    want = np.array([sum(int(bufs[r][i]) for r in range(world)) % P for i in range(n)], dtype=np.uint64)
    assert np.array_equal(got, want)
    assert np.array_equal(hg.shard_combine_host(bufs), want)   # the caller-side combine agrees


def test_bench_sharded_two_processes_on_one_gpu():
    """The N>1 bench path end to end with two real processes (gloo all-reduce through host tensors, both ranks on
    device 0): rendezvous, per-rank job ownership, the all-reduce, replay; bench.py itself asserts that the sharded
    proof equals the single-GPU proof."""
    import json, subprocess, sys
    from hglib import ROOT
    env = dict(os.environ, HG_BENCH_BACKEND="gloo", HG_BENCH_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--ring-degree", "4096", "--crt-moduli", "2",
           "--no-cpu-baseline"]
    outp = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert outp.returncode == 0, outp.stderr[-3000:]
    lines = [l for l in outp.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] == d["ms_per_step"]


@pytest.mark.parametrize("n,k,bits", [(1024, 1, 27), (8192, 4, 55)])
def test_lasso_node_inside_a_larger_transcript(ctx, n, k, bits, tmp_path, monkeypatch):
    """hg_lasso_prove_at: the node entered after `skip` challenges writes exactly the bytes the full prover writes for it. The
    position is COMPUTED, not searched for: the proof map of the full prove names the byte offset of the node's section and the
    number of challenges squeezed before it; the section must equal hg_lasso_prove_at's output for that number, and the oracle's
    Lasso prover entered at the same point must agree."""
    import re
    mp = tmp_path / "map.tsv"
    monkeypatch.setenv("HG_PROOF_MAP", str(mp))
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = bfv.get_inputs(os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    full, _ = bfv.prove(ctx, pk, w)
    monkeypatch.delenv("HG_PROOF_MAP")
    rows = [l.rstrip("\n").split("\t", 1) for l in open(mp)]
    at = next(i for i, r in enumerate(rows) if r[1].startswith("lasso node: enters after"))
    skip = int(re.search(r"enters after (\d+) squeezed", rows[at][1]).group(1))
    start = int(rows[at][0])
    end = next(int(r[0]) for r in rows[at + 1:] if not r[1].startswith(("lasso", "grand product")))
    assert skip > 0 and 0 < start < end <= len(full)
    lasso_in, _ = pk.circuit_eval(w)
    base, _ = hg.LassoNode(pk).prove_claim_reduction(ctx, lasso_in)
    pr, _ = hg.LassoNode(pk).prove_claim_reduction(ctx, lasso_in, chain_skip=skip)
    assert len(pr) == end - start == len(base) and pr != base
    assert pr == full[start:end], _first_diff(pr, full[start:end], 16)
    ref, _ = orclib.lasso_prove_f("goldilocks", orclib.params(n, k), lasso_in, threads=8, chain_skip=skip)
    assert pr == ref, _first_diff(pr, ref, 16)
    for other in (skip - 1, skip + 1):                                     # one challenge off: other bytes
        assert hg.LassoNode(pk).prove_claim_reduction(ctx, lasso_in, chain_skip=other)[0] != pr
    pk.free()


def test_proof_map_covers_the_stream_and_diff_tool_localises_a_flip(ctx, tmp_path, monkeypatch):
    """HG_PROOF_MAP labels every byte range of the proof (scripts/proof_diff.py uses it to name the first diverging
    protocol element against a proof dumped by the Rust reference)."""
    import subprocess, sys
    mp = tmp_path / "map.tsv"
    monkeypatch.setenv("HG_PROOF_MAP", str(mp))
    bfv = hg.BfvEncrypt.new(1024, 1)
    pk = bfv.setup(ctx)
    w = bfv.get_inputs(os.path.join(orclib.GOLDEN, "sk_enc_1024_1x27_65537.json"))
    proof, _ = bfv.prove(ctx, pk, w)
    monkeypatch.delenv("HG_PROOF_MAP")
    rows = [l.rstrip("\n").split("\t", 1) for l in open(mp)]
    offs = [int(r[0]) for r in rows]
    assert offs == sorted(offs) and offs[0] == 0 and offs[-1] == len(proof) and rows[-1][1] == "end of proof"
    assert any("collation" in r[1] for r in rows) and any("fft node" in r[1] for r in rows) and any("vanilla node" in r[1] for r in rows)
    # flip one byte inside the collation rounds and let the tool find it
    start = next(int(r[0]) for r in rows if "collation" in r[1])
    bad = bytearray(proof); bad[start + 24] ^= 1
    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    a.write_bytes(proof); b.write_bytes(bytes(bad))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "proof_diff.py"), str(a), str(b), str(mp)], capture_output=True, text=True)
    assert r.returncode == 1 and "collation" in r.stdout and "byte %d" % (start + 24) in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "proof_diff.py"), str(a), str(a), str(mp)], capture_output=True, text=True)
    assert r.returncode == 0 and "identical" in r.stdout


# ---- BN254 slice (hg_bn254_field_op, hg_challenges_bn254, hg_sumcheck_bn254) ----------------------------------------
def test_bn254_field_kernels_against_python_integers(ctx):
    bn = orclib.bn254()
    rng = random.Random(254)
    edge = [0, 1, 2, bn.R - 1, bn.R - 2, (1 << 64) - 1, 1 << 64, (1 << 128) + 5, (1 << 253), bn.R >> 1]
    a = edge + [rng.randrange(bn.R) for _ in range(3000)]
    b = list(reversed(edge)) + [rng.randrange(bn.R) for _ in range(3000)]
    assert ctx.bn254_field_op(0, a, b) == [(x + y) % bn.R for x, y in zip(a, b)]
    assert ctx.bn254_field_op(1, a, b) == [(x - y) % bn.R for x, y in zip(a, b)]
    assert ctx.bn254_field_op(2, a, b) == [(x * y) % bn.R for x, y in zip(a, b)]
    assert ctx.bn254_field_op(3, a, b) == [(x * y) % bn.R for x, y in zip(a, b)]                    # column-accumulator product
    assert ctx.bn254_field_op(4, a, b) == [(x * y + x * x + y * y) % bn.R for x, y in zip(a, b)]    # three products, one reduction


def test_bn254_loose_arithmetic_against_python_integers(ctx):
    """bn254_lazy.hpp - the branch-free arithmetic of the hot bn254 kernels, where a residue is any representative below 2p, a
    difference is y - x + 2p and a reduction ends in a floating-point quotient estimate - on RAW operands including every edge of
    its contracts: 0, 1, p - 1, p, p + 1, 2p - 1 (sums / differences), and for products anything up to 2^256 - 1."""
    bn = orclib.bn254()
    p = bn.R
    Rinv = pow(1 << 256, -1, p)
    rng = random.Random(2540)
    lt2p = [0, 1, 2, p - 1, p, p + 1, 2 * p - 2, 2 * p - 1, (1 << 254) - 1, 1 << 253, (1 << 64) - 1, 1 << 64, (1 << 128) + 5]
    a = lt2p + [rng.randrange(2 * p) for _ in range(4000)]
    b = list(reversed(lt2p)) + [rng.randrange(2 * p) for _ in range(4000)]
    assert ctx.bn254_field_op(7, a, b) == [(x + y) % p for x, y in zip(a, b)]
    assert ctx.bn254_field_op(8, a, b) == [(x - y) % p for x, y in zip(a, b)]
    assert ctx.bn254_field_op(9, a, b) == [((x * y + x * x + y * y + (x - y + 2 * p) * y) * Rinv) % p for x, y in zip(a, b)]
    r = (1 << 200) + 12345
    anyv = [0, 1, p, 2 * p, 4 * p - 1, (1 << 256) - 1, (1 << 256) - 2, 1 << 255, (1 << 224) - 1, 0xFFFFFFFF, 1 << 32] + [rng.randrange(1 << 256) for _ in range(4000)]
    a6 = [a[i % len(a)] for i in range(len(anyv))]
    assert ctx.bn254_field_op(6, a6, anyv) == [(x + r * d * Rinv) % p for x, d in zip(a6, anyv)]
    a5 = anyv
    b5 = list(reversed(anyv))
    assert ctx.bn254_field_op(5, a5, b5) == [(x * y * Rinv) % p for x, y in zip(a5, b5)]
    # bn254_mfma.hpp: the fold x + r (y - x) as an int8 matrix product (any 256-bit operands; lengths that are not a multiple of 64
    # leave a partial wave). Same residues as the multiply-add form above.
    for cut in (len(anyv), 4001, 65, 3):
        assert ctx.bn254_field_op(10, a5[:cut], b5[:cut]) == [(x + r * (y - x) * Rinv) % p for x, y in zip(a5[:cut], b5[:cut])]
    # the largest sum a reduction sees in the kernels: 50 products of operands below 4p (the column accumulators carry it exactly)
    big = [4 * p - 1] * 64
    assert ctx.bn254_field_op(5, big, big) == [((4 * p - 1) ** 2 * Rinv) % p] * 64


def test_bn254_challenges_match_the_oracle():
    bn = orclib.bn254()
    assert hg.challenges_bn254(40) == bn.challenges(40, orclib.keccak256)


@pytest.mark.parametrize("kind,ntab,nv", [(0, 6, 8), (0, 25, 5), (1, 4, 9), (1, 50, 4), (2, 2, 10), (2, 6, 7), (1, 2, 1), (2, 2, 0)])
def test_bn254_sumcheck_kernels_bit_exact(ctx, kind, ntab, nv):
    """HIP sum-check rounds over bn256::Fr (256-bit Montgomery limbs) against the Python-integer oracle, all three shapes."""
    bn = orclib.bn254()
    rng = random.Random(1000 * kind + 10 * ntab + nv)
    tabs = [[rng.randrange(bn.R) for _ in range(1 << nv)] for _ in range(ntab)]
    npw = ntab if kind == 0 else (ntab // 2 if kind == 1 else 0)
    pw = [rng.randrange(bn.R) for _ in range(npw)]
    claim = rng.randrange(bn.R)
    skip = rng.randrange(30)
    chal = bn.challenges(skip + nv, orclib.keccak256)[skip:]
    msgs, point, evals, sums = ctx.sumcheck_bn254(kind, tabs, pw, claim, skip)
    emsgs, eevals, esums = bn.sumcheck(kind, tabs, pw, claim, chal)
    assert point == chal and msgs == emsgs and evals == eevals and sums == esums


@pytest.mark.parametrize("nv", [0, 1, 5, 11])
def test_bn254_mle_eval(ctx, nv):
    bn = orclib.bn254()
    rng = random.Random(nv)
    table = [rng.randrange(bn.R) for _ in range(1 << nv)]
    point = [rng.randrange(bn.R) for _ in range(nv)]
    assert ctx.mle_eval_bn254(table, point) == bn.mle_eval(table, point)


@pytest.mark.parametrize("log2n", [1, 4, 8])
def test_bn254_ntt_matches_the_definition_and_inverts(ctx, log2n):
    bn = orclib.bn254()
    rng = random.Random(log2n)
    rows = [[rng.randrange(bn.R) for _ in range(1 << log2n)] for _ in range(3)]
    fwd = ctx.ntt_bn254(rows)
    assert fwd == [bn.ntt(r) for r in rows]
    assert ctx.ntt_bn254(fwd, inverse=True) == rows
    big = [[rng.randrange(bn.R) for _ in range(1 << 14)]]   # size-independent property at a larger size: round trip
    assert ctx.ntt_bn254(ctx.ntt_bn254(big), inverse=True) == big


@pytest.mark.parametrize("nb,nv", [(1, 1), (2, 3), (6, 6), (50, 4), (4, 9)])
def test_bn254_grand_product_bit_exact(ctx, nb, nv):
    """prove_grand_product over Fr (product tree, root products, layered degree-3 sum-checks with the poly(0) quirk, mu folds)
    against the Python oracle: proof bytes (32-byte big-endian elements), final claims and point."""
    bn = orclib.bn254()
    rng = random.Random(100 * nb + nv)
    tabs = [[rng.randrange(bn.R) for _ in range(1 << nv)] for _ in range(nb)]
    skip = rng.randrange(20)
    need = 1 + sum(2 + n for n in range(1, nv))
    chal = bn.challenges(skip + need, orclib.keccak256)[skip:]
    proof, claims, point = ctx.grand_product_bn254(tabs, skip)
    eproof, eclaims, epoint = bn.grand_product(tabs, chal)
    assert proof == b"".join(int(v).to_bytes(32, "big") for v in eproof)
    assert claims == eclaims and point == epoint
    # the final claims are the tables' multilinear extensions at the final point (the relation the verifier relies on)
    assert claims == [bn.mle_eval(t, point) for t in tabs]


def test_bn254_lasso_node_bit_exact(ctx):
    """The Lasso node of the n=1024 circuit over bn256::Fr: HIP (integer split / counters shared with the Goldilocks path, the rest
    over Fr) against the Python oracle fed with the C oracle's integer tables; the node input is the reference fixture's
    range-shifted lookup table (small integers, the same in both fields)."""
    bn = orclib.bn254()
    p = orclib.params(1024, 1)
    lasso_in, _, info = orclib.circuit_eval(p, orclib.fixture_inputs(1024, 1, 27))
    P = orclib.lasso_polys(p, lasso_in)
    skip = 3
    chal = bn.challenges(skip + bn.lasso_challenge_count(P["nu"]), orclib.keccak256)[skip:]
    eproof, er, eclaimed = bn.lasso_prove(P, chal)
    bfv = hg.BfvEncrypt.new(1024, 1)
    pk = bfv.setup(ctx)
    proof, r, claimed = ctx.lasso_prove_bn254(pk, [int(v) for v in lasso_in], skip)
    assert r == er and claimed == eclaimed
    assert proof == b"".join(int(v).to_bytes(32, "big") for v in eproof)
    with pytest.raises(hg.HgError):   # a value that is not range-shifted (>= 2^64) is refused, not silently truncated
        ctx.lasso_prove_bn254(pk, [int(v) for v in lasso_in[:-1]] + [bn.R - 1], skip)


def test_bn254_lasso_node_full_size_accepted_by_the_verifier(ctx):
    """BASELINE config 5 shape: the Lasso node of n=32768 k=16 (nu = 21, 25 memories) over bn256::Fr. Too large for the
    Python prover oracle; the proof is checked by the oracle's verifier (round consistency of every sum-check, layer
    chaining, the memory-checking hash relations at both grand-product points) and a tampered proof is rejected."""
    import time
    bn = orclib.bn254()
    bfv = hg.BfvEncrypt.new(32768, 16)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 5)
    lasso_in = pk.circuit_eval(w)[0]
    t0 = time.time()
    proof, r, claimed = ctx.lasso_prove_bn254(pk, [int(v) for v in lasso_in], 0, cap=1 << 24)
    print("hg_lasso_prove_bn254 n=32768 k=16: %.1f ms, %d proof bytes" % ((time.time() - t0) * 1e3, len(proof)))
    elems = [int.from_bytes(proof[i:i + 32], "big") for i in range(0, len(proof), 32)]
    mems, _ = orclib.lasso_layout(orclib.params(32768, 16))
    mem_dim = [int(m.split("@")[1]) for m in mems]
    def cutoff(name):
        if name == "full": return 65536
        b = int(name.split("_")[1])
        return (1 << ((b.bit_length() - 1) % 16)) + b % 65536
    mem_cutoff = [cutoff(m.split("@")[0]) for m in mems]
    chal = bn.challenges(bn.lasso_challenge_count(21), orclib.keccak256)
    assert bn.lasso_verify(elems, 21, mem_dim, mem_cutoff, chal) == (r, claimed)
    bad = list(elems); bad[len(bad) // 3] = (bad[len(bad) // 3] + 1) % bn.R
    with pytest.raises(ValueError):
        bn.lasso_verify(bad, 21, mem_dim, mem_cutoff, chal)


# ---- BfvEncrypt::prove over bn256::Fr (hg_witness_from_json_bn254, hg_circuit_eval_bn254, hg_prove_bn254) -------------------
BN_FIXTURE = os.path.join(orclib.GOLDEN, "bn254_sk_enc_1024_1x27_65537.json")


def _elems(proof):
    return [int.from_bytes(proof[i:i + 32], "big") for i in range(0, len(proof), 32)]


def test_bn254_witness_generation_on_the_reference_fixture(ctx):
    """The reference's own bn254 witness (bfv-gkr/src/data/bn254): loader + Circuit::evaluate over Fr on the device against the
    Python oracle, and the circuit relation sum == ct0is on the device output."""
    import json
    G = orclib.bn254_gkr()
    bfv = hg.BfvEncrypt.new(1024, 1)
    pk = bfv.setup(ctx)
    w = hg.Witness.from_json_bn254(bfv.params, BN_FIXTURE)
    inputs, ct0is = G.layout_inputs(1024, 1, json.load(open(BN_FIXTURE)))
    Cc, lasso_in, _, sum_id = G.build_circuit(orclib.constants(1024, 1))
    vals = G.circuit_evaluate(Cc, inputs)
    got_sum = ctx.circuit_eval_bn254(pk, w, 0)
    assert got_sum == vals[sum_id] == ct0is
    assert ctx.circuit_eval_bn254(pk, w, 1) == vals[lasso_in]
    assert ctx.circuit_eval_bn254(pk, w, 2) == ct0is
    # the Goldilocks fixture of the same parameter set is the same integer witness: it proves over Fr as well
    w_gl = hg.Witness.from_json(bfv.params, os.path.join(orclib.GOLDEN, "sk_enc_1024_1x27_65537.json"))
    s_gl = ctx.circuit_eval_bn254(pk, w_gl, 0)
    assert s_gl == ctx.circuit_eval_bn254(pk, w_gl, 2)


def test_bn254_prove_bit_exact_on_the_reference_fixture(ctx):
    """hg_prove_bn254 on the reference's bn254 fixture (n=1024): every proof element equals the Python oracle's; the oracle's
    verifier accepts it and rejects a tampered copy."""
    import json
    G, bn = orclib.bn254_gkr(), orclib.bn254()
    c = orclib.constants(1024, 1)
    inputs, ct0is = G.layout_inputs(1024, 1, json.load(open(BN_FIXTURE)))
    chal = bn.challenges(3000, orclib.keccak256)
    prove_fn, verify_fn = orclib.bn254_lasso_fns(orclib.params(1024, 1))
    trace = []
    expect, _ = G.prove(c, inputs, ct0is, chal, prove_fn, trace)
    bfv = hg.BfvEncrypt.new(1024, 1)
    pk = bfv.setup(ctx)
    w = hg.Witness.from_json_bn254(bfv.params, BN_FIXTURE)
    proof, wms, pms = ctx.prove_bn254(pk, w)
    got = _elems(proof)
    if got != expect:
        first = next(i for i, (a, b) in enumerate(zip(got, expect)) if a != b) if len(got) == len(expect) else min(len(got), len(expect))
        where = [t for t in trace if t[0] <= first][-1]
        pytest.fail("proof differs at element %d (%s, starts at %d); lengths %d / %d" % (first, where[1], where[0], len(got), len(expect)))
    import hashlib
    gold = json.load(open(os.path.join(orclib.GOLDEN, "oracle_proof_digests.json")))["bn254_1024_1"]
    assert len(proof) == gold["bytes"] and hashlib.sha256(proof).hexdigest() == gold["sha256"]
    assert G.verify(c, inputs, ct0is, got, chal, verify_fn)
    assert hg.verify_bn254(pk, w, proof) == (True, "")              # the product's own host verifier over Fr
    bad = list(got)
    bad[len(bad) // 2] = (bad[len(bad) // 2] + 1) % G.R
    with pytest.raises(ValueError):
        G.verify(c, inputs, ct0is, bad, chal, verify_fn)
    assert not hg.verify_bn254(pk, w, b"".join(v.to_bytes(32, "big") for v in bad))[0]


@pytest.mark.parametrize("n,k,bits", BN_FIX)
def test_bn254_prove_bit_exact_on_every_reference_fixture(ctx, n, k, bits):
    """hg_prove_bn254 on every bn254 witness the reference holds (bfv-gkr/src/data/bn254/): loader + device circuit evaluation
    (sum == ct0is), every proof byte equal to the C++ Fr oracle's (orcbn_prove; its digest is pinned in tests/golden), the
    oracle's verifier and the product's host verifier accept it, a flipped byte and another fixture's witness are rejected."""
    import hashlib, json
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.from_json_bn254(bfv.params, os.path.join(orclib.GOLDEN, f"bn254_sk_enc_{n}_{k}x{bits}_65537.json"))
    inp = orclib.bn254_fixture_inputs(n, k, bits)
    for f in ("s", "e", "k1", "ais", "r1is", "r2is", "ct0is"):
        assert (w.arrays()[f] == inp.d[f]).all(), f
    assert ctx.circuit_eval_bn254(pk, w, 0) == ctx.circuit_eval_bn254(pk, w, 2)          # sum node == ct0is, over Fr, on the device
    proof, wms, pms = ctx.prove_bn254(pk, w)
    p = orclib.params(n, k)
    ref, _ = orclib.prove_f("bn254", p, inp, threads=8)
    assert proof == ref, _first_diff(proof, ref)
    gold = json.load(open(os.path.join(orclib.GOLDEN, "oracle_proof_digests.json")))[f"bn254_{n}_{k}"]
    assert len(proof) == gold["bytes"] and hashlib.sha256(proof).hexdigest() == gold["sha256"]
    ok, err = orclib.verify_f("bn254", p, inp, proof, threads=8)
    assert ok, err
    assert hg.verify_bn254(pk, w, proof) == (True, "")
    bad = bytearray(proof)
    bad[len(bad) // 2 + 31] ^= 1
    assert not hg.verify_bn254(pk, w, bytes(bad))[0]
    # the Goldilocks fixture of the same parameter set is another sample: its public inputs do not fit this proof
    w_gl = hg.Witness.from_json(bfv.params, os.path.join(orclib.GOLDEN, f"sk_enc_{n}_{k}x{bits}_65537.json"))
    assert not hg.verify_bn254(pk, w_gl, proof)[0]
    pk.free()


def test_bn254_prove_synthetic_k2_accepted_by_the_oracle_verifier(ctx):
    """n=4096 k=2 synthetic witness (alpha claims on the shared inputs, two CRT components): the GPU proof over Fr passes the Python
    oracle's verifier, including the final input-claim checks against the witness."""
    G, bn = orclib.bn254_gkr(), orclib.bn254()
    n, k = 4096, 2
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 77)
    d = w.arrays()
    SZ = 2 * n
    lift = G.lift_signed
    inputs = [[lift(v) for v in d[f]] for f in ("s", "e", "k1")]
    inputs += [[lift(v) for v in d["ais"][z * SZ:(z + 1) * SZ]] for z in range(k)]
    inputs += [[lift(v) for v in d["r1is"][z * SZ:(z + 1) * SZ]] for z in range(k)]
    inputs.append([lift(v) for v in d["r2is"]])
    ct0is = [lift(v) for v in d["ct0is"]]
    proof, wms, pms = ctx.prove_bn254(pk, w)
    p = orclib.params(n, k)
    lasso_in = [int(v) for v in ctx.circuit_eval_bn254(pk, w, 1)]
    layout = orclib.lasso_polys(p, np.array(lasso_in, dtype=np.uint64))
    _, verify_fn = orclib.bn254_lasso_fns(p)
    chal = bn.challenges(4000, orclib.keccak256)
    assert G.verify(orclib.constants(n, k), inputs, ct0is, _elems(proof), chal, lambda e, c: verify_fn(e, c, layout))
    assert hg.verify_bn254(pk, w, proof) == (True, "")
    assert not hg.verify_bn254(pk, hg.Witness.synthetic(bfv.params, 78), proof)[0]   # another witness: input claims fail


@pytest.mark.parametrize("n,k", [(1024, 1), (2048, 1), (4096, 2), (8192, 4), (16384, 8), (32768, 16)])
def test_bn254_prove_every_element_equals_the_cpp_oracle(ctx, n, k):
    """hg_prove_bn254 against the C++ oracle compiled over bn256::Fr (oracle/fr.hpp, orcbn_prove): EVERY proof element, at every
    built-in parameter set up to BASELINE config 5 (n=32768 k=16) - several claims per node, alpha combinations, k CRT
    components. The oracle's verifier then accepts the bytes and rejects a flipped one.
    [REF bfv-gkr/src/sk_encryption_circuit.rs:614-626: the bn254 test family]"""
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 5 + n)
    proof, wms, pms = ctx.prove_bn254(pk, w, cap=1 << 25)
    p = orclib.params(n, k)
    inp = orclib.Inputs(w.arrays())
    threads = min(64, os.cpu_count() or 8)
    ref, tm = orclib.prove_f("bn254", p, inp, threads=threads)
    print("bn254 n=%d k=%d: hg_prove_bn254 %.1f ms, oracle %.0f ms on %d threads, %d elements" % (n, k, pms, tm[1], threads, len(ref) // 32))
    assert proof == ref, _first_diff(proof, ref)
    ok, err = orclib.verify_f("bn254", p, inp, proof, threads=threads)
    assert ok, err
    bad = bytearray(proof)
    bad[len(bad) // 2 + 31] ^= 1
    assert not orclib.verify_f("bn254", p, inp, bytes(bad), threads=threads)[0]
    assert hg.verify_bn254(pk, w, proof) == (True, "")
    pk.free()


def test_bn254_prove_config5_shape_accepted_by_the_host_verifier(ctx):
    """BASELINE config 5 (n=32768 k=16) over bn256::Fr: too large for the Python oracle; the proof must be accepted by
    hg_verify_bn254 (itself cross-checked against the oracle at n=1024: test_bn254_host_verifier_agrees_with_the_oracle),
    a tampered copy rejected, and it has as many elements as the Goldilocks proof of the same parameter set."""
    bfv = hg.BfvEncrypt.new(32768, 16)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 5)
    proof, wms, pms = ctx.prove_bn254(pk, w, cap=1 << 25)
    print("hg_prove_bn254 n=32768 k=16: witness %.1f ms, prove %.1f ms, %d bytes" % (wms, pms, len(proof)))
    gl, _ = bfv.prove(ctx, pk, w)
    assert len(proof) // 32 == len(gl) // 16
    assert hg.verify_bn254(pk, w, proof) == (True, "")
    bad = bytearray(proof)
    bad[len(bad) // 2 + 31] ^= 1
    assert not hg.verify_bn254(pk, w, bytes(bad))[0]


# ---- invalid witnesses: the prover must behave exactly like the CPU restatement, and every verifier must reject -------------
def _tampered_witness(bfv, kind, field_p):
    """A synthetic n=1024 witness made invalid. kind: "range" = an error coefficient outside its range-check bound (the Lasso
    lookup then returns the wrong sub-table value); "relation" = one ct0 coefficient changed (the circuit relation fails);
    "huge" = a secret coefficient that is not a small integer at all."""
    d = {f: a.copy() for f, a in hg.Witness.synthetic(bfv.params, 4242).arrays().items()}
    if kind == "range":
        d["e"][5] = 1000          # e_bound = 19
    elif kind == "relation":
        d["ct0is"][7] = (int(d["ct0is"][7]) + 1) % field_p
    else:
        d["s"][3] = 1 << 40       # s_bound = 1
    return hg.Witness.from_arrays(bfv.params, d)


@pytest.mark.parametrize("kind", ["range", "relation", "huge"])
def test_invalid_witness_same_transcript_as_the_oracle_and_rejected(ctx, kind):
    bfv = hg.BfvEncrypt.new(1024, 1)
    pk = bfv.setup(ctx)
    w = _tampered_witness(bfv, kind, P)
    proof, _ = bfv.prove(ctx, pk, w)             # no crash, no special casing: the reference prover does not validate either
    p = orclib.params(1024, 1)
    inp = orclib.Inputs(w.arrays())
    ref, _ = orclib.prove(p, inp, threads=4)
    assert proof == ref
    ok_o, _ = orclib.verify(p, inp, proof)
    ok_p, why = hg.verify(pk, w, proof)
    assert not ok_o and not ok_p and why
    pk.free()


@pytest.mark.parametrize("kind", ["range", "relation"])
def test_bn254_invalid_witness_rejected_by_both_verifiers(ctx, kind):
    G, bn = orclib.bn254_gkr(), orclib.bn254()
    n, k = 1024, 1
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = _tampered_witness(bfv, kind, P)
    proof, _, _ = ctx.prove_bn254(pk, w)
    assert not hg.verify_bn254(pk, w, proof)[0]
    d = w.arrays()
    lift = G.lift_signed
    inputs = [[lift(v) for v in d[f]] for f in ("s", "e", "k1", "ais", "r1is", "r2is")]
    ct0is = [lift(v) for v in d["ct0is"]]
    p = orclib.params(n, k)
    prove_fn, verify_fn = orclib.bn254_lasso_fns(p)
    chal = bn.challenges(3000, orclib.keccak256)
    expect, _ = G.prove(orclib.constants(n, k), inputs, ct0is, chal, prove_fn)
    assert _elems(proof) == expect               # bit-exact on an invalid witness too
    with pytest.raises(ValueError):
        G.verify(orclib.constants(n, k), inputs, ct0is, expect, chal, verify_fn)


@pytest.mark.parametrize("n,k", [(2048, 1), (8192, 4), (16384, 8)])
def test_bn254_prove_other_parameter_sets_accepted_by_the_host_verifier(ctx, n, k):
    """The remaining built-in parameter sets over bn256::Fr (different table lengths, chunk counts and claim multiplicities): the
    proof has the Goldilocks proof's element count, the host verifier accepts it, rejects a flipped byte and another witness."""
    bfv = hg.BfvEncrypt.new(n, k)
    pk = bfv.setup(ctx)
    w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
    proof, _, _ = ctx.prove_bn254(pk, w, cap=1 << 25)
    gl, _ = bfv.prove(ctx, pk, w)
    assert len(proof) // 32 == len(gl) // 16
    assert hg.verify_bn254(pk, w, proof) == (True, "")
    bad = bytearray(proof)
    bad[(len(bad) // 3) | 31] ^= 4
    assert not hg.verify_bn254(pk, w, bytes(bad))[0]
    assert not hg.verify_bn254(pk, hg.Witness.synthetic(bfv.params, 1), proof)[0]
    proof2, _, _ = ctx.prove_bn254(pk, w, cap=1 << 25)   # determinism + arena reuse
    assert proof2 == proof
    pk.free()


@pytest.mark.gpu
def test_eq_factored_rounds_are_the_path_that_runs():
    """The eq-factored PRODSUM rounds must actually be taken at the headline size - otherwise HG_NO_PS_EQ=1 would compare
    the materialised form with itself: 14 of the 81 first-wave node reductions (65 Libra phase-1 / FFT reductions and, since round 6,
    the 16 mul nodes' phase 2 beside them), most of the table entries."""
    import subprocess, sys
    from hglib import ROOT
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as entry\n"
        "hg = entry.load_package()\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)\n"
        "w = hg.Witness.synthetic(bfv.params, 5); v = hg.witness_gen(ctx, pk, w); out = hg.ProofBuffer()\n"
        "hg.prove_resident(ctx, pk, v, out)\n"
        "print('RAN')\n"
    ) % (ROOT,)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HG_DEBUG="eq"), cwd=ROOT)
    assert r.returncode == 0 and "RAN" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    lines = [l.split() for l in r.stderr.splitlines() if l.startswith("[hg eq]")]
    assert lines and int(lines[0][2]) == 14 and int(lines[0][4]) == 81, r.stderr[-2000:]
    assert int(lines[0][9]) * 10 > int(lines[0][11]) * 7, r.stderr[-2000:]     # most of the entries (13.4 M of 18.3 M)


@pytest.mark.gpu
def test_slot_form_is_the_path_that_runs():
    """The slot form of grand product #1 must actually be taken where it applies - otherwise the switch tests above would
    compare the memory form with itself. HG_DEBUG=slots makes the library report the layers it adopted; HG_TIMES=bn the bn254 prove's."""
    import subprocess, sys
    from hglib import ROOT
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as entry\n"
        "hg = entry.load_package()\n"
        "ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(4096, 2); pk = bfv.setup(ctx)\n"
        "w = hg.Witness.synthetic(bfv.params, 5); v = hg.witness_gen(ctx, pk, w); out = hg.ProofBuffer()\n"
        "hg.prove_resident(ctx, pk, v, out)\n"
        "ctx.prove_bn254(pk, w, cap=1 << 24)\n"
        "print('RAN')\n"
    ) % (ROOT,)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HG_DEBUG="slots", HG_TIMES="bn"), cwd=ROOT)
    assert r.returncode == 0 and "RAN" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    adopted = [l for l in r.stderr.splitlines() if l.startswith("[hg slots] adopted:")]
    assert adopted and int(adopted[0].split()[3]) >= 2, r.stderr[-2000:]          # the top layer and at least one below it
    bn = [l for l in r.stderr.splitlines() if "slot rows for" in l]
    assert bn and int(bn[0].split("read rows:")[1].split()[0]) < int(bn[0].split("slot rows for")[1].split()[0]), r.stderr[-2000:]
    assert int(bn[0].split("rows;")[1].split()[0]) >= 1, r.stderr[-2000:]       # ... and in the bn254 prove
