#!/usr/bin/env python3
"""Soak of the sequential prover (modes 1-3): many proves of rotating witnesses, every proof compared with the first proof of its
witness and mode (which the caller has compared with the oracle in the test suite). usage: seq_soak.py n k proves"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
hg = entry.load_package()
n, k, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(n, k); pk = bfv.setup(ctx)
ws = [hg.Witness.synthetic(bfv.params, 900 + i) for i in range(3)]
vals = [hg.witness_gen(ctx, pk, w) for w in ws]
out = hg.ProofBuffer()
ref = {}
bad = 0
t0 = time.perf_counter()
for i in range(N):
    j, mode = i % 3, (1, 3, 2, 3)[i % 4]
    hg.prove_resident_mode(ctx, pk, vals[j], out, mode)
    b = out.bytes()
    if (j, mode) not in ref:
        ref[(j, mode)] = b
        ok, why = hg.verify(pk, ws[j], b, mode=mode)
        assert ok, why
    elif b != ref[(j, mode)]:
        bad += 1
        print("MISMATCH at prove %d (witness %d, mode %d)" % (i, j, mode), flush=True)
    if i % 7 == 0:   # the fast path in between (shares streams, arena, result buffer)
        hg.prove_resident(ctx, pk, vals[j], out)
print("n=%d k=%d: %d proves, %d mismatches, %.1f s" % (n, k, N, bad, time.perf_counter() - t0))
sys.exit(1 if bad else 0)
