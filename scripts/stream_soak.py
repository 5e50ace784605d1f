#!/usr/bin/env python3
"""Soak of hg_prove_stream: runs of random length and order over five witnesses, every proof compared with hg_prove of its witness.
usage: stream_soak.py n k runs"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
hg = entry.load_package()
n, k, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(n, k); pk = bfv.setup(ctx)
ws = [hg.Witness.synthetic(bfv.params, 300 + i) for i in range(5)]
refs = [bfv.prove(ctx, pk, w)[0] for w in ws]
import random
random.seed(1)
bad = 0
t0 = time.perf_counter(); np_ = 0
for r in range(R):
    order = [random.randrange(5) for _ in range(random.randrange(1, 12))]
    proofs, tm = bfv.prove_stream(ctx, pk, [ws[i] for i in order])
    np_ += len(order)
    for j, i in enumerate(order):
        if proofs[j] != refs[i]:
            bad += 1; print("MISMATCH run %d pos %d" % (r, j), flush=True)
    if r % 5 == 0:
        assert bfv.prove(ctx, pk, ws[r % 5])[0] == refs[r % 5]
print("n=%d: %d runs, %d proofs, %d mismatches, %.1f s" % (n, R, np_, bad, time.perf_counter() - t0))
sys.exit(1 if bad else 0)
