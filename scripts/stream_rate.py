#!/usr/bin/env python3
"""Per-proof time of hg_prove_stream over a run of witnesses (the bench line's end_to_end.pipelined_arrays_to_proof_ms) against
hg_prove one by one. usage: stream_rate.py [n k run reps]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
hg = entry.load_package()
n, k, run, reps = (int(a) for a in (sys.argv[1:5] + ["32768", "16", "16", "5"][len(sys.argv) - 1:]))
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(n, k); pk = bfv.setup(ctx)
ws = [hg.Witness.synthetic(bfv.params, 900 + i) for i in range(4)]
refs = [bfv.prove(ctx, pk, w)[0] for w in ws]
for _ in range(3):
    for w in ws: bfv.prove(ctx, pk, w)
one = []
for r in range(reps):
    t0 = time.perf_counter()
    for i in range(run): bfv.prove(ctx, pk, ws[i % 4])
    one.append((time.perf_counter() - t0) / run * 1e3)
order = [ws[i % 4] for i in range(run)]
for _ in range(3): bfv.prove_stream(ctx, pk, order)
pipe = []
for r in range(reps):
    t0 = time.perf_counter()
    proofs, tm = bfv.prove_stream(ctx, pk, order)
    pipe.append((time.perf_counter() - t0) / run * 1e3)
    assert all(proofs[i] == refs[i % 4] for i in range(run))
print("hg_prove %.3f ms per proof, hg_prove_stream %.3f ms per proof (medians of %d runs of %d)" % (statistics.median(one), statistics.median(pipe), reps, run))
