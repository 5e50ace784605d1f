#!/usr/bin/env python3
"""Throughput of independent proofs on ONE GPU when more than one is in flight: C contexts (each with its own streams, result
buffers, launch graphs), one host thread per context, every thread proving its own resident witnesses in a loop.
usage: two_contexts.py [n k [contexts [proofs_per_thread]]]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
hg = entry.load_package()
n, k = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 16)
C = int(sys.argv[3]) if len(sys.argv) > 3 else 2
K = int(sys.argv[4]) if len(sys.argv) > 4 else 40
bfv = hg.BfvEncrypt.new(n, k)
ws = [hg.Witness.synthetic(bfv.params, 0x4752454330 + i) for i in range(2)]
ctxs, pks, vals, outs, refs = [], [], [], [], []
for c in range(C):
    ctx = hg.Context(0); pk = bfv.setup(ctx)
    v = [hg.witness_gen(ctx, pk, w) for w in ws]
    out = hg.ProofBuffer()
    ref = []
    for j in range(2):
        for i in range(4):
            hg.prove_resident(ctx, pk, v[j], out)
        ref.append(out.bytes())
    ctxs.append(ctx); pks.append(pk); vals.append(v); outs.append(out); refs.append(ref)
assert all(r == refs[0] for r in refs)
def run(c, count, bad):
    for i in range(count):
        hg.prove_resident(ctxs[c], pks[c], vals[c][i & 1], outs[c])
        if outs[c].bytes() != refs[c][i & 1]:
            bad.append((c, i))
for nthreads in range(1, C + 1):
    bad = []
    th = [threading.Thread(target=run, args=(c, K, bad)) for c in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = (time.perf_counter() - t0) * 1e3
    print("%d proofs in flight: %d proofs in %.1f ms = %.3f ms per proof%s" % (nthreads, nthreads * K, dt, dt / (nthreads * K), "  MISMATCHES %r" % bad[:3] if bad else ""))
