#!/usr/bin/env python3
"""Per-rank cost of the sharded proof, measured on ONE GPU: every virtual rank of a world-rank job runs its share
(prove_shard_begin) in turn; prints wall ms per rank (begin = enqueue + the rank's only stream sync), then checks
that the combined buffers replay to the unsharded proof. usage: shard_times.py [n k] [worlds...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
hg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
worlds = [int(a) for a in sys.argv[3:]] or [2, 4, 8]
ctx = hg.Context(0)
bfv = hg.BfvEncrypt.new(n, k)
pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
vals = hg.witness_gen(ctx, pk, w)
out = hg.ProofBuffer()
for _ in range(3):
    ref = hg.prove_resident(ctx, pk, vals, out).bytes()
t0 = time.perf_counter(); hg.prove_resident(ctx, pk, vals, out); t1 = time.perf_counter()
print("single GPU: %.2f ms wall, gpu %.2f ms" % ((t1 - t0) * 1e3, out.timings()["gpu_ms"]))
for world in worlds:
    times, parts = [], []
    for rep in range(2):
        times, parts = [], []
        for r in range(world):
            t0 = time.perf_counter()
            p = hg.prove_shard_begin(ctx, pk, vals, r, world).copy()
            times.append((time.perf_counter() - t0) * 1e3)
            parts.append(p)
    hg.prove_shard_combine(ctx, np.stack(parts), world)
    t0 = time.perf_counter()
    got = hg.prove_shard_finish(ctx, out).bytes()
    fin = (time.perf_counter() - t0) * 1e3
    tc = time.perf_counter(); hg.prove_shard_combine(ctx, np.stack(parts), world); tc = (time.perf_counter() - tc) * 1e3
    print("world %d: per-rank begin ms %s  max %.2f  finish %.2f (replay %.2f)  combine %.2f  ok=%s" % (world, " ".join("%.2f" % t for t in times), max(times), fin, out.timings()["replay_ms"], tc, got == ref))
