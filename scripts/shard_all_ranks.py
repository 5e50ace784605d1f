#!/usr/bin/env python3
"""Per-rank wall time of hg_prove_shard_begin (graph replay, median of 5) for EVERY virtual rank of a world. usage: shard_all_ranks.py [world]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
hg = entry.load_package()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 32768); vals = hg.witness_gen(ctx, pk, w)
res = []
for r in range(world):
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); hg.prove_shard_begin(ctx, pk, vals, r, world); ts.append((time.perf_counter() - t0) * 1e3)
    res.append(statistics.median(ts[3:]))
print("world %d: per-rank ms (graph replay): %s; max %.3f" % (world, " ".join("%.3f" % t for t in res), max(res)))
