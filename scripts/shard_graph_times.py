#!/usr/bin/env python3
"""Per-rank time of a sharded proof when the rank's share is replayed from its launch graph: seven calls of hg_prove_shard_begin for
the first and the last rank of each world size (walk, walk, capture, four replays). usage: shard_graph_times.py [worlds...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
hg = entry.load_package()
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 32768); vals = hg.witness_gen(ctx, pk, w)
for world in [int(a) for a in sys.argv[1:]] or [8, 4, 2]:
    for r in (0, world - 1):
        ts = []
        for i in range(7):
            t0 = time.perf_counter(); hg.prove_shard_begin(ctx, pk, vals, r, world); ts.append((time.perf_counter() - t0) * 1e3)
        print("world %d rank %d: begin ms per call (walk, walk, capture, replays): %s" % (world, r, " ".join("%.2f" % t for t in ts)))
