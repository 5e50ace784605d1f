#!/usr/bin/env python3
"""Seven calls of hg_prove_shard_begin for one virtual rank (the last ones replay its launch graph): run under
`rocprofv3 --kernel-trace` and feed the CSV to trace_timeline.py. usage: shard_trace.py <world> <rank>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
hg = entry.load_package()
world, rank = int(sys.argv[1]), int(sys.argv[2])
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 32768); vals = hg.witness_gen(ctx, pk, w)
import time
for i in range(7):
    hg.prove_shard_begin(ctx, pk, vals, rank, world)
    time.sleep(0.01)
