"""Kernel-class breakdown of single virtual ranks of an 8-rank sharded proof (one GPU). usage: shard_prof.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
hg = entry.load_package()
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 32768); vals = hg.witness_gen(ctx, pk, w)
out = hg.ProofBuffer()
for _ in range(2): hg.prove_resident(ctx, pk, vals, out)
world = 8
for r in (0, 5, 7):
    for _ in range(2): hg.prove_shard_begin(ctx, pk, vals, r, world)
    ctx.profile(2); ctx.profile_reset()
    hg.prove_shard_begin(ctx, pk, vals, r, world)
    hg.prove_shard_finish(ctx, out)   # replays this rank's (partial) buffer: only here to collect the event timings
    ctx.profile(0)
    st = {s["name"]: round(s["total_ms"], 3) for s in ctx.profile_get() if s["launches"]}
    print("rank", r, "sum %.3f" % sum(st.values()), st)
