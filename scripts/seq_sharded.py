#!/usr/bin/env python3
"""hg_prove_resident_mode_sharded at n=32768 k=16 (or argv n k), mode 3, with 1, 2 and 4 ranks as threads of this process on ONE GPU
(hg_group_local): checks every rank's proof against the single-rank proof and prints the per-rank time and the number of
all-reduces. The ranks share a device, so the times show the cost of the scheme's bookkeeping (every rank still folds everything and
waits for the slowest rank once per round), not a multi-GPU speed-up. usage: seq_sharded.py [n k]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
hg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = 3
bfv = hg.BfvEncrypt.new(n, k)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
ctxs = [hg.Context(0) for _ in range(4)]
pks = [bfv.setup(c) for c in ctxs]
vals = [hg.witness_gen(c, pk, w) for c, pk in zip(ctxs, pks)]
out0 = hg.ProofBuffer()
for _ in range(2):
    ref = hg.prove_resident_mode(ctxs[0], pks[0], vals[0], out0, mode).bytes()
print("single rank: %.1f ms, %d mailbox round trips" % (out0.timings()["prove_ms"], int(out0.timings()["enqueue_ms"])))
for world in (1, 2, 4):
    for rep in range(2):
        group = hg.Group.local(world)
        res = [None] * world
        def run(r):
            out = hg.ProofBuffer()
            try:
                hg.prove_resident_mode_sharded(ctxs[r], pks[r], vals[r], out, mode, r, group)
                res[r] = (out.bytes(), out.timings())
            except Exception as e:
                print("rank", r, "failed:", e)
        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ts: t.start()
        for t in ts: t.join()
        print("  world %d rep %d:" % (world, rep), ["ok" if (x is not None and x[0] == ref) else ("none" if x is None else "DIFFERS") for x in res], flush=True)
        assert all(x is not None and x[0] == ref for x in res), "a rank's proof differs"
    print("world %d: per-rank prove ms %s, %d all-reduces per proof" % (world, ["%.1f" % x[1]["prove_ms"] for x in res], int(res[0][1]["replay_ms"])))
