#!/usr/bin/env python3
"""The drop-in caller's cold path step by step (fresh context, hg_setup, hg_warmup, then one hg_prove per NEW witness), with the library's
own stage timings next to the wall clock of each call. usage: cold_path.py [n k] [nowarm]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
hg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
warm = "nowarm" not in sys.argv
t0 = time.perf_counter(); ctx = hg.Context(0); t1 = time.perf_counter()
bfv = hg.BfvEncrypt.new(n, k)
pk = bfv.setup(ctx); t2 = time.perf_counter()
wu = bfv.warmup(ctx, pk) if warm else 0
t3 = time.perf_counter()
print("hg_create %.1f ms, hg_setup %.1f ms, hg_warmup %.1f ms (library: %.1f)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, wu))
for i in range(5):
    w = hg.Witness.synthetic(bfv.params, 900 + i)
    t = time.perf_counter()
    proof, tm = bfv.prove(ctx, pk, w, cap=1 << 20)
    wall = (time.perf_counter() - t) * 1e3
    print("prove %d: wall %.3f ms; library total %.3f = upload %.3f + witness %.3f + prove %.3f (gpu %.3f, enqueue %.3f, sync %.3f, replay %.3f)" %
          (i, wall, tm["total_ms"], tm["upload_ms"], tm["witness_ms"], tm["prove_ms"], tm["gpu_ms"], tm["enqueue_ms"], tm["sync_ms"], tm["replay_ms"]))
