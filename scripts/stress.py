"""Stability loop on one GPU: 1500 resident proves (the cached launch graph dropped and re-captured every 300), a BN254 prove every
100, a four-virtual-rank sharded proof every 250, EVERY proof compared with the first; prints the free GPU memory before and after."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as entry
hg = entry.load_package()
def free_mb(): return torch.cuda.mem_get_info(0)[0] / 2**20
n, k = 32768, 16
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(n, k); pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n); vals = hg.witness_gen(ctx, pk, w); out = hg.ProofBuffer()
for _ in range(5): hg.prove_resident(ctx, pk, vals, out)
ref = out.bytes(); ctx.prove_bn254(pk, w, cap=1 << 25); refb = ctx.prove_bn254(pk, w, cap=1 << 25)[0]
f0 = free_mb(); t0 = time.time()
import numpy as np
def sharded(world):   # one proof as `world` virtual ranks on this GPU: partial result buffers combined, transcript replayed once
    parts = [np.array(hg.prove_shard_begin(ctx, pk, vals, r, world), copy=True) for r in range(world)]
    hg.prove_shard_combine(ctx, np.stack(parts), world)
    return hg.prove_shard_finish(ctx, out).bytes()
for i in range(1500):
    assert hg.prove_resident(ctx, pk, vals, out).bytes() == ref, i   # (graph replays, except right after an invalidation)
    if i % 300 == 0:
        ctx.set_option("one_stream", i % 600 == 0)   # drops and re-captures the launch graph now and then
    if i % 250 == 7:
        assert sharded(4) == ref, i                  # other shares of the same key in between: the cached graph must not survive them
    if i % 100 == 0:
        assert ctx.prove_bn254(pk, w, cap=1 << 25)[0] == refb
ctx.set_option("one_stream", 0)
assert hg.prove_resident(ctx, pk, vals, out).bytes() == ref
print("1500 proves + 15 bn254 proves in %.1f s; free GPU memory %.0f -> %.0f MiB; last prove %.3f ms" % (time.time() - t0, f0, free_mb(), out.timings()["prove_ms"]))
