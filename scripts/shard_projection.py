import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
hg = entry.load_package()
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 5); out = hg.ProofBuffer()
fv = hg.witness_gen(ctx, pk, w); ref = hg.prove_resident(ctx, pk, fv, out).bytes(); fv.free()
for wsz in (2, 4, 8):
    svals = [hg.witness_gen_shard(ctx, pk, w, r, wsz) for r in range(wsz)]
    per, parts = [], []
    for r in range(wsz):
        ts = []
        for i in range(7):
            t0 = time.perf_counter(); part = hg.prove_shard_begin(ctx, pk, svals[r], r, wsz); ts.append((time.perf_counter() - t0) * 1e3)
        per.append(sorted(ts[3:])[1]); parts.append(part.copy())
    g = np.stack(parts)
    t0 = time.perf_counter(); hg.prove_shard_combine(ctx, g, wsz); t1 = time.perf_counter(); pb = hg.prove_shard_finish(ctx, out); t2 = time.perf_counter()
    print(wsz, "per rank", [round(x, 3) for x in per], "combine %.3f ms finish %.3f ms (replay %.3f)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, out.timings()["replay_ms"]), "same" if pb.bytes() == ref else "DIFFERENT")
    for v in svals: v.free()
