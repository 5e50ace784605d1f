import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
hg = entry.load_package()
ctx = hg.Context(0); bfv = hg.BfvEncrypt.new(32768, 16); pk = bfv.setup(ctx)
ws = [hg.Witness.synthetic(bfv.params, 7000 + i) for i in range(4)]
vals = [hg.witness_gen(ctx, pk, w) for w in ws]
out = hg.ProofBuffer()
refs = []
for v in vals:
    for _ in range(3): hg.prove_resident(ctx, pk, v, out)   # walk, walk, capture
    refs.append(out.bytes())
assert len(set(refs)) == 4
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
import time
t0 = time.time()
for it in range(N):
    j = (it * 5 + it // 7) % 4
    if hg.prove_resident(ctx, pk, vals[j], out).bytes() != refs[j]: bad += 1
print("early-replay soak: %d graph proves over 4 witnesses in rotation, %d mismatches, %.1f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
