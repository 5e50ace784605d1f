#!/usr/bin/env python3
"""Runs W warm-up + K resident proves of one config (default n=32768 k=16) — the command profiled by rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
hg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = hg.Context(0)
bfv = hg.BfvEncrypt.new(n, k)
pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + n)
vals = hg.witness_gen(ctx, pk, w)
out = hg.ProofBuffer()
first = None
for _ in range(steps):
    hg.prove_resident(ctx, pk, vals, out)
    first = first or out.bytes()
    assert out.bytes() == first, "proof changed between runs"
    print(out.timings())
if os.environ.get("HG_PROOF_OUT"):
    open(os.environ["HG_PROOF_OUT"], "wb").write(out.bytes())
