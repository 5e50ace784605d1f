#!/usr/bin/env python3
"""Times hg_prove_bn254 (BfvEncrypt::prove over bn256::Fr) at the BASELINE config-5 shape n=32768 k=16 on a synthetic witness,
3 runs, and checks the proof has as many elements as the Goldilocks proof of the same parameter set (same protocol, E = F).
usage: bn254_prove_bench.py [n k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
hg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = hg.Context(0)
bfv = hg.BfvEncrypt.new(n, k)
pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 5)
gl_len = len(bfv.prove(ctx, pk, w)[0])
for i in range(3):
    t0 = time.perf_counter()
    proof, wms, pms = ctx.prove_bn254(pk, w, cap=1 << 25)
    wall = (time.perf_counter() - t0) * 1e3
    print("hg_prove_bn254 n=%d k=%d: witness %.1f ms, prove %.1f ms (call %.1f ms), %d elements (Goldilocks proof: %d)"
          % (n, k, wms, pms, wall, len(proof) // 32, gl_len // 16))
    assert len(proof) // 32 == gl_len // 16
