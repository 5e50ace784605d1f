#!/usr/bin/env python3
"""Times hg_lasso_prove_bn254 (the Lasso node over bn256::Fr) at the BASELINE config-5 shape n=32768 k=16: the C call only
(input already packed), 3 runs. usage: bn254_lasso_bench.py [n k]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
hg = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = hg.Context(0)
bfv = hg.BfvEncrypt.new(n, k)
pk = bfv.setup(ctx)
w = hg.Witness.synthetic(bfv.params, 0x4752454330 + 5)
lasso_in = pk.circuit_eval(w)[0]
packed = np.zeros((lasso_in.size, 4), dtype=np.uint64)
packed[:, 0] = lasso_in
packed = packed.reshape(-1)
nu = lasso_in.size.bit_length() - 1
cap = 1 << 24
buf = (C.c_uint8 * cap)()
ln = C.c_size_t(0)
claim = np.zeros((nu + 1) * 4, dtype=np.uint64)
u64p = C.POINTER(C.c_uint64)
for i in range(3):
    t0 = time.perf_counter()
    rc = hg.lib().hg_lasso_prove_bn254(ctx.h, pk.h, packed.ctypes.data_as(u64p), 0, buf, cap, C.byref(ln), claim.ctypes.data_as(u64p))
    assert rc == 0
    print("hg_lasso_prove_bn254 n=%d k=%d (nu=%d): %.1f ms, %d proof bytes" % (n, k, nu, (time.perf_counter() - t0) * 1e3, ln.value))
