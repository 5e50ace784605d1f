#!/usr/bin/env python3
"""Runs hg_sumcheck_bn254 (grand-product shape) on random tables; profiled with rocprofv3 --kernel-trace --stats to get the
duration of k_bn_round. usage: bn254_bench.py [nv] [ntab]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
hg = entry.load_package()
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ntab = int(sys.argv[2]) if len(sys.argv) > 2 else 50
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
rng = np.random.default_rng(5)
ctx = hg.Context(0)
# random limbs with the top limb masked below the modulus' top limb: canonical values
def table(n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    return a.reshape(-1)
import ctypes as C
packed = [table(1 << nv) for _ in range(ntab)]
u64p = C.POINTER(C.c_uint64)
ptrs = (u64p * ntab)(*[p.ctypes.data_as(u64p) for p in packed])
pw = table(ntab // 2); claim = table(1)
d = 3
msgs = np.zeros(nv * (d + 1) * 4, dtype=np.uint64); point = np.zeros(nv * 4, dtype=np.uint64)
evals = np.zeros(ntab * 4, dtype=np.uint64); sums = np.zeros(nv * d * 4, dtype=np.uint64)
P = lambda a: a.ctypes.data_as(u64p)
for _ in range(3):
    rc = hg.lib().hg_sumcheck_bn254(ctx.h, 1, nv, ntab, ptrs, P(pw), ntab // 2, P(claim), 0, P(msgs), P(point), P(evals), P(sums))
    assert rc == 0
print("ok", int(msgs[0]))
