#!/usr/bin/env python3
"""The two-table collation sum-check alone (hg_sumcheck, kind 0, two base tables of 2^21 entries), three times: run under
`rocprofv3 --kernel-trace --stats` for the isolated kernel durations. usage: col_bench.py [log2 n]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
hg = entry.load_package()
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 21
P = (1 << 64) - (1 << 32) + 1
rng = np.random.default_rng(1)
tabs = [rng.integers(0, P, 1 << nv, dtype=np.uint64) for _ in range(2)]
pw = np.array([[1, 0], [1, 0]], dtype=np.uint64)
claim = np.array([3, 4], dtype=np.uint64)
ctx = hg.Context(0)
for _ in range(3):
    ctx.sumcheck(0, tabs, [True, True], pw, claim, 5)
