#!/usr/bin/env python3
"""Busy time vs. span of the last prove in a rocprofv3 kernel-trace CSV: how much of the GPU time is inter-kernel gap.
usage: trace_gaps.py <dir> [n_proves]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# split into proves by the largest gaps
gaps = sorted(((rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)), reverse=True)[: n]
cut = sorted(i for _, i in gaps)[-1] if n > 1 else -1
last = rows[cut + 1:]
span = last[-1][1] - last[0][0]
busy = sum(e - s for s, e, _ in last)
gap_hist = {}
tot_gap = 0
for i in range(len(last) - 1):
    g = last[i + 1][0] - last[i][1]
    tot_gap += max(g, 0)
print("last prove: %d dispatches, span %.3f ms, busy %.3f ms, gaps %.3f ms (avg %.2f us)" % (len(last), span / 1e6, busy / 1e6, tot_gap / 1e6, tot_gap / 1e3 / max(1, len(last) - 1)))
small = [(e - s) / 1e3 for s, e, _ in last if e - s < 10000]
print("dispatches shorter than 10 us: %d, total %.3f ms" % (len(small), sum(small) / 1e3))
