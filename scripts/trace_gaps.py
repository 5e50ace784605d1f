#!/usr/bin/env python3
"""Busy time vs. span of the last prove in a rocprofv3 kernel-trace CSV: how much of the GPU time is inter-kernel gap.
usage: trace_gaps.py <dir> [n_proves]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the last prove = everything after the last inter-kernel gap longer than 150 us (host replay between proves)
cut = max([i for i in range(len(rows) - 1) if rows[i + 1][0] - rows[i][1] > 150000] or [-1])
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
big_gaps = sorted(((last[i + 1][0] - last[i][1]) / 1e3, last[i][2][:50], last[i + 1][2][:50]) for i in range(len(last) - 1))[-8:]
for g, a, b in big_gaps: print("gap %.1f us between %s -> %s" % (g, a, b))
