#!/usr/bin/env python3
"""Lists the dispatches of one kernel from a rocprofv3 kernel-trace CSV: grid size and duration (last prove only).
usage: trace_kernel.py <dir-with-*_kernel_trace.csv> <kernel-substring>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
per = {}
for r in rows:
    key = (int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]), int(r.get("Grid_Size_Y", 1)))
    per.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(per, reverse=True):
    v = per[k]
    print(k, "n=%d" % len(v), "min %.1f us  median %.1f us" % (min(v), sorted(v)[len(v) // 2]))
