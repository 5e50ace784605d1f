#!/usr/bin/env python3
"""Every dispatch of the LAST hg_prove_bn254 in a rocprofv3 kernel trace, in start order: start (us from the prove's first kernel), duration, queue, kernel.
usage: bn_dispatch_list.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("hg::bn::", "").replace("hg::dev::", "").replace("void ", ""),
                r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
idx = max(i for i, r in enumerate(rows) if "k_bn_low_limb" in r[2])
gi = max(i for i, r in enumerate(rows[:idx]) if "k_bn_gate_eval" in r[2])
last = rows[gi + 1:]
t0 = last[0][0]
print("kernels", len(last), "span %.2f ms" % ((max(r[1] for r in last) - t0) / 1e6))
for s, e, n, q, g, w in last:
    print("%9.1f %8.1f q%s %8s/%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, g, w, n[:60]))
