#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV: per kernel name count / total / avg / max duration, plus the
largest dispatches. Usage: summarize_trace.py <kernel_trace.csv> [top]"""
import csv, sys, collections
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[r["Kernel_Name"]]
    a[0] += 1; a[1] += d; a[2] = max(a[2], d)
tot = sum(a[1] for a in agg.values())
print(f"dispatches {len(rows)}  total kernel time {tot/1e3:.3f} ms")
print(f"{'count':>7} {'total_us':>12} {'avg_us':>10} {'max_us':>10} {'%':>6}  kernel")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:10.1f} {100*a[1]/tot:6.2f}  {name[:110]}")
