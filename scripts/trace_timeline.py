#!/usr/bin/env python3
"""Timeline of the last prove in a rocprofv3 kernel-trace CSV: start offset, duration, queue and name of every dispatch,
plus how much of the span has two kernels in flight. usage: trace_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
cut = max([i for i in range(len(rows) - 1) if rows[i + 1][0] - max(e for _, e, _, _ in rows[:i + 1]) > 150000] or [-1])
last = rows[cut + 1:]
t0 = last[0][0]
span = max(e for _, e, _, _ in last) - t0
ev = sorted([(s, 1) for s, e, _, _ in last] + [(e, -1) for s, e, _, _ in last])
depth, prev, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - prev)
    depth += d
    prev = t
print("span %.3f ms; time with k kernels in flight: %s" % (span / 1e6, {k: round(v / 1e6, 3) for k, v in sorted(hist.items())}))
for s, e, n, q in last:
    print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:80].replace("hg::dev::", "")))
