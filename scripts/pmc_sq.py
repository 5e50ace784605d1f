#!/usr/bin/env python3
"""Per-kernel sums of SQ counters from a rocprofv3 --pmc counter_collection.csv (one pass).
Usage: pmc_sq.py <counter_collection.csv> [top]"""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    c = r["Counter_Name"]
    if c not in names: names.append(c)
    agg[k][c] += float(r["Counter_Value"])
    if c == names[0]: cnt[k] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print("launches " + " ".join(f"{n:>22}" for n in names) + "  kernel")
key = names[0]
for k in sorted(agg, key=lambda k: -agg[k][key])[:top]:
    print(f"{cnt[k]:8d} " + " ".join(f"{agg[k][n]:22.0f}" for n in names) + "  " + k)
