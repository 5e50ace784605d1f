#!/usr/bin/env python3
"""Per-kernel sums of SQ counters from a rocprofv3 --pmc counter_collection.csv (one pass).
Usage: pmc_sq.py <counter_collection.csv> [top] [out.json]   (the json: per kernel launches and counter sums, with the code hash)"""
import csv, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from code_hash import code_hash
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:100]
    c = r["Counter_Name"]
    if c not in names: names.append(c)
    agg[k][c] += float(r["Counter_Value"])
    if c == names[0]: cnt[k] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print("launches " + " ".join(f"{n:>22}" for n in names) + "  kernel")
key = names[0]
for k in sorted(agg, key=lambda k: -agg[k][key])[:top]:
    print(f"{cnt[k]:8d} " + " ".join(f"{agg[k][n]:22.0f}" for n in names) + "  " + k)

if len(sys.argv) > 3:
    out = {k: dict(launches=cnt[k], **{n: agg[k][n] for n in names}) for k in agg}
    proves = max([v for k, v in cnt.items() if "k_clear_words" in k] or [0])   # one k_clear_words launch per prove
    out["_meta"] = {"code_hash": code_hash(), "proves": proves, "workload": f"scripts/prove_once.py 32768 16 {proves} ({proves} resident proves, no warm-up; sums over all launches)"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
