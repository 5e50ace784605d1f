#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide prescribes) per
kernel: HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE counts 64 B per 128-B request on gfx950
(wide coalesced streams), hence the factor 2; WRITE_SIZE is exact for 16-B-per-lane stores.
Usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]"""
import csv, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from code_hash import code_hash

def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
print(f"{'launches':>8} {'fetch_KB(raw)':>16} {'write_KB':>14} {'HBM_MB/launch':>14}  kernel")
for k in sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, [0, 0])[1])):
    n, f = fetch[k]
    w = write.get(k, [n, 0.0])[1]
    per = (2 * f + w) * 1024 / max(n, 1)
    out[k] = {"launches": n, "fetch_kb_raw": f, "write_kb": w, "hbm_bytes_per_launch": per}
    print(f"{n:8d} {f:16.1f} {w:14.1f} {per/1e6:14.2f}  {k[:100]}")
if len(sys.argv) > 3:
    proves = max([v["launches"] for k, v in out.items() if "k_clear_words" in k] or [0])   # one k_clear_words launch per prove
    out["_meta"] = {"code_hash": code_hash(), "proves": proves, "workload": f"scripts/prove_once.py 32768 16 {proves} ({proves} resident proves, no warm-up; per-launch averages)"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
