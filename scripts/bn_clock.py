#!/usr/bin/env python3
"""Per kernel (largest dispatch of each): duration, effective clock = GRBM_GUI_ACTIVE / 8 / duration, VALU instructions per wave-quad-cycle and the
wait shares, from one rocprofv3 --kernel-trace --pmc run directory. usage: bn_clock.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0]))}
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    cnt[r["Dispatch_Id"]]["_name"] = r["Kernel_Name"].split("(")[0].replace("hg::bn::", "").replace("hg::dev::", "")
rows = []
for did, c in cnt.items():
    t = trace.get(did)
    if not t: continue
    dur = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
    rows.append((dur, c))
rows.sort(key=lambda r: -r[0])
print("%9s %7s %9s %8s %8s %8s %8s  kernel" % ("us", "GHz", "Minst", "inst/wqc", "waitany", "waitinst", "active"))
seen = collections.Counter()
for dur, c in rows:
    n = c["_name"]
    seen[n] += 1
    if seen[n] > 3 or dur < 20: continue
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("%9.1f %7.3f %9.2f %8.3f %8.3f %8.3f %8.3f  %s" % (dur, c.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur * 1e3), c.get("SQ_INSTS_VALU", 0) / 1e6, c.get("SQ_INSTS_VALU", 0) / wc,
          c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc, n[:50]))
