#!/bin/bash
# HBM traffic of the bn254 kernels: FETCH_SIZE and WRITE_SIZE in separate passes (gfx950: bytes = (2 FETCH + WRITE) * 1024), with durations
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/bnf_$tag -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/bnw_$tag -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY > $O/${tag}_bn_traffic.txt
import csv, glob, collections
def load(d, name):
    tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0]))}
    out = []
    for r in csv.DictReader(open(glob.glob(d + "/*counter_collection.csv")[0])):
        if r["Counter_Name"] != name: continue
        t = tr.get(r["Dispatch_Id"])
        dur = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3 if t else 0
        out.append((r["Kernel_Name"].split("(")[0].replace("hg::bn::", "").replace("hg::dev::", ""), float(r["Counter_Value"]), dur))
    return out
F = load("$O/bnf_$tag", "FETCH_SIZE"); W = load("$O/bnw_$tag", "WRITE_SIZE")
# same dispatch order in both runs: pair by index
print("%9s %9s %9s %8s  kernel" % ("us", "read MB", "write MB", "TB/s"))
rows = []
for (n, f, d), (n2, w, d2) in zip(F, W):
    if n != n2: continue
    rows.append((d, 2 * f * 1024 / 1e6, w * 1024 / 1e6, n))
seen = collections.Counter()
for d, rmb, wmb, n in sorted(rows, key=lambda r: -r[0]):
    seen[n] += 1
    if seen[n] > 4 or d < 30: continue
    print("%9.1f %9.1f %9.1f %8.2f  %s" % (d, rmb, wmb, (rmb + wmb) / d / 1e6 * 1e6 / 1e6 if d else 0, n[:50]))
PY
rm -rf $O/bnf_$tag $O/bnw_$tag
head -40 $O/${tag}_bn_traffic.txt
