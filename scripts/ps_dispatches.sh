#!/bin/bash
# Per-dispatch durations and grid sizes of the PRODSUM kernels of one isolated (one-stream) prove. usage: scripts/ps_dispatches.sh
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
HG_ONE_STREAM=1 rocprofv3 --kernel-trace -d $O/prof_ps -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 3 > /dev/null 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/prof_ps/*kernel_trace.csv")[0])))
ps = [r for r in rows if "k_ps_" in r["Kernel_Name"]]
n = len(ps) // 3
for r in ps[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%-12s grid %8s wg %5s  %8.1f us" % (r["Kernel_Name"].split("(")[0].split("::")[-1], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), d))
PY
rm -rf $O/prof_ps
