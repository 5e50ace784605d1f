#!/bin/bash
# Timeline of the last prove of an isolated (one-stream) run: every dispatch in order with its start offset and duration.
# usage: scripts/one_stream_timeline.sh   (environment settings are passed through)
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
HG_ONE_STREAM=1 rocprofv3 --kernel-trace -d $O/prof_tl -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 3 > /dev/null 2>&1
python3 - <<PY
import csv, glob
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(glob.glob("$O/prof_tl/*kernel_trace.csv")[0]))), key=lambda r: r[0])
# the last prove starts at the last k_clear_words
starts = [i for i, r in enumerate(rows) if "k_clear_words" in r[2]]
last = rows[starts[-1]:]
t0 = last[0][0]
busy = sum(e - s for s, e, _ in last)
print("dispatches %d, span %.1f us, kernel time %.1f us" % (len(last), (last[-1][1] - t0) / 1e3, busy / 1e3))
for s, e, n in last:
    print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n.replace("hg::dev::", "").replace("hg::", "")[:100]))
PY
rm -rf $O/prof_tl
