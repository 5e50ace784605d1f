#!/bin/bash
# Timeline of the last prove of an isolated (one-stream) run: every dispatch in order with its duration and the gap before it.
# usage: scripts/one_stream_timeline.sh [env settings]
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
HG_ONE_STREAM=1 rocprofv3 --kernel-trace -d $O/prof_tl -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 3 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/scripts/trace_timeline.py $(ls $O/prof_tl/*kernel_trace.csv | head -1)
rm -rf $O/prof_tl
