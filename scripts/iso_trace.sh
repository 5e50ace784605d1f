#!/bin/bash
# Kernel durations of the resident prove, overlapped (two streams) and isolated (HG_ONE_STREAM=1). usage: scripts/iso_trace.sh <tag>
tag=${1:-x}
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof_$tag -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 6 > /dev/null 2>&1
HG_ONE_STREAM=1 rocprofv3 --kernel-trace -d $O/profi_$tag -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 6 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/scripts/summarize_trace.py $(ls $O/prof_$tag/*kernel_trace.csv | head -1) 60 > $O/${tag}_kernel_trace_summary.txt
python3 $GRAFT_REPO_ROOT/scripts/summarize_trace.py $(ls $O/profi_$tag/*kernel_trace.csv | head -1) 60 > $O/${tag}_one_stream_kernel_trace_summary.txt
rm -rf $O/prof_$tag $O/profi_$tag
