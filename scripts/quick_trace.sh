#!/bin/bash
# Quick look on the GPU box: interleaved A/B against build/base (if present), then a one-stream kernel trace of six resident proves.
# usage (through gpurun): scripts/quick_trace.sh [lines=24]
cd $GRAFT_REPO_ROOT
[ -f build/base/libhypergreco.so ] && R=${R:-3} scripts/ab.sh "HG_LIB=build/base/libhypergreco.so" "A=1" 2>&1 | tail -2
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/qt
HG_ONE_STREAM=1 rocprofv3 --kernel-trace --stats -d $O/qt -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 6 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_trace.py $(ls $O/qt/*kernel_trace.csv | head -1) ${1:-24}
rm -rf $O/qt
