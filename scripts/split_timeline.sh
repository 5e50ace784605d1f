#!/bin/bash
# multi-stream timeline of the last of six resident proves (graph replay), with and without the split rounds -> gpurun_out/<tag>_{split,nosplit}_timeline.txt
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in split nosplit; do
  rm -rf $O/st_$v
  if [ $v = nosplit ]; then export HG_NO_SPLIT=1; else unset HG_NO_SPLIT; fi
  rocprofv3 --kernel-trace -d $O/st_$v -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 6 > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/scripts/trace_timeline.py $(ls $O/st_$v/*kernel_trace.csv | head -1) > $O/${tag}_${v}_timeline.txt
  rm -rf $O/st_$v
  head -1 $O/${tag}_${v}_timeline.txt
done
