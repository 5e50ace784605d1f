#!/bin/bash
# every dispatch of one bn254 prove (n=32768 k=16), in start order -> gpurun_out/<tag>_bn254_dispatches.txt (under the profiler the host falls behind
# with its enqueues: durations are right, the gaps between the two queues are not representative - NOTEBOOK.md round 6)
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/bnd_$tag -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/bn_dispatch_list.py $(ls $O/bnd_$tag/*kernel_trace.csv | head -1) > $O/${tag}_bn254_dispatches.txt 2>&1
rm -rf $O/bnd_$tag
head -3 $O/${tag}_bn254_dispatches.txt
