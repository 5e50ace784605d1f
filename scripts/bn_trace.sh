#!/bin/bash
# kernel trace of scripts/bn254_prove_bench.py -> gpurun_out/<tag>_bn254_{summary,timeline,buckets}.txt
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bnprof_$tag -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > $O/${tag}_bn254_prove_times.txt 2>&1
cd $GRAFT_REPO_ROOT
T=$(ls $O/bnprof_$tag/*kernel_trace.csv | head -1)
python scripts/summarize_trace.py $T 60 > $O/${tag}_bn254_prove_kernel_trace_summary.txt
python scripts/bn_timeline.py $T > $O/${tag}_bn254_buckets.txt 2>&1
python scripts/trace_timeline.py $T > $O/${tag}_bn254_timeline.txt 2>&1
rm -rf $O/bnprof_$tag
cat $O/${tag}_bn254_prove_times.txt | tail -3
