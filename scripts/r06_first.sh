#!/bin/bash
# round 6, first GPU call: issue-rate table, baseline bench line, per-dispatch list of one bn254 prove
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
( cd scripts/ub && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 ratebench.hip -o ratebench 2>/dev/null; ./ratebench ) > $O/r06_ratebench.txt 2>&1
python bench.py > $O/r06_a_bench.json 2> $O/r06_a_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/bnprof_r06a -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > $O/r06_a_bn254_prove_times.txt 2>&1
cd $GRAFT_REPO_ROOT
T=$(ls $O/bnprof_r06a/*kernel_trace.csv | head -1)
python scripts/summarize_trace.py $T 60 > $O/r06_a_bn254_prove_kernel_trace_summary.txt
python scripts/bn_dispatch_list.py $T > $O/r06_a_bn254_dispatches.txt 2>&1
rm -rf $O/bnprof_r06a
tail -3 $O/r06_a_bn254_prove_times.txt; head -50 $O/r06_ratebench.txt
