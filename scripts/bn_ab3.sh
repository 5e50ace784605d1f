#!/bin/bash
# same-box A/B of the config-5 bn254 prove: build/base (HEAD of the round's start) vs the working tree; then the bn254 tests and a per-dispatch list
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2; do
  HG_LIB=$GRAFT_REPO_ROOT/build/base/libhypergreco.so python3 scripts/bn254_prove_bench.py 2>&1 | grep hg_prove | tail -2 | sed 's/^/base: /'
  python3 scripts/bn254_prove_bench.py 2>&1 | grep hg_prove | tail -2 | sed 's/^/new:  /'
done > $O/${tag}_bn_ab.txt
cat $O/${tag}_bn_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bn254" > $O/${tag}_bn_tests.txt 2>&1
tail -3 $O/${tag}_bn_tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/bnprof_$tag -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
T=$(ls $O/bnprof_$tag/*kernel_trace.csv | head -1)
python scripts/summarize_trace.py $T 30 > $O/${tag}_bn254_prove_kernel_trace_summary.txt
python scripts/bn_dispatch_list.py $T > $O/${tag}_bn254_dispatches.txt 2>&1
rm -rf $O/bnprof_$tag
head -12 $O/${tag}_bn254_prove_kernel_trace_summary.txt
