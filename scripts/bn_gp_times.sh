#!/bin/bash
# first three k_bn_gp_round_jobs launch durations of a traced BN254 prove, for each HG_BN_DBG value given
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for d in "$@"; do
  cd /tmp && export TMPDIR=/tmp
  HG_BN_DBG=$d rocprofv3 --kernel-trace -d $O/bnp_$d -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  echo "dbg=$d: $(python scripts/trace_timeline.py $(ls $O/bnp_$d/*kernel_trace.csv | head -1) | grep gp_round_jobs | head -3 | awk '{printf "%s ", $2}')"
  rm -rf $O/bnp_$d
done
