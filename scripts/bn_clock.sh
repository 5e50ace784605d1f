#!/bin/bash
# effective clock and wait breakdown of the bn254 round kernels: GRBM_GUI_ACTIVE / 8 / duration per dispatch, SQ wait counters; base vs working tree
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for which in base new; do
  if [ $which = base ]; then export HG_LIB=$GRAFT_REPO_ROOT/build/base/libhypergreco.so; else unset HG_LIB; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/bnclk_${tag}_$which -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 scripts/bn_clock.py $O/bnclk_${tag}_base > $O/${tag}_bn_clock_base.txt 2>&1
python3 scripts/bn_clock.py $O/bnclk_${tag}_new > $O/${tag}_bn_clock_new.txt 2>&1
rm -rf $O/bnclk_${tag}_base $O/bnclk_${tag}_new
head -30 $O/${tag}_bn_clock_base.txt; head -30 $O/${tag}_bn_clock_new.txt
