#!/bin/bash
# bn254 GPU tests + three timed config-5 proves (no profiler) -> gpurun_out/<tag>_bn_quick.txt
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bn254" > $O/${tag}_bn_tests.txt 2>&1
tail -3 $O/${tag}_bn_tests.txt
python3 scripts/bn254_prove_bench.py > $O/${tag}_bn_quick.txt 2>&1; python3 scripts/bn254_prove_bench.py >> $O/${tag}_bn_quick.txt 2>&1
grep hg_prove $O/${tag}_bn_quick.txt
