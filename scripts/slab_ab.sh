#!/bin/bash
# slab rounds on / off: parity tests, one-stream timeline, prove times
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "headline or full_prove or sumcheck_kernels or grand_product or slot_form or strict" 2>&1 | tail -3
for v in "HG_NO_SLABS=1" "HG_SLAB_ROUNDS=4" "HG_SLAB_ROUNDS=5" "HG_SLAB_ROUNDS=6"; do
  echo "== $v"; env $v python3 scripts/prove_once.py 32768 16 8 2>&1 | tail -3 | python3 -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l.strip()); print('prove %.3f gpu %.3f' % (d['prove_ms'], d['gpu_ms']))"
done
bash scripts/one_stream_timeline.sh > $O/r06_slab_timeline.txt 2>&1
grep -n "k_st_step<1, E2\|k_gp_slab\|k_st_tail<1>\|k_gp_slot_regroup\|span" $O/r06_slab_timeline.txt | cut -c1-110
