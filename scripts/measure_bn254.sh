#!/bin/bash
# BN254 config-5 prove: kernel stats + one SQ counter pass (VALU instructions) -> gpurun_out/<tag>_bn254_*
tag=${1:-r02}
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bnprof_$tag -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > $O/${tag}_bn254_prove_times.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/bnpmc_$tag -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bn254_prove_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_trace.py $(ls $O/bnprof_$tag/*kernel_trace.csv | head -1) 40 > $O/${tag}_bn254_prove_kernel_trace_summary.txt
python scripts/pmc_sq.py $(ls $O/bnpmc_$tag/*counter_collection.csv | head -1) 40 > $O/${tag}_bn254_pmc_sq.txt
python - <<PY
import csv, json, collections, glob, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT/scripts")
from code_hash import code_hash
f = glob.glob("$O/bnpmc_$tag/*counter_collection.csv")[0]
tot = collections.Counter(); per = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "SQ_INSTS_VALU" and "hg::bn::" in r["Kernel_Name"]:
        per[r["Kernel_Name"].split("(")[0]] += float(r["Counter_Value"])
proves = 3  # scripts/bn254_prove_bench.py runs three proves
WITNESS_GEN = ("k_bn_ntt_stage", "k_bn_ntt4_", "k_bn_gate_eval", "k_bn_lift_signed", "k_bn_lift_jobs", "k_bn_bitrev", "k_bn_scale", "k_bn_powers")  # bn_witness_gen: outside the timed prove
wit = sum(v for k, v in per.items() if any(w in k for w in WITNESS_GEN))
out = {"code_hash": code_hash(), "command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -- python3 scripts/bn254_prove_bench.py (3 proves of n=32768 k=16)",
       "valu_wave_insts_per_prove": sum(per.values()) / proves,
       "witness_gen_valu_wave_insts_per_prove": wit / proves,
       "prove_valu_wave_insts_per_prove": (sum(per.values()) - wit) / proves,
       "by_kernel_per_prove": {k: v / proves for k, v in sorted(per.items(), key=lambda kv: -kv[1])}}
json.dump(out, open("$O/${tag}_bn254_pmc_sq.json", "w"), indent=1)
print(out["valu_wave_insts_per_prove"])
PY
rm -rf $O/bnprof_$tag $O/bnpmc_$tag
head -14 $O/${tag}_bn254_prove_kernel_trace_summary.txt; cat $O/${tag}_bn254_prove_times.txt | tail -3
