#!/bin/bash
# Round-6 measurement set (run on the GPU box through gpurun): bench line, rocprofv3 kernel traces (three streams / one stream /
# the bench command itself), HBM traffic (two PMC passes) and an SQ counter pass on the same resident-prove workload; the counter
# files carry the hash of the sources they were taken on (scripts/code_hash.py). usage: scripts/measure_r06.sh <tag> [pmc]
tag=${1:-r06_a}
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [ "$2" = "pmc" ]; then
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_${tag}_FETCH -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_${tag}_WRITE -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/pmc_${tag}_SQ -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $(ls $O/pmc_${tag}_FETCH/*counter_collection.csv | head -1) $(ls $O/pmc_${tag}_WRITE/*counter_collection.csv | head -1) $O/${tag}_pmc_hbm_traffic.json > $O/${tag}_pmc_hbm_traffic.txt
python scripts/pmc_sq.py $(ls $O/pmc_${tag}_SQ/*counter_collection.csv | head -1) 40 $O/${tag}_pmc_sq.json > $O/${tag}_pmc_sq.txt
rm -rf $O/pmc_${tag}_FETCH $O/pmc_${tag}_WRITE $O/pmc_${tag}_SQ
# bench.py reads the counter files from profiles/: put this run's there for the bench line below (they are committed afterwards)
cp $O/${tag}_pmc_hbm_traffic.json $GRAFT_REPO_ROOT/profiles/${tag%_*}_pmc_hbm_traffic.json
cp $O/${tag}_pmc_sq.json $GRAFT_REPO_ROOT/profiles/${tag%_*}_pmc_sq.json
fi
cd $GRAFT_REPO_ROOT
# the issue model's inputs: what each opcode class issues at on THIS box, and the class mix of the kernels as compiled from these sources
( cd scripts/ub && ./ratebench ) > $GRAFT_REPO_ROOT/profiles/${tag%_*}_ratebench.txt 2>&1
python3 scripts/isa_census.py hyper-greco_amd/csrc/kernels.hip hyper-greco_amd/csrc/bn254.hip --json $GRAFT_REPO_ROOT/profiles/${tag%_*}_isa_mix.json > $O/${tag}_isa_census.txt 2>&1
cp $GRAFT_REPO_ROOT/profiles/${tag%_*}_ratebench.txt $O/${tag}_ratebench.txt; cp $GRAFT_REPO_ROOT/profiles/${tag%_*}_isa_mix.json $O/${tag}_isa_mix.json
python bench.py > $O/bench_$tag.json 2> $O/bench_$tag.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 6 > $O/prove_once_$tag.txt 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/profb_$tag -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_${tag}_under_rocprof.json 2>/dev/null
HG_ONE_STREAM=1 rocprofv3 --kernel-trace --stats -d $O/profi_$tag -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 6 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_trace.py $(ls $O/profb_$tag/*kernel_trace.csv | head -1) 60 > $O/${tag}_bench_kernel_trace_summary.txt
python3 scripts/summarize_trace.py $(ls $O/profi_$tag/*kernel_trace.csv | head -1) 60 > $O/${tag}_one_stream_kernel_trace_summary.txt
python3 scripts/summarize_trace.py $(ls $O/prof_$tag/*kernel_trace.csv | head -1) 60 > $O/${tag}_kernel_trace_summary.txt
cp $(ls $O/prof_$tag/*kernel_stats.csv | head -1) $O/${tag}_rocprofv3_kernel_stats.csv
rm -rf $O/prof_$tag $O/profb_$tag $O/profi_$tag
cat $O/bench_$tag.json | head -c 1500; head -30 $O/${tag}_kernel_trace_summary.txt
