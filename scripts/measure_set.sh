set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r01j.json 2> gpurun_out/bench_r01j.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1j -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_r01j_prof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_j_FETCH -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_j_WRITE -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_trace.py $(ls gpurun_out/prof_r1j/*kernel_trace.csv | head -1) 50 > gpurun_out/prof_r1j_summary.txt
python scripts/pmc_summary.py $(ls gpurun_out/pmc_j_FETCH/*counter_collection.csv | head -1) $(ls gpurun_out/pmc_j_WRITE/*counter_collection.csv | head -1) gpurun_out/pmc_r01j.json > gpurun_out/pmc_r01j.txt
cp $(ls gpurun_out/prof_r1j/*kernel_stats.csv | head -1) gpurun_out/prof_r1j_kernel_stats.csv
rm -rf gpurun_out/prof_r1j gpurun_out/pmc_j_FETCH gpurun_out/pmc_j_WRITE
cat gpurun_out/bench_r01j.json; head -12 gpurun_out/pmc_r01j.txt
