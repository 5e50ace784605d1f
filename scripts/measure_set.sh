set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r01i.json 2> gpurun_out/bench_r01i.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1i -o runc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_r01i_prof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_i_FETCH -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_i_WRITE -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prove_once.py 32768 16 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/summarize_trace.py $(ls gpurun_out/prof_r1i/*kernel_trace.csv | head -1) 50 > gpurun_out/prof_r1i_summary.txt
python scripts/pmc_summary.py $(ls gpurun_out/pmc_i_FETCH/*counter_collection.csv | head -1) $(ls gpurun_out/pmc_i_WRITE/*counter_collection.csv | head -1) gpurun_out/pmc_r01i.json > gpurun_out/pmc_r01i.txt
cp $(ls gpurun_out/prof_r1i/*kernel_stats.csv | head -1) gpurun_out/prof_r1i_kernel_stats.csv
rm -rf gpurun_out/prof_r1i gpurun_out/pmc_i_FETCH gpurun_out/pmc_i_WRITE
cat gpurun_out/bench_r01i.json; head -12 gpurun_out/pmc_r01i.txt
