#!/bin/bash
# mode 3 (absorbing transcript + extension-field memory checking) at n=32768 k=16: the library's own breakdown and a kernel trace
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
HG_TIMES=seq python3 scripts/mode_times.py 32768 16 > $O/r06_seq_times.txt 2>&1
tail -40 $O/r06_seq_times.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/seqprof -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/mode_times.py 32768 16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_trace.py $(ls $O/seqprof/*kernel_trace.csv | head -1) 30 > $O/r06_seq_kernel_trace_summary.txt
rm -rf $O/seqprof
head -34 $O/r06_seq_kernel_trace_summary.txt
