#!/bin/bash
# rocprofv3 kernel trace of a short default bench -> gpurun_out/trace_summary.txt (+ the per-kernel stats csv)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/trace_tmp; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline "$@" > gpurun_out/trace_bench.json 2> /dev/null
python3 tools/trace_summary.py $(find $d -name "*kernel_trace.csv") > gpurun_out/trace_summary.txt 2>&1
find $d -name "*kernel_stats.csv" -exec cp {} gpurun_out/trace_kernel_stats.csv \;
rm -rf $d
cat gpurun_out/trace_summary.txt
