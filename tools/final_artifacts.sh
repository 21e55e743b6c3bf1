#!/bin/bash
# Regenerate the judged artifacts on the GPU box: full GPU test suite, default bench line, rocprofv3 kernel stats of the
# same command, PMC traffic.  Outputs under gpurun_out/final/ (copy into profiles/ afterwards).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
o=gpurun_out/final; rm -rf $o; mkdir -p $o
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $o/pytest_gpu.txt; cat $o/pytest_gpu.txt
python bench.py > $o/bench.json 2> $o/bench.err; tail -1 $o/bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -o t -- python3 bench.py --no-cpu-baseline > $o/bench_under_rocprof.json 2> /dev/null
find $o/prof -name "*kernel_stats.csv" -exec cp {} $o/kernel_stats.csv \;
python3 tools/trace_streams.py $(find $o/prof -name "*kernel_trace.csv") > $o/trace_streams.txt 2>&1
rm -rf $o/prof
tools/pmc_traffic.sh $o/hbm_traffic_pmc.json
head -12 $o/kernel_stats.csv | cut -c1-160
