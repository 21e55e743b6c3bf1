#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/tu_$$
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline ${1:+--model $1} > /dev/null 2>&1
python3 tools/trace_turn.py $(find $d -name "*kernel_trace.csv")
rm -rf $d
