#!/bin/bash
# usage: trace_model.sh <model>: per-stream timeline of one training step (rocprofv3 kernel trace + tools/trace_streams.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/tm_$$
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 bench.py --model $1 --steps 6 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
python3 tools/trace_streams.py $(find $d -name "*kernel_trace.csv")
rm -rf $d
