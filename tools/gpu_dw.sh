#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_gpu_dwconv.py tests/test_gpu_production_shapes.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --per-launch 2> gpurun_out/perlaunch.txt > gpurun_out/perlaunch.json
grep dwconv gpurun_out/perlaunch.txt
python -c "
import json;d=json.loads(open('gpurun_out/perlaunch.json').read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step']);[print(r['kernel'],r['ms_per_step'],r['frac']) for r in d['roofline']['depthwise']]"
