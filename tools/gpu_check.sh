#!/bin/bash
# one GPU round trip: full -m gpu suite, then the default bench and the fp32 / eval variants
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=40 -s -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; tail -1 gpurun_out/bench_bf16.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-all > gpurun_out/bench_bf16_prof.json 2> gpurun_out/bench_bf16_prof.err; tail -25 gpurun_out/bench_bf16_prof.err
python bench.py --steps 10 --warmup 3 --dtype f32 --no-cpu-baseline > gpurun_out/bench_f32.json 2> gpurun_out/bench_f32.err; tail -1 gpurun_out/bench_f32.json
python bench.py --steps 20 --warmup 5 --eval --no-cpu-baseline > gpurun_out/bench_eval.json 2> gpurun_out/bench_eval.err; tail -1 gpurun_out/bench_eval.json
