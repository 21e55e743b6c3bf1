#!/bin/bash
# usage: tools/gpu_tests.sh [pytest args]   (default: the whole -m gpu suite)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest ${@:-tests} -m gpu -q --maxfail=40 -s -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
