#!/bin/bash
# isolated timings of the pointwise backward kernels (weight gradient, data gradient) on every projection / expansion shape of
# MobileNetV2 @224^2, B = 256.  usage: tools/time_pw_bwd.sh [out-file]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=${1:-gpurun_out/pw_bwd_isolated.txt}
: > $out
for shp in "3211264 12544 32 16" "802816 3136 96 24" "802816 3136 144 24" "200704 784 144 32" "200704 784 192 32" \
           "50176 196 192 64" "50176 196 384 64" "50176 196 384 96" "50176 196 576 96" \
           "12544 49 576 160" "12544 49 960 160" "12544 49 960 320" "12544 49 160 960" "12544 49 320 1280"; do
  for kind in pwwgrad pwdgrad; do
    python tools/run_kernel.py $kind $shp --reps 20 --nrep 16 2>&1 | tail -1 >> $out
  done
done
cat $out
