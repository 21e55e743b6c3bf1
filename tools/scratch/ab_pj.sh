#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "3211264 12544 32 16" "802816 3136 96 24" "802816 3136 144 24" "200704 784 144 32" "200704 784 192 32"; do
  python tools/run_kernel.py pwdgrad $shape --reps 20 --nrep 16 2>&1 | tail -1
  python tools/run_kernel.py pwwgrad $shape --reps 20 --nrep 16 2>&1 | tail -1
  for sp in 256 512 768; do
    echo "S=$sp $(T3D_FUSED_SPLITS=$sp python tools/run_kernel.py pwbwd_proj $shape --reps 20 --nrep 16 2>&1 | tail -1)"
  done
done
for shape in "3211264 12544 16 96" "802816 3136 24 144" "200704 784 32 192"; do
  for sp in 256 512 768; do
    echo "S=$sp $(T3D_FUSED_SPLITS=$sp python tools/run_kernel.py pwbwd_yf $shape --reps 20 --nrep 16 2>&1 | tail -1)"
  done
done
