#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "802816 3136 144 24" "802816 3136 96 24" "200704 784 192 32" "200704 784 144 32" "50176 196 384 64" "50176 196 576 96" "12544 49 960 160" "3211264 12544 32 16"; do
  for e in "X=1" "T3D_WG_SK=2" "T3D_WG_SK=4" "T3D_WG_BLOCKS=512" "T3D_WG_BLOCKS=512 T3D_WG_SK=2" "T3D_WG_MIN_STEPS=4"; do
    r=$(env $e python tools/run_kernel.py pwwgrad $shape --reps 20 --nrep 16 2>&1 | tail -1)
    echo "$e | $r"
  done
done
