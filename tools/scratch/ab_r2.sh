#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "pwfwd 3211264 12544 16 96" "pwdgrad 3211264 12544 96 24" "pwfwd 200704 784 32 192" "pwdgrad 200704 784 192 32" "pwfwd 50176 196 576 96"; do
  a=$(python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  b=$(T3D_EXP_R2=1 python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  echo "R=1: $a"; echo "R=2: $b"
done
