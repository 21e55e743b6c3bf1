#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "pwfwd 802816 3136 144 24" "pwfwd 3211264 12544 32 16" "pwfwd 200704 784 192 32" "pwdgrad_yf 3211264 12544 16 96" "pwdgrad_yf 802816 3136 24 144" "pwfwd 50176 196 384 64" "pwfwd 200704 784 144 32" "pwdgrad_yf 50176 196 64 384" "pwfwd 50176 196 64 384" "pwdgrad 50176 196 384 64"; do
  a=$(python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  b=$(T3D_EXP_R4=1 T3D_EXP_R8=1 python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  echo "base: $a"; echo "R+  : $b"
done
