#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "pwfwd 50176 196 96 576" "pwdgrad 50176 196 576 96" "pwfwd 50176 196 64 384" "pwdgrad 50176 196 384 64" "pwfwd 12544 49 160 960" "pwdgrad 12544 49 960 160" "pwfwd 802816 3136 24 144" "pwdgrad 802816 3136 144 24" "pwfwd 12544 49 320 1280" "pwdgrad 200704 784 192 32"; do
  a=$(python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  b=$(T3D_EXP_NT=6 python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  echo "base: $a"; echo "NT=6: $b"
done
