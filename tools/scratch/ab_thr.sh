#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "pwfwd 3211264 12544 16 96" "pwdgrad 3211264 12544 96 24" "pwfwd 802816 3136 24 144" "pwfwd 802816 3136 144 24" "pwdgrad 802816 3136 144 24" "pwfwd 3211264 12544 32 16" "pwfwd 200704 784 32 192" "pwfwd 50176 196 96 576" "pwdgrad_yf 3211264 12544 16 96"; do
  a=$(python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  b=$(T3D_EXP_THREADS=512 python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  c=$(T3D_EXP_THREADS=256 python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  echo "base: $a"; echo "512 : $b"; echo "256 : $c"
done
