#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pwconv.py tests/test_gpu_production_shapes.py -x -q 2>&1 | tail -2
for args in "pwfwd 3211264 12544 16 96" "pwfwd 802816 3136 24 144" "pwfwd 200704 784 32 192" "pwfwd 50176 196 64 384" "pwfwd 50176 196 96 576" "pwfwd 12544 49 160 960" "pwdgrad 3211264 12544 96 24" "pwdgrad 802816 3136 144 24" "pwdgrad 200704 784 192 32" "pwdgrad 50176 196 576 96" "pwdgrad_yf 3211264 12544 16 96" "pwfwd 3211264 12544 32 16"; do
  a=$(T3D_PW_NO_LSTORE=1 python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  b=$(python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  echo "direct: $a"; echo "lstore: $b"
done
