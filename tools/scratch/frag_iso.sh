#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_pwconv.py -x -q -k fragment 2>&1 | grep -E "assert|Error|error|passed|failed" | head -20
for args in "pwfwd 50176 196 384 64" "pwfwd 50176 196 64 384" "pwfwd 12544 49 160 960" "pwfwd 200704 784 192 32" "pwdgrad 50176 196 576 96" "pwdgrad 12544 49 960 160" "pwfwd 802816 3136 144 24"; do
  echo -n "plain: "; python tools/run_kernel.py $args --reps 30 --nrep 16 2>&1 | tail -1
  echo -n "frag:  "; python tools/run_kernel.py $args --reps 30 --nrep 16 --frag 2>&1 | tail -1
done
