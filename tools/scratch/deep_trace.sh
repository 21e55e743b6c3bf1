#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "pwfwd 12544 49 960 160" "pwfwd 12544 49 576 160" "pwfwd 12544 49 960 320" "pwdgrad 12544 49 160 960" "pwdgrad 12544 49 320 1280"; do
  python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | grep -v amdgpu | tail -1
  T3D_TRACE=deep python tools/run_kernel.py $args --reps 20 --nrep 16 --frag 2>&1 | grep -v amdgpu | tail -3
done
