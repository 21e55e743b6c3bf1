#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp T3D_TRACE=deep
for e in 0 1 2 3; do
  if [ $e != 0 ]; then export T3D_LIB=tools/scratch/ab/libt3d_hip_exp$e.so; fi
  echo "== exp $e (1: no operand loads in the loop, 2: no weight loads, 3: 1 of 16 MFMAs)"
  for args in "pwfwd 12544 49 960 160" "pwdgrad 12544 49 160 960"; do
    python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | grep -v amdgpu | tail -3 | head -2
  done
done
