#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "pwfwd 12544 49 960 160" "pwfwd 12544 49 960 320" "pwfwd 12544 49 576 160" "pwfwd 12544 49 320 1280" "pwdgrad 12544 49 160 960" "pwdgrad 12544 49 320 1280" "pwdgrad 12544 49 960 160" "pwfwd 50176 196 576 96" "pwdgrad 50176 196 96 576"; do
  for mk in 1000000 128; do
    echo -n "mink=$mk  "; T3D_PW_TILED_MINK=$mk python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1
  done
done
