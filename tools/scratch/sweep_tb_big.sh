#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "dwfwd 256 112 112 32 3 1" "dwfwd 256 56 56 144 3 1" "dwfwd 256 112 112 96 3 2" "dwfwd 256 56 56 144 3 2" "dwbwd 256 112 112 32 3 1" "dwbwd 256 56 56 144 3 1" "dwbwd 256 112 112 96 3 2" "dwbwd 256 56 56 144 3 2"; do
  echo "== $args"
  for tb in 0 256 384 512 640 768 1024; do
    echo -n "tb=$tb "; T3D_DW_TB=$tb python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1 | sed 's/.*: //'
  done
done
