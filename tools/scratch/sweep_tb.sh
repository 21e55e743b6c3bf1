#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "dwfwd 256 28 28 192 3 1" "dwfwd 256 14 14 384 3 1" "dwfwd 256 14 14 576 3 1" "dwfwd 256 7 7 960 3 1" "dwfwd 256 28 28 192 3 2" "dwfwd 256 14 14 576 3 2" "dwbwd 256 28 28 192 3 1" "dwbwd 256 14 14 384 3 1" "dwbwd 256 14 14 576 3 1" "dwbwd 256 7 7 960 3 1" "dwbwd 256 28 28 192 3 2" "dwbwd 256 14 14 576 3 2"; do
  echo "== $args"
  for tb in 0 256 384 512 640 768 896 1024 1280; do
    echo -n "tb=$tb "; T3D_DW_TB=$tb python tools/run_kernel.py $args --reps 30 --nrep 16 2>&1 | tail -1 | sed 's/.*: //'
  done
done
