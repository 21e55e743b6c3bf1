#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "3211264 12544 16 96" "802816 3136 24 144" "200704 784 32 192" "50176 196 64 384" "50176 196 96 576"; do
  echo "NEW $(python tools/run_kernel.py yfprep $shape --reps 50 --nrep 8 2>&1 | tail -1)"
done
