#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "3211264 12544 16 96" "802816 3136 24 144" "200704 784 32 192"; do
  python tools/run_kernel.py pwdgrad_yf $shape --reps 20 --nrep 16 2>&1 | tail -1
  python tools/run_kernel.py pwwgrad_yf $shape --reps 20 --nrep 16 2>&1 | tail -1
  python tools/run_kernel.py pwbwd_yf $shape --reps 20 --nrep 16 2>&1 | tail -1
  python tools/run_kernel.py pwbwd_yf $shape --reps 20 --nrep 16 --with-finish 2>&1 | tail -1
done
