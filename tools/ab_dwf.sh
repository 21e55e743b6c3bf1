#!/bin/bash
# A/B of the depthwise forward kernels between two builds of the library (T3D_LIB=<old .so> vs the in-tree one)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OLD=${OLD:-tools/scratch/ab/libt3d_hip_old.so}
for args in "dwfwd 256 112 112 32 3 1" "dwfwd 256 56 56 144 3 1" "dwfwd 256 28 28 192 3 1" "dwfwd 256 14 14 384 3 1" "dwfwd 256 14 14 576 3 1" "dwfwd 256 7 7 960 3 1" "dwfwd 256 112 112 96 3 2" "dwfwd 256 56 56 144 3 2" "$@"; do
  a=$(T3D_LIB=$OLD python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  b=$(python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  echo "OLD $a"; echo "NEW $b"
done
