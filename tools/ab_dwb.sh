#!/bin/bash
# A/B of one kernel family between two builds of the library (T3D_LIB=<old .so> vs the in-tree one), isolated launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OLD=${OLD:-tools/scratch/ab/libt3d_hip_old.so}
for args in "dwbwd 256 56 56 144 3 1 --res" "dwbwd 256 28 28 192 3 1 --res" "dwbwd 256 14 14 384 3 1 --res" "dwbwd 256 14 14 576 3 1 --res" "dwbwd 256 7 7 960 3 1 --res" \
            "dwbwd 256 112 112 32 3 1" "dwbwd 256 56 56 144 3 1" "$@"; do
  a=$(T3D_LIB=$OLD python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  b=$(python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1)
  echo "OLD $a"; echo "NEW $b"
done
