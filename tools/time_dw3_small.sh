#!/bin/bash
# 3x3 depthwise on the small planes of MobileNetV2 (B = 256): register tiles (dwconv_tile.hip) against the row-walk kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shp in "256 28 28 192 3 1" "256 28 28 192 3 2" "256 14 14 384 3 1" "256 14 14 576 3 1" "256 14 14 576 3 2" "256 7 7 960 3 1"; do
  for kind in dwfwd dwbwd; do
    echo -n "walk: "; T3D_DW3_TILE_MAX=0 python tools/run_kernel.py $kind $shp --reps 20 --nrep 8 2>&1 | tail -1
    echo -n "tile: "; T3D_DW3_TILE_MAX=28 python tools/run_kernel.py $kind $shp --reps 20 --nrep 8 2>&1 | tail -1
  done
done
