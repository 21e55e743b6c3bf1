#!/bin/bash
# launch-geometry sweep of the 5x5 depthwise kernels (row chunks per image, target workgroups)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== baseline"; bash tools/time_dw5.sh
for shp in "256 56 56 72 5 2 relu" "256 28 28 120 5 1 relu" "256 14 14 672 5 2 hswish" "256 7 7 960 5 1 hswish"; do
  set -- $shp
  for ch in 1 2; do for tb in 256 512 1024 2048; do
    echo -n "fwd chunks=$ch tb=$tb: "; T3D_DWK_CHUNKS=$ch T3D_DWK_TB=$tb python tools/run_kernel.py dwfwd $1 $2 $3 $4 $5 $6 --act $7 --gap --reps 20 --nrep 8 2>&1 | tail -1
    echo -n "bwd chunks=$ch tb=$tb: "; T3D_DW5_CHUNKS=$ch T3D_DW5_TB=$tb python tools/run_kernel.py dwbwd $1 $2 $3 $4 $5 $6 --act $7 --reps 20 --nrep 8 2>&1 | tail -1
  done; done
done
