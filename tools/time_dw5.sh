#!/bin/bash
# isolated timings of MobileNetV3-large's 5x5 / squeeze-excite depthwise layers (forward with pooled sums, backward), B = 256
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shp in "256 56 56 72 5 2 relu" "256 28 28 120 5 1 relu" "256 14 14 480 3 1 hswish" "256 14 14 672 3 1 hswish" "256 14 14 672 5 2 hswish" "256 7 7 960 5 1 hswish"; do
  set -- $shp
  python tools/run_kernel.py dwfwd $1 $2 $3 $4 $5 $6 --act $7 --gap --reps 20 --nrep 8 2>&1 | tail -1
  python tools/run_kernel.py dwbwd $1 $2 $3 $4 $5 $6 --act $7 --reps 20 --nrep 8 2>&1 | tail -1
done
