#!/bin/bash
# in-kernel phase timeline of the pointwise stream kernel on the small-stage shapes (needs the -DT3D_PW_TRACE build:
#   HIPCC_EXTRA=-DT3D_PW_TRACE python 3d-object-detection.pytorch_amd/build.py --force   before gpurun)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp T3D_TRACE=1
for args in "pwfwd 12544 49 960 160" "pwfwd 12544 49 160 960" "pwfwd 50176 196 384 64" "pwfwd 50176 196 576 96" "pwdgrad 50176 196 576 96" "pwdgrad 12544 49 960 160" "pwfwd 802816 3136 144 24"; do
  python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -4
done
