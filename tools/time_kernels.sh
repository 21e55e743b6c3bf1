#!/bin/bash
# isolated kernel timings (one stream, nothing else running) for A/B comparisons
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for args in "pwfwd 3211264 12544 16 96" "pwfwd 3211264 12544 32 16" "pwfwd 802816 3136 144 24" "pwfwd 802816 3136 24 144" "pwfwd 200704 784 192 32" "pwfwd 50176 196 384 64" "pwfwd 12544 49 960 160" \
            "pwdgrad 802816 3136 144 24" "pwdgrad 3211264 12544 96 24" "pwdgrad 200704 784 192 32" "pwdgrad 50176 196 576 96" "pwdgrad_yf 3211264 12544 16 96" "pwdgrad_yf 802816 3136 24 144" \
            "dwfwd 256 112 112 32 3 1" "dwfwd 256 112 112 96 3 2" "dwfwd 256 56 56 144 3 1" "dwfwd 256 14 14 384 3 1" "dwfwd 256 7 7 960 3 1" \
            "dwbwd 256 112 112 32 3 1" "dwbwd 256 112 112 96 3 2" "dwbwd 256 56 56 144 3 1" "dwbwd 256 28 28 192 3 1" "dwbwd 256 14 14 384 3 1" "dwbwd 256 14 14 576 3 1" "dwbwd 256 7 7 960 3 1"; do
  python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1
done
# deep contractions of the 7x7 stage: streaming kernel (row-major weights) | deep-contraction kernel (fragment-order weights)
for args in "pwfwd 12544 49 960 160" "pwfwd 12544 49 576 160" "pwfwd 12544 49 960 320" "pwdgrad 12544 49 160 960" "pwdgrad 12544 49 320 1280"; do
  echo -n "stream: "; python tools/run_kernel.py $args --reps 20 --nrep 16 2>&1 | tail -1
  echo -n "deep:   "; python tools/run_kernel.py $args --reps 20 --nrep 16 --frag 2>&1 | tail -1
done
