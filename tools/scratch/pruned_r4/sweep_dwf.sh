#!/bin/bash
# sweep the block-count target of the streaming depthwise forward kernels (16 reduction replicas)
for nb in 256 512 768 1024 1536 2048; do
  echo "== blocks $nb"
  for sh in "256 112 112 32 3 1" "256 112 112 96 3 2" "256 56 56 144 3 1" "256 56 56 144 3 2" "256 28 28 192 3 1" "256 28 28 192 3 2" "256 14 14 384 3 1" "256 14 14 576 3 1" "256 14 14 576 3 2" "256 7 7 960 3 1"; do
    T3D_DWF_BLOCKS=$nb python tools/run_kernel.py dwfwd $sh --reps 10 --nrep 16 2>&1 | tail -1
  done
done
