#!/bin/bash
# sweep block size / block-count target of the streaming depthwise backward kernels (16 reduction replicas, as the engine runs them)
for nth in 256 512; do
for nb in 128 192 256 384 512; do
  echo "== threads $nth blocks $nb"
  for sh in "256 112 112 96 3 2" "256 56 56 144 3 2" "256 28 28 192 3 2" "256 14 14 576 3 2"; do
    T3D_DWB_THREADS=$nth T3D_DWB2_BLOCKS=$nb python tools/run_kernel.py dwbwd $sh --reps 10 --nrep 16 2>&1 | tail -1
  done
  for sh in "256 7 7 960 3 1" "256 14 14 576 3 1" "256 14 14 384 3 1" "256 28 28 192 3 1" "256 56 56 144 3 1" "256 112 112 32 3 1"; do
    T3D_DWB_THREADS=$nth T3D_DWB1_BLOCKS=$nb python tools/run_kernel.py dwbwd $sh --reps 10 --nrep 16 2>&1 | tail -1
  done
done
done
