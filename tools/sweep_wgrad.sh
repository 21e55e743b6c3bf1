#!/bin/bash
# usage: sweep_wgrad.sh  -> prints total wgrad ms per (blocks, flush MB) setting
for blocks in 512 1024; do for mb in 8 32 128; do
  T3D_WG_BLOCKS=$blocks T3D_WG_FLUSH_MB=$mb python bench.py --steps 2 --warmup 2 --per-launch --no-cpu-baseline 2>&1 | grep t3d_pwconv_wgrad | awk -v b=$blocks -v m=$mb '{s+=$(NF-6)} END {printf "blocks %d flushMB %d: wgrad %.2f ms\n", b, m, s/1000}'
done; done
