#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/debug_bf16_grads.py > gpurun_out/dbg_yfree_on.txt 2>&1
T3D_YFREE_MIN=0 python tools/debug_bf16_grads.py > gpurun_out/dbg_yfree_off.txt 2>&1
DBG_ROUND_W=1 T3D_YFREE_MIN=0 python tools/debug_bf16_grads.py > gpurun_out/dbg_roundw.txt 2>&1
tail -3 gpurun_out/dbg_yfree_on.txt
