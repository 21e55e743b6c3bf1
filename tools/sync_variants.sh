#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; env $1 T3D_FORCE_SYNC=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 8 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'; }
run "GPU_MAX_HW_QUEUES=4"
run "GPU_MAX_HW_QUEUES=8"
run "GPU_MAX_HW_QUEUES=8 T3D_NO_SIDE_STREAM=1"
run "GPU_MAX_HW_QUEUES=4 T3D_NO_SIDE_STREAM=1"
run "GPU_MAX_HW_QUEUES=8 OMP_NUM_THREADS=8"
