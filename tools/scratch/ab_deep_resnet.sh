#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 0 1; do
  echo -n "resnet50 T3D_PW_FRAG=$v  "; T3D_PW_FRAG=$v python bench.py --model resnet50 --batch 64 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done
for v in 0 1; do
  echo -n "mnv3_large T3D_PW_FRAG=$v  "; T3D_PW_FRAG=$v python bench.py --model mobilenetv3_large --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done
