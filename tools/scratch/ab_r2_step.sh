#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2 3; do
  echo -n "R=1  "; python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
  echo -n "R=2  "; T3D_EXP_R2=1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done
