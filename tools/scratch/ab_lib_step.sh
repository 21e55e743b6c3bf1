#!/bin/bash
# usage: ab_lib_step.sh <other .so>   (A/B of the headline step: in-tree library vs another build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do
  echo -n "in-tree  "; python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  echo -n "$1  "; T3D_LIB=$1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
