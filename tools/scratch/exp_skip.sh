#!/bin/bash
# (needs the throw-away hack T3D_EXP_SKIP_SMALL_WGRAD in engine._wgrad: "if M <= threshold: return" -- not in the tree)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 0 12544 50176 200704; do
  echo -n "skip wgrad of layers with M <= $v:  "; T3D_EXP_SKIP_SMALL_WGRAD=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
