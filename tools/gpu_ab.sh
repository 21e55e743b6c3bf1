#!/bin/bash
# usage: gpu_ab.sh "ENV1=.. ENV2=.." ...   -> one bench line per environment setting ("" = defaults)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for e in "$@"; do
  env $e python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('[$e]', d['value'], d['ms_per_step'], d['config']['final_loss'], [ (r['kernel'], r['ms_per_step']) for r in d['roofline']['depthwise']])"
done
