#!/bin/bash
# usage: tools/ab.sh "ENV1=.. ENV2=.." "ENV3=.." ...  -- bench.py once per environment set, one summary line each
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
out=gpurun_out/ab.txt; : > $out
for e in "$@"; do
  echo "== $e" >> $out
  env $e python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d.get('roofline', {})
print(d['ms_per_step'], d['config']['step_ms_min_med_max'], 'host', d['config']['host_issue_ms_per_step'], 'loss', d['config']['final_loss'], r.get('frac'), [(x['kernel'], x['avg_launch_us']) for x in r.get('depthwise', [])])" >> $out 2>&1
done
cat $out
