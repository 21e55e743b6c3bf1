#!/bin/bash
# usage: prof_model.sh <model>  -> per-step kernel totals of the bench for that model (rocprofv3 kernel trace)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/pm_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --model $1 --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 - $d <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
steps=15
for r in rows[:45]:
    print('%-92s %5.1f/step %7.3f ms avg %7.1f us'%(r['Name'][:92], int(r['Calls'])/steps, float(r['TotalDurationNs'])/1e6/steps, float(r['AverageNs'])/1e3))
print('sum', sum(float(r['TotalDurationNs']) for r in rows)/1e6/steps)
PY
rm -rf $d
