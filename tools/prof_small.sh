#!/bin/bash
# kernel-only durations (rocprofv3 kernel trace) of the pointwise kernels on the small-spatial layers
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for spec in "pwfwd 12544 49 160 960" "pwfwd 12544 49 960 160" "pwdgrad 12544 49 160 960" "pwdgrad 12544 49 960 160" "pwwgrad 12544 49 160 960" "pwwgrad 12544 49 960 160" "pwwgrad 12544 49 960 320" "pwdgrad 50176 196 576 96" "pwdgrad 50176 196 96 576" "pwwgrad 50176 196 576 96" "pwwgrad 50176 196 96 576" "pwdgrad 50176 196 384 64" "pwwgrad 50176 196 384 64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps_$i -o p -- python3 tools/run_kernel.py $spec --reps 20 --nrep 16 > /dev/null 2>&1
  echo "== $spec"
  python3 - "gpurun_out/ps_$i" <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if 'pw' in n or 'wgrad' in n or 'reduce' in n:
        print('   %-60s calls %4s avg %8.1f us' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
  rm -rf gpurun_out/ps_$i
done
