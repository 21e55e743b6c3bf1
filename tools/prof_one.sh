#!/bin/bash
# usage: prof_one.sh "<ENV=.. ENV=..>" <run_kernel args...>  -> kernel-only average durations (rocprofv3 kernel trace)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
envs="$1"; shift
d=gpurun_out/po_$$
for e in $envs; do export $e; done
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 tools/run_kernel.py "$@" --reps 20 --nrep 16 > /dev/null 2>&1
echo "== [$envs] $@"
python3 - "$d" <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if any(t in n for t in ('pw_', 'wgrad', 'reduce', 'dw3', 'dw_')):
        print('   %-70s calls %4s avg %8.1f us' % (n[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf $d
