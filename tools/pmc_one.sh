#!/bin/bash
# usage: pmc_one.sh "<COUNTER COUNTER ..>" <run_kernel args...>   -> per-kernel average of each counter (one pass per counter)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ctrs="$1"; shift
for c in $ctrs; do
  d=gpurun_out/pmc_$$_$c
  rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 tools/run_kernel.py "$@" --reps 5 --nrep 16 > /dev/null 2>&1
  python3 - "$d" "$c" <<'PY'
import sys,glob,csv,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
if not f: print(sys.argv[2],'no output'); sys.exit()
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    acc[(r['Kernel_Name'][:50], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k,c),v in acc.items():
    if any(t in k for t in ('dw3','dw_','pw_','wgrad')): print('%-52s %-22s n=%3d avg %.4g' % (k,c,len(v),sum(v)/len(v)))
PY
  rm -rf $d
done
