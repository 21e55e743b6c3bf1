cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/trace_tmp; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python3 - $(find $d -name "*kernel_trace.csv") <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
agg = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name']
    if 'wgrad_reduce' in k or 'yfree_combine' in k:
        key = ('reduce' if 'reduce' in k else 'combine', r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Grid_Size_Y'))
        agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), 'avg %.1f us' % (sum(v) / len(v)), 'min %.1f' % min(v), 'total %.0f' % sum(v))
PY
rm -rf $d
