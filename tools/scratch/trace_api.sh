#!/bin/bash
# kernel trace + HIP runtime API trace: for every kernel of a window, when the host enqueued it and when it started
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/trace_tmp3; rm -rf $d
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $d -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
find $d -name "*.csv" | head
python3 - $(find $d -name "*kernel_trace.csv") $(find $d -name "*hip_api_trace.csv") <<'PY'
import csv, sys
k = list(csv.DictReader(open(sys.argv[1])))
api = {r['Correlation_Id']: r for r in csv.DictReader(open(sys.argv[2]))}
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id'], r['Correlation_Id']) for r in k)
ad = [i for i, r in enumerate(rows) if 'adamw' in r[2] and r[3] != rows[0][3]]
ad = [i for i, r in enumerate(rows) if 'adamw' in r[2]]
ref = rows[ad[len(ad) // 2]][0]
for s, e, n, q, c in rows:
    if ref - 1100e3 <= s <= ref + 500e3 and not any(t in n for t in ('reduce', 'fillBuffer', 'combine')):
        a = api.get(c)
        enq = (int(a['Start_Timestamp']) - ref) / 1e3 if a else float('nan')
        print('q%s start %8.1f dur %6.1f  enqueued %9.1f (%s)  %s' % (q, (s - ref) / 1e3, (e - s) / 1e3, enq, a['Function'] if a else '?', n[:50].replace('(anonymous namespace)::', '')))
PY
rm -rf $d
