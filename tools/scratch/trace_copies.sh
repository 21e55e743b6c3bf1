#!/bin/bash
# kernel + memory-copy trace of a short bench: every kernel and every copy in a window around the loss kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/trace_tmp2; rm -rf $d
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $d -o t -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
ls $d/*/ | head
python3 - $(find $d -name "*kernel_trace.csv") $(find $d -name "*memory_copy_trace.csv") <<'PY'
import csv, sys
k = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q' + r['Queue_Id'] + ' ' + r['Kernel_Name'][:70]) for r in csv.DictReader(open(sys.argv[1])))
c = []
if len(sys.argv) > 2:
    rows = list(csv.DictReader(open(sys.argv[2])))
    print(rows[0].keys() if rows else 'no copies')
    for r in rows:
        c.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '?') + ' ' + r.get('Size', r.get('Bytes', '?'))))
ad = [i for i, r in enumerate(k) if 'loss_kernel' in r[2]]
ref = k[ad[len(ad) // 2]][0]
for s, e, n in sorted(k + c):
    if ref - 100e3 <= s <= ref + 400e3:
        print('%9.1f .. %9.1f %7.1f us  %s' % ((s - ref) / 1e3, (e - ref) / 1e3, (e - s) / 1e3, n.replace('(anonymous namespace)::', '')))
PY
rm -rf $d
