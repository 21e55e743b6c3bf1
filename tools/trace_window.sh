#!/bin/bash
# rocprofv3 kernel trace of a short bench; prints every kernel of both queues in a window of one step.
# usage: tools/trace_window.sh <anchor kernel substring> <from us> <to us> [bench args]     (times relative to the anchor's start)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
anchor=$1; t0=$2; t1=$3; shift 3
d=gpurun_out/trace_tmp; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - "$anchor" $t0 $t1 $(find $d -name "*kernel_trace.csv") <<'PY'
import csv, sys
anchor, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id'], r['Grid_Size_X']) for r in csv.DictReader(open(sys.argv[4])))
ad = [i for i, r in enumerate(rows) if anchor in r[2]]
ref = rows[ad[len(ad) // 2]][0]
for s, e, n, q, g in rows:
    if ref + t0 * 1e3 <= s <= ref + t1 * 1e3:
        print('q%s %8.1f .. %8.1f  %6.1f us  g=%-7s %s' % (q, (s - ref) / 1e3, (e - ref) / 1e3, (e - s) / 1e3, g, n[:80].replace('(anonymous namespace)::', '')))
PY
rm -rf $d
