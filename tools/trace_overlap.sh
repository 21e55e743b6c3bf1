#!/bin/bash
# For every launch of a kernel (substring) in one bench run: its duration and the kernels of the OTHER queue that were resident while
# it ran (rocprofv3 kernel trace).  Tells a small critical-stream kernel's waiting-for-a-slot time from its own work.
# usage: tools/trace_overlap.sh <kernel substring> [bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
anchor=$1; shift
d=gpurun_out/trace_tmp; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - "$anchor" $(find $d -name "*kernel_trace.csv") <<'PY'
import csv, sys
anchor = sys.argv[1]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', ''), r['Queue_Id'], r['Grid_Size_X'],
               r.get('VGPR_Count', r.get('Arch_VGPR_Count', '?')), r.get('LDS_Block_Size', '?'))
              for r in csv.DictReader(open(sys.argv[2])))
hits = [r for r in rows if anchor in r[2]]
hits = hits[len(hits) // 2: len(hits) // 2 + 24]
for s, e, n, q, g, v, l in hits:
    print('%-44s q%s %7.1f us  grid %s' % (n[:44], q, (e - s) / 1e3, g))
    for s2, e2, n2, q2, g2, v2, l2 in rows:
        if q2 != q and s2 < e and e2 > s:
            print('      beside q%s %-70s [%8.1f .. %8.1f] of [0 .. %6.1f]  %6.1f us grid %s vgpr %s lds %s' % (q2, n2[:70], (s2 - s) / 1e3, (e2 - s) / 1e3, (e - s) / 1e3, (e2 - s2) / 1e3, g2, v2, l2))
PY
rm -rf $d
