#!/bin/bash
# per kernel of one bench run: launches, average duration, registers, scratch (spills), LDS, workgroup size -- from the rocprofv3 kernel
# trace's own columns.  Spilling or one-wave-per-SIMD kernels that matter show up at the top.  usage: tools/trace_resources.sh [bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/trace_tmp; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - $(find $d -name "*kernel_trace.csv") <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    a = agg.setdefault(k, [0, 0.0, r])
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
g = lambda r, *ks: next((r[k] for k in ks if k in r), '?')
print('   total us  calls   avg us  vgpr agpr scratch    lds  wg   kernel')
for k, (n, t, r) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print('%11.0f %6d %8.1f  %4s %4s %7s %6s %4s  %s' % (t, n, t / n, g(r, 'VGPR_Count', 'Arch_VGPR_Count'), g(r, 'Accum_VGPR_Count'), g(r, 'Scratch_Size', 'Private_Segment_Size'),
          g(r, 'LDS_Block_Size', 'Group_Segment_Size'), g(r, 'Workgroup_Size_X', 'Workgroup_Size'), k[:100]))
PY
rm -rf $d
