# SQ issue counters of the s=1 depthwise backward (112x112x32, B = 256), one rocprofv3 --pmc pass per counter group
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU"; do
  d=gpurun_out/pmc_sq; rm -rf $d
  rocprofv3 --pmc $grp --output-format csv -d $d -o p -- python3 tools/run_kernel.py dwbwd 256 112 112 32 3 1 --reps 3 --nrep 16 > /dev/null 2>&1
  python3 - $(find $d -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'dw3_bwd2' in r['Kernel_Name']:
        a = agg[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(agg.items()): print('%-24s per launch %.4g' % (k, v / n))
PY
done
rm -rf gpurun_out/pmc_sq
