# gated (squeeze-excite) against plain weight / data gradients of MobileNetV3-large's gated projection layers, kernel-only durations
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shp in "50176 196 672 112" "200704 784 120 40" "12544 49 960 160"; do
  for g in "--act relu6" "--act hswish" "--act hswish --gate"; do
    d=gpurun_out/trace_tmp; rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 tools/run_kernel.py pwwgrad $shp --reps 20 --nrep 16 $g > /dev/null 2>&1
    echo "== $shp $g"
    python3 tools/kstats.py $(find $d -name "*kernel_stats.csv") 2>/dev/null | grep -i "wgrad" | cut -c1-150
  done
done
rm -rf gpurun_out/trace_tmp
