#!/bin/bash
# kernel trace of the default bench (one GPU): per-kernel totals per step + idle-gap analysis of the timed region
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/trace
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/trace_bench.json 2> gpurun_out/trace_err.txt
tail -1 gpurun_out/trace_bench.json
python3 - $d <<'PY'
import sys,glob,csv,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# last 10 steps: find the step period from the loss kernel
loss=[i for i,r in enumerate(rows) if 'loss' in r[2]]
i0,i1=loss[-9],loss[-1]     # 8 full steps
seg=rows[i0:i1]
T=(seg[-1][1]-seg[0][0])/8/1e3
busy=0; cur_s,cur_e=seg[0][0],seg[0][1]
for s,e,_ in seg[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print('step %.1f us; GPU busy (union of kernels) %.1f us; idle %.1f us'%(T,busy/8/1e3,T-busy/8/1e3))
tot=collections.defaultdict(float); cnt=collections.Counter()
for s,e,n in seg: tot[n[:60]]+=(e-s)/8/1e3; cnt[n[:60]]+=1
for k,v in sorted(tot.items(),key=lambda kv:-kv[1])[:28]: print('%-62s %5.1f/step %8.1f us'%(k,cnt[k]/8,v))
print('sum of kernel durations %.1f us'%sum(tot.values()))
PY
cp $d/*/*kernel_stats.csv gpurun_out/trace_kernel_stats.csv 2>/dev/null || find $d -name "*kernel_stats.csv" -exec cp {} gpurun_out/trace_kernel_stats.csv \;
python3 tools/trace_streams.py $(find $d -name "*kernel_trace.csv") | tee gpurun_out/trace_streams.txt; rm -rf $d
