#!/bin/bash
# kernel trace of the bench with the RCCL gradient exchange forced on one rank: where does the multi-GPU path lose time?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 T3D_FORCE_SYNC=1
d=gpurun_out/trace_sync
rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --gpus 1 --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/trace_sync_bench.json 2> gpurun_out/trace_sync_err.txt
tail -1 gpurun_out/trace_sync_bench.json | cut -c1-200
python3 - $d <<'PY'
import sys,glob,csv,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],r.get('Queue_Id','0')) for r in csv.DictReader(open(f))]
rows.sort()
loss=[i for i,r in enumerate(rows) if 'loss_kernel' in r[2]]
i0,i1=loss[-3],loss[-2]
seg=rows[i0:i1]; t0=seg[0][0]
print('step %.1f us'%((seg[-1][1]-t0)/1e3))
qs=collections.defaultdict(list)
for s,e,n,q in seg: qs[q].append((s,e,n))
for q,v in qs.items(): print('queue',q,len(v),'kernels busy %.1f us span %.1f..%.1f'%(sum(e-s for s,e,_ in v)/1e3,(v[0][0]-t0)/1e3,(max(e for _,e,_ in v)-t0)/1e3))
# union busy and biggest idle gaps
ev=sorted((s,e,n) for s,e,n,_ in seg)
cur=ev[0][1]; gaps=[]
for s,e,n in ev[1:]:
    if s>cur: gaps.append((s-cur,(cur-t0)/1e3,n[:60]))
    cur=max(cur,e)
print('idle total %.1f us'%(sum(g[0] for g in gaps)/1e3))
for g in sorted(gaps,reverse=True)[:12]: print('  gap %.1f us at %.1f before %s'%(g[0]/1e3,g[1],g[2]))
for s,e,n,q in seg:
    if 'ccl' in n.lower() or 'Reduce' in n and 'nccl' in n.lower(): print('  rccl: %.1f..%.1f q%s %s'%((s-t0)/1e3,(e-t0)/1e3,q,n[:60]))
PY
rm -rf $d
