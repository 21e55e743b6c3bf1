"""Kernels of both queues around the end of the backward pass (rocprofv3 kernel trace csv): who waits for whom before the optimizer."""
import sys, csv
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
loss = [i for i, r in enumerate(rows) if 'loss_kernel' in r[2]]
k = min(range(max(1, len(loss) - 8), len(loss)), key=lambda j: rows[loss[j]][0] - rows[loss[j - 1]][0])
seg = rows[loss[k - 1]:loss[k]]
t0 = seg[0][0]
opt = [s for s, e, n, q in seg if 'adamw' in n]
tend = opt[0] if opt else seg[-1][1]
for s, e, n, q in seg:
    if tend - 900e3 <= s <= tend + 100e3:
        print(f'q{q} {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  {(e - s) / 1e3:6.1f} us  {n[:90]}')
