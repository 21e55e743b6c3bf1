"""Kernels of both queues around the forward -> backward turn (rocprofv3 kernel trace csv): the serial section with the loss in it."""
import sys, csv
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
loss = [i for i, r in enumerate(rows) if 'loss_kernel' in r[2]]
t0 = rows[loss[-4]][0]
for s, e, n, q in rows:
    if t0 - 350e3 <= s <= t0 + 650e3:
        print(f'q{q} {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  {(e - s) / 1e3:6.1f} us  {n[:100]}')
