"""Where one training step's time is, from a rocprofv3 kernel trace csv: per queue busy / idle, and per kernel name the
launches, device time and the idle gap in front of each launch on its queue.  usage: trace_summary.py <kernel_trace.csv>"""
import sys, csv, collections, re
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
loss = [i for i, r in enumerate(rows) if 'loss_kernel' in r[2]]
k = min(range(max(1, len(loss) - 8), len(loss)), key=lambda j: rows[loss[j]][0] - rows[loss[j - 1]][0])
seg = rows[loss[k - 1]:loss[k]]
t0, t1 = seg[0][0], seg[-1][1]
print('step %.1f us, %d kernels' % ((t1 - t0) / 1e3, len(seg)))
qs = collections.defaultdict(list)
for s, e, n, q in seg:
    qs[q].append((s, e, n))


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    return n.split('(')[0][:64]


for q, v in sorted(qs.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in v)
    print('\nqueue %s: %d kernels, busy %.1f us, span %.1f..%.1f' % (q, len(v), busy / 1e3, (v[0][0] - t0) / 1e3, (max(e for _, e, _ in v) - t0) / 1e3))
    agg = collections.defaultdict(lambda: [0, 0, 0])
    prev = None
    for s, e, n in v:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
        if prev is not None and s > prev:
            a[2] += s - prev
        prev = e if prev is None else max(prev, e)
    print('   %-66s %5s %9s %8s %9s' % ('kernel', 'n', 'busy us', 'avg us', 'gap-before us'))
    for n, (c, b, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print('   %-66s %5d %9.1f %8.1f %9.1f' % (n, c, b / 1e3, b / c / 1e3, g / 1e3))
    print('   total gaps on this queue: %.1f us' % (sum(a[2] for a in agg.values()) / 1e3))
