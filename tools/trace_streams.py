"""Per-queue timeline summary of one training step from a rocprofv3 kernel trace csv (tools/trace_step.sh keeps it)."""
import sys, csv, collections
f = sys.argv[1]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in csv.DictReader(open(f))]
rows.sort()
loss = [i for i, r in enumerate(rows) if 'loss_kernel' in r[2]]
# one full step: loss -> backward -> optimizer -> next forward -> loss; the shortest of the last ones (the bench's
# host-issue probe steps at the end contain synchronisation gaps)
k = min(range(max(1, len(loss) - 8), len(loss)), key=lambda j: rows[loss[j]][0] - rows[loss[j - 1]][0])
i0, i1 = loss[k - 1], loss[k]
seg = rows[i0:i1]
t0 = seg[0][0]
qs = collections.defaultdict(list)
for s, e, n, q in seg: qs[q].append((s, e, n))
print('step %.1f us' % ((seg[-1][1] - t0) / 1e3))
for q, v in qs.items():
    busy = sum(e - s for s, e, _ in v)
    print('queue %s: %d kernels, busy %.1f us, first %.1f last-end %.1f' % (q, len(v), busy / 1e3, (v[0][0] - t0) / 1e3, (max(e for _, e, _ in v) - t0) / 1e3))
# timeline in 250us buckets: which queue is active
main = max(qs, key=lambda q: len(qs[q]))
side = [q for q in qs if q != main]
if side:
    sv = qs[side[0]]
    print('side stream span %.1f .. %.1f us' % ((sv[0][0] - t0) / 1e3, (sv[-1][1] - t0) / 1e3))
    # main-stream kernels that start after the side stream's last end, and the gap where only side runs
    mv = qs[main]
    # find backward end on main: the last kernel before the optimizer (multi_tensor_apply)
    opt = [s for s, e, n in mv if 'multi_tensor' in n]
    if opt:
        print('optimizer starts at %.1f us' % ((opt[0] - t0) / 1e3))
    # intervals where main is idle but side is busy
    idle = 0; prev = mv[0][1]
    for s, e, n in mv[1:]:
        if s > prev:
            # overlap of (prev, s) with side kernels
            ov = sum(max(0, min(s, e2) - max(prev, s2)) for s2, e2, _ in sv)
            idle += ov
        prev = max(prev, e)
    print('main idle while side busy: %.1f us' % (idle / 1e3))
    last = sorted(sv, key=lambda r: r[1])[-6:]
    for s, e, n in last: print('   side tail: %.1f..%.1f %s' % ((s - t0) / 1e3, (e - t0) / 1e3, n[:70]))
    lastm = [r for r in mv if r[0] < (opt[0] if opt else 1 << 62)][-6:]
    for s, e, n in lastm: print('   main tail: %.1f..%.1f %s' % ((s - t0) / 1e3, (e - t0) / 1e3, n[:70]))
