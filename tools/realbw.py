"""Per-launch REAL HBM bytes (every tensor a kernel must touch once) vs time, from `bench.py --per-launch` output.
Lists the launches with the largest excess over a 4.5 TB/s pace."""
import re, sys, collections
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*(t3d_\S+)\s+\((.*?)\)\s+([\d.]+) us', l)
    if not m: continue
    n, a, us = m.group(1), [int(v) for v in m.group(2).split(',')], float(m.group(3))
    if 'dwconv' in n:
        _, B, H, W, C, k, s = a[:7]
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        i, o = B * H * W * C * 2, B * Ho * Wo * C * 2
        by = i + o if 'fwd' in n else 2 * (i + o)
    else:
        if n.endswith('yfree'):
            M, HW, K, N = a[:4]
        else:
            _, M, HW, K, N = a[:5]
        if n == 't3d_pwconv_fwd': by = M * (K + N) * 2
        elif n == 't3d_pwconv_dgrad': by = M * (2 * N + 2 * K) * 2 if K > N else M * (2 * N + 2 * K) * 2
        elif n == 't3d_pwconv_dgrad_yfree': by = M * (N + 3 * K) * 2
        elif n == 't3d_pwconv_wgrad': by = M * (2 * N + K) * 2
        else: by = M * (N + 2 * K) * 2
    rows.append((n, a, us, by))
tot = collections.defaultdict(lambda: [0., 0.])
for n, a, us, by in rows:
    tot[n][0] += us; tot[n][1] += by
for n, (us, by) in tot.items(): print(f'{n:26s} {us:8.0f} us {by / 1e6:9.0f} MB  {by / us / 1e6:5.2f} TB/s real')
print('total %.0f us, %.0f MB' % (sum(v[0] for v in tot.values()), sum(v[1] for v in tot.values()) / 1e6))
ex = sorted(rows, key=lambda r: -(r[2] - r[3] / 4.5e6))[:28]
for n, a, us, by in ex: print(f'  {n[4:]:22s} {str(a):40s} {us:7.1f} us {by / 1e6:7.1f} MB {by / us / 1e6:5.2f} TB/s  excess {us - by / 4.5e6:6.1f} us')
