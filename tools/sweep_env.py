"""Profiling aid: run bench.py --per-launch under several env settings and print per-family totals.
usage: sweep_env.py FAMILY 'VAR1=a,b VAR2=c,d' """
import itertools, os, re, subprocess, sys
fam = sys.argv[1]
axes = [(kv.split('=')[0], kv.split('=')[1].split(',')) for kv in sys.argv[2].split()]
for combo in itertools.product(*[v for _, v in axes]):
    env = dict(os.environ, **{k: val for (k, _), val in zip(axes, combo)})
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '2', '--warmup', '2', '--per-launch', '--no-cpu-baseline'],
                         env=env, capture_output=True, text=True)
    tot, rows = 0.0, []
    for l in (out.stderr + out.stdout).split('\n'):
        m = re.match(r'\s+(t3d_\w+)\s+\((.*?)\)\s+([\d.]+) us', l)
        if m and m.group(1) == fam:
            tot += float(m.group(3)); rows.append((m.group(2), float(m.group(3))))
    print(dict(zip([k for k, _ in axes], combo)), f'{fam}: {tot / 1e3:.2f} ms', flush=True)
    if os.environ.get('ROWS'):
        for r in rows: print('    ', r)
