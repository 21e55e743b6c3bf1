#!/bin/bash
# usage: fam.sh "<ENV=..>" -> per-family serial times of one bench step (no side stream) + the real step time
cd $GRAFT_REPO_ROOT
for e in $1; do export $e; done
echo "== [$1]"
T3D_NO_SIDE_STREAM=1 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --per-launch 2>&1 | grep -E "pwconv|dwconv" | python3 -c "
import sys,re,collections
t=collections.defaultdict(float)
for l in sys.stdin:
    m=re.match(r'\s*(\S+)\s+\((.*?)\)\s+([\d.]+) us',l)
    if m: t[m.group(1)]+=float(m.group(3))
print('   '+'  '.join('%s %.0f'%(k[4:],v) for k,v in t.items()))"
python bench.py --steps 20 --warmup 8 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
