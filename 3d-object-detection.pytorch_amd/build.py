"""Builds libt3d_hip.so (the C-ABI HIP library for gfx950) in-tree with hipcc.

No torch types cross the boundary, so this is a plain `hipcc -shared` build, not a
torch extension.  Incremental: a source is recompiled only when it or a header is
newer than its object.  Usage:  python build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INC = os.path.join(os.path.dirname(HERE), 'include')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libt3d_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wno-unused-result',
         *os.environ.get('HIPCC_EXTRA', '').split()]      # e.g. -DT3D_PW_TRACE for tools/pw_trace.sh (debug builds only)


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
           [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith('.h')]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + '.o')
        # (a unit that is another unit compiled for a second storage type -- `#include "x.hip"` -- follows that source too)
        incs = [os.path.join(CSRC, m) for m in __import__('re').findall(r'#include "(\w+\.hip)"', open(src).read())]
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs + incs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([HIPCC, *FLAGS, '-c', src, '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
        if os.environ.get('T3D_BUILD_WARNINGS') and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + '.o') for s in srcs]
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'built {LIB} ({len(jobs)} recompiled of {len(srcs)})')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
