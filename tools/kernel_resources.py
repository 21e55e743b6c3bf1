"""Register / LDS / scratch use of every kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950), one line each.
usage: python tools/kernel_resources.py <file.hip> [name substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-c', src, '-o', '/dev/null',
       '-Rpass-analysis=kernel-resource-usage']
cache = "/tmp/kres_" + os.path.basename(src) + ".txt"
if os.path.exists(cache) and os.path.getmtime(cache) > os.path.getmtime(os.path.join(ROOT, "3d-object-detection.pytorch_amd", "csrc", src)):
    err = open(cache).read()
else:
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.join(ROOT, "3d-object-detection.pytorch_amd", "csrc")).stderr
    open(cache, "w").write(err)
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r'remark: .*?(Function Name|Name): (\S+)', line)
    if m:
        cur = m.group(2); rows[cur] = {}
        continue
    m = re.search(r'remark: .*?\s+([A-Za-z ]+\w)(?: \[bytes/lane\]| \[bytes/block\]| \[waves/SIMD\])?: (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
dem = subprocess.run(['c++filt'], input='\n'.join(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, name in zip(rows, dem):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    if sub in name:
        r = rows[mangled]
        print(f"vgpr {r.get('VGPRs', -1):3d} agpr {r.get('AGPRs', -1):3d} sgpr {r.get('SGPRs', -1):3d} scratch {r.get('ScratchSize', -1):4d} occ {r.get('Occupancy', -1)} lds {r.get('LDS Size', -1):6d}  {name[:120]}")
