#!/bin/bash
# HBM traffic per kernel family for the default bench workload: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# SEPARATE passes (MI355X_MICROARCH.md: the HBM/rocprofv3 section), FETCH_SIZE doubled (gfx950 correction), units KB.
# usage (on the GPU box): tools/pmc_traffic.sh <out.json> [commit]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=${1:-gpurun_out/hbm_traffic_pmc.json}
STEPS=3; WARM=2
export T3D_DEVICE_WARMUP_S=0        # (the clock warm-up copies would count into the all-kernels total)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o p -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline > /dev/null 2>&1
done
python3 - "$out" $((STEPS + WARM + 2)) "${2:-?}" <<'PY'
import sys, glob, csv, json, re, collections
out, nsteps = sys.argv[1], int(sys.argv[2])
def family(k):
    if 'dw3_bwd' in k or 'dw_bwd' in k: return 't3d_dwconv_bwd'
    if 'dw3_fwd' in k or 'dw_fwd' in k: return 't3d_dwconv_fwd'
    m = re.search(r'pw_stream_kernel<([^>]*)>', k)
    if m:
        a = [v.strip() for v in m.group(1).split(',')]
        if len(a) > 4 and a[4] == 'true': return 't3d_pwconv_dgrad_yfree'
        return 't3d_pwconv_dgrad' if a[2] == 'true' else 't3d_pwconv_fwd'
    m = re.search(r'pw_wgrad_tr_kernel<([^>]*)>', k)
    if m:
        a = [v.strip() for v in m.group(1).split(',')]
        return 't3d_pwconv_wgrad_yfree' if len(a) > 6 and a[6] == 'true' else 't3d_pwconv_wgrad'
    if 'wgrad_reduce' in k or 'yfree_combine' in k or 'yfree_prep' in k: return 'pw_wgrad_reduce+yfree_small'
    if 'pw_gemm' in k or 'pwconv' in k or 'pw_' in k: return 'pw_other'
    return None
fam = collections.defaultdict(lambda: dict(fetch_kb=0.0, write_kb=0.0, dispatches_fetch_pass=0, dispatches_write_pass=0))
tot = dict(FETCH_SIZE=0.0, WRITE_SIZE=0.0)
for c, key, cnt in (('FETCH_SIZE', 'fetch_kb', 'dispatches_fetch_pass'), ('WRITE_SIZE', 'write_kb', 'dispatches_write_pass')):
    f = glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % c, recursive=True)
    for r in csv.DictReader(open(f[0])):
        v = float(r['Counter_Value']); tot[c] += v
        fm = family(r['Kernel_Name'])
        if fm: fam[fm][key] += v; fam[fm][cnt] += 1
for v in fam.values(): v['hbm_bytes_per_step'] = round((2 * v['fetch_kb'] + v['write_kb']) * 1024 / nsteps)
# the depthwise kernels one by one, under the names bench.py's roofline block uses (rocprof kernel names)
kern = collections.defaultdict(lambda: dict(fetch_kb=0.0, write_kb=0.0, dispatches=0))
for c, key in (('FETCH_SIZE', 'fetch_kb'), ('WRITE_SIZE', 'write_kb')):
    f = glob.glob('gpurun_out/pmc_%s/**/*counter_collection.csv' % c, recursive=True)
    for r in csv.DictReader(open(f[0])):
        m = re.search(r'(dw3_bwd2_kernel|dw3_bwd_s2_kernel|dw3_fwd2_kernel|dw3_fwd_kernel|dw3_bwd_s1_kernel)', r['Kernel_Name'])
        if m:
            kern[m.group(1)][key] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE': kern[m.group(1)]['dispatches'] += 1
for v in kern.values(): v['hbm_bytes_per_step'] = round((2 * v['fetch_kb'] + v['write_kb']) * 1024 / nsteps)
import hashlib, os
def src_hash():
    h = hashlib.sha256()
    d = os.path.join('3d-object-detection.pytorch_amd', 'csrc')
    for f in sorted(x for x in os.listdir(d) if x.startswith('dwconv3') and x.endswith('_stream.hip')):
        h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]
def conv_hash():          # every source of the library: bench.py quotes the per-family ratios only for the kernels this pass saw
    h = hashlib.sha256()
    d = os.path.join('3d-object-detection.pytorch_amd', 'csrc')
    for f in sorted(x for x in os.listdir(d) if x.endswith(('.hip', '.h'))):
        h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]
res = dict(dw3_source_sha256=src_hash(), conv_source_sha256=conv_hash(), note='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 2` (7 steps: 2 warm-up, 2 host-issue probes, 3 timed), '
                'MobileNetV2 224^2 B=256 bf16; FETCH_SIZE doubled per the gfx950 correction; per-family sums (tools/pmc_traffic.sh)',
           commit=(sys.argv[3] if len(sys.argv) > 3 else '?'), kernels=kern,
           steps=nsteps, all_kernels_hbm_bytes_per_step=round((2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024 / nsteps), families=fam)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: v['hbm_bytes_per_step'] for k, v in fam.items()}), res['all_kernels_hbm_bytes_per_step'])
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
