#!/bin/bash
# Matrix-core utilisation per kernel family from rocprofv3 SQ counters (missing #5: "MFMA utilisation as a
# counter"): SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's MFMA pipe is busy, summed over the chip; = 16 x the number of
# v_mfma_f32_16x16x32_bf16 wave-instructions, MI355X_MICROARCH.md cycle constants) against the kernel's own duration (the
# --kernel-trace of the same run; under --pmc every dispatch runs alone) x 2.4 GHz x 1024 SIMDs.  (The gfx94x `MfmaUtil` formula
# divides by GRBM_GUI_ACTIVE x SIMDs; on this chip that counter comes back summed over the 8 XCDs, so it is kept as a cross-check
# only.)  One --pmc pass with --kernel-trace, no other trace option beside it.
# usage (on the GPU box): tools/pmc_mfma.sh <out.json> [bench args...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=${1:-gpurun_out/mfma_util_pmc.json}; shift
export T3D_DEVICE_WARMUP_S=0
rm -rf gpurun_out/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - "$out" "$*" <<'PY'
import sys, glob, csv, json, re, collections
out, args = sys.argv[1], sys.argv[2]
f = glob.glob('gpurun_out/pmc_mfma/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
def short(k):
    k = re.sub(r'\(anonymous namespace\)::|t3d_pw::|void ', '', k)
    m = re.match(r'(_ZN12_GLOBAL__N_1\d+)?(\w+?)(I|<|\().*', k)
    return (m.group(2) if m else k)[:60]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    per[(r['Dispatch_Id'], r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
# kernel durations of the SAME run (--kernel-trace beside --pmc): GRBM_GUI_ACTIVE on this chip comes back summed over the 8 XCDs
# (measured: ~8x duration x clock), so the utilisation is quoted against the dispatch's own duration x CLOCK_GHZ, and the
# GUI_ACTIVE / (duration x clock) ratio is kept in the file as the cross-check
CLOCK_GHZ = 2.4          # MI355X peak engine clock (MI355X_MICROARCH.md); the 2.5 PF dense bf16 peak is quoted at it
dur = {}
kt = glob.glob('gpurun_out/pmc_mfma/**/*kernel_trace.csv', recursive=True)
if kt:
    for r in csv.DictReader(open(kt[0])):
        try: dur[r['Dispatch_Id']] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        except (KeyError, ValueError): pass
fam = collections.defaultdict(lambda: dict(dispatches=0, mfma_busy=0.0, gui=0.0, sq_busy=0.0, ns=0.0))
for (d, k), c in per.items():
    e = fam[short(k)]
    e['dispatches'] += 1; e['mfma_busy'] += c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); e['gui'] += c.get('GRBM_GUI_ACTIVE', 0.0); e['sq_busy'] += c.get('SQ_BUSY_CYCLES', 0.0)
    e['ns'] += dur.get(d, 0.0)
NSIMD = 1024
res = {}
def util(e):
    cyc = e['ns'] * CLOCK_GHZ if e['ns'] > 0 else e['gui'] / 8.0
    return e['mfma_busy'] / (cyc * NSIMD) if cyc > 0 else 0.0
for k, e in sorted(fam.items(), key=lambda kv: -kv[1]['gui']):
    if e['gui'] <= 0: continue
    res[k] = dict(dispatches=e['dispatches'], kernel_ns=e['ns'], gui_active_cycles=e['gui'], mfma_busy_cycles=e['mfma_busy'],
                  gui_active_per_duration_cycle=round(e['gui'] / (e['ns'] * CLOCK_GHZ), 2) if e['ns'] > 0 else None,
                  mfma_util=round(util(e), 5))
tot = dict(mfma_busy=sum(e['mfma_busy'] for e in fam.values()), gui=sum(e['gui'] for e in fam.values()), ns=sum(e['ns'] for e in fam.values()))
tot_gui, tot_mfma = (tot['ns'] * CLOCK_GHZ if tot['ns'] > 0 else tot['gui'] / 8.0), tot['mfma_busy']
json.dump(dict(note='rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 3 --warmup 2 ' + args + '`; '
               'mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs) per kernel name (all its dispatches; durations from '
               'the --kernel-trace of the same run), i.e. the share of SIMD-cycles the matrix pipe was busy while the kernel had the GPU to itself '
               '(dispatches are serialised under --pmc); GRBM_GUI_ACTIVE is kept as a cross-check (it comes back summed over the 8 XCDs); all kernels = the same ratio over every dispatch',
               all_kernels_mfma_util=round(tot_mfma / (tot_gui * NSIMD), 5) if tot_gui else None, kernels=res), open(out, 'w'), indent=1)
for k, v in list(res.items())[:14]: print(f"{k:50s} {v['dispatches']:5d} disp  mfma_util {v['mfma_util']:.4f}")
print('all kernels:', round(tot_mfma / (tot_gui * NSIMD), 5) if tot_gui else None)
PY
rm -rf gpurun_out/pmc_mfma
