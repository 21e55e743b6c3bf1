#!/bin/bash
# usage: tools/isa.sh <csrc file> <mangled-name substring> [out.s]   -> kernel ISA + resource usage
f=/root/repo/3d-object-detection.pytorch_amd/csrc/$1; out=${3:-/tmp/kern.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/full.s $f 2>/dev/null || { echo compile failed; exit 1; }
L=$(grep -n "^_Z[A-Za-z0-9_]*$2[A-Za-z0-9_]*:" /tmp/full.s | head -1 | cut -d: -f1)
E=$(awk -v s=$L 'NR>s && /^\.Lfunc_end/ {print NR; exit}' /tmp/full.s)
sed -n "${L},${E}p" /tmp/full.s > $out
awk -v s=$E 'NR>s && NR<s+60' /tmp/full.s | grep "NumVgprs\|Occupancy\|ScratchSize\|NumSgprs" 
echo "lines: $(wc -l < $out)"
