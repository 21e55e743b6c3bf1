#!/bin/bash
# Upper bounds on what removing a family of launches can buy: bench.py with those entry points skipped (garbage
# numerics, valid step time).  Output: gpurun_out/ablate.txt
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
out=gpurun_out/ablate.txt; : > $out
run() { echo "== $1" >> $out; env $2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = d.get('roofline', {})
print(d['ms_per_step'], d['config']['step_ms_min_med_max'], r.get('frac'), [(x['kernel'], x['avg_launch_us']) for x in r.get('depthwise', [])])" >> $out 2>&1; }
run baseline T3D_X=0
run no_finalize T3D_ABLATE=t3d_bn_finalize,t3d_bn_bwd_finalize
run no_bn_apply T3D_ABLATE=t3d_bn_apply
run no_wgrad T3D_ABLATE=t3d_pwconv_wgrad,t3d_pwconv_wgrad_yfree,t3d_head_bwd_weights
run no_side_stream T3D_NO_SIDE_STREAM=1
run no_dw_bwd T3D_ABLATE=t3d_dwconv_bwd
run no_dw_fwd T3D_ABLATE=t3d_dwconv_fwd
run no_dgrad T3D_ABLATE=t3d_pwconv_dgrad,t3d_pwconv_dgrad_yfree,t3d_pwconv_yfree_prep
run no_pw_fwd T3D_ABLATE=t3d_pwconv_fwd
run no_finalize_no_apply T3D_ABLATE=t3d_bn_finalize,t3d_bn_bwd_finalize,t3d_bn_apply
cat $out
