#!/bin/bash
# everything the round's profiles/ are made of, in one GPU call.  usage: tools/profile_round.sh <tag> <commit>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-r6_g}; commit=${2:-?}; o=gpurun_out/$tag; mkdir -p $o
python3 bench.py --steps 30 --warmup 8 > $o/bench_bf16.json 2> $o/bench_bf16.err
python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-all > /dev/null 2> $o/bench_bf16_families.txt
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --per-launch > /dev/null 2> $o/bench_bf16_per_launch.txt
python3 bench.py --steps 10 --warmup 3 --dtype f32 --no-cpu-baseline > $o/bench_f32.json 2> /dev/null
python3 bench.py --steps 30 --warmup 8 --eval --no-cpu-baseline > $o/bench_eval.json 2> /dev/null
python3 bench.py --steps 30 --warmup 8 --eval --eval-dtype bf16 --no-cpu-baseline > $o/bench_eval_bf16_optin.json 2> /dev/null
python3 bench.py --steps 30 --warmup 8 --eval --eval-sync --no-cpu-baseline > $o/bench_eval_val_step.json 2> /dev/null
python3 bench.py --steps 30 --warmup 8 --eval --eval-sync --eval-dtype bf16 --no-cpu-baseline > $o/bench_eval_bf16_optin_val_step.json 2> /dev/null
T3D_F32_TILED=1 python3 bench.py --steps 30 --warmup 8 --eval --no-cpu-baseline > $o/bench_eval_f32_tiled_kernel.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --model mobilenetv3_large --no-cpu-baseline > $o/bench_mnv3_large.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --model mobilenetv3_small --no-cpu-baseline > $o/bench_mnv3_small.json 2> /dev/null
python3 bench.py --steps 30 --warmup 8 --eval --eval-dtype f16 --no-cpu-baseline > $o/bench_eval_f16.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 --model resnet50 --batch 64 --no-cpu-baseline > $o/bench_resnet50.json 2> /dev/null
python3 bench.py --steps 30 --warmup 8 --engine --no-cpu-baseline > $o/bench_bf16_engine_loop.json 2> /dev/null
T3D_STEP_PLAN=0 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline > $o/bench_bf16_direct_step.json 2> /dev/null
T3D_PLAN_HANDOFF=0 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline > $o/bench_bf16_event_forks.json 2> /dev/null
T3D_IMPLICIT3=1 python3 bench.py --steps 20 --warmup 5 --model resnet50 --batch 64 --no-cpu-baseline > $o/bench_resnet50_implicit3x3.json 2> /dev/null
python3 tools/time_expdw.py > $o/expdw_fused_forward_timings.txt 2>&1
python3 tools/time_pw_f32.py > $o/pw_f32_reg_vs_tiled.txt 2>&1
bash tools/time_kernels.sh > $o/isolated_kernel_timings.txt 2>&1
python3 tools/bench_two_stage.py --detector 2> /dev/null | tail -1 > $o/two_stage_pipeline_bench.jsonl
python3 tools/bench_two_stage.py --dets 64 2> /dev/null | tail -1 >> $o/two_stage_pipeline_bench.jsonl
# rocprofv3 kernel trace + stats of the default bench (the program itself after `--`)
d=gpurun_out/trace_$tag; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_bf16_under_rocprof.json 2> /dev/null
find $d -name "*kernel_stats.csv" -exec cp {} $o/kernel_stats.csv \;
python3 tools/trace_streams.py $(find $d -name "*kernel_trace.csv") > $o/stream_timeline.txt 2>&1
python3 tools/trace_summary.py $(find $d -name "*kernel_trace.csv") > $o/step_kernel_table.txt 2>&1
rm -rf $d
# round 6: kernel stats of the reference-pinned model and of ResNet-50 (missing #4), matrix-core utilisation from the
# SQ counters (missing #5), isolated timings of the pointwise backward and of the small-plane depthwise kernels
for m in mobilenetv3_large resnet50; do
  d=gpurun_out/trace_${tag}_$m; rm -rf $d
  extra=""; [ $m = resnet50 ] && extra="--batch 64"
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --model $m $extra --steps 12 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
  find $d -name "*kernel_stats.csv" -exec cp {} $o/${m}_kernel_stats.csv \;
  python3 tools/trace_summary.py $(find $d -name "*kernel_trace.csv") > $o/${m}_step_kernel_table.txt 2>&1
  rm -rf $d
done
bash tools/pmc_mfma.sh $o/mfma_util_pmc.json > $o/mfma_util_pmc.log 2>&1
bash tools/pmc_mfma.sh $o/mfma_util_pmc_resnet50.json --model resnet50 --batch 64 > $o/mfma_util_pmc_resnet50.log 2>&1
bash tools/time_pw_bwd.sh $o/pw_bwd_isolated.txt > /dev/null 2>&1
bash tools/time_dw5.sh > $o/dw5_isolated.txt 2>&1
bash tools/time_dw3_small.sh > $o/dw3_small_planes_tile_vs_walk.txt 2>&1
# round 6b: gated against plain weight gradients (kernel-only durations), registers / scratch of the kernels each model launches,
# what runs beside the gate's backward
bash tools/time_gated_pw.sh > $o/gated_pw_wgrad_isolated.txt 2>&1
bash tools/trace_resources.sh > $o/launched_kernel_resources.txt 2>&1
bash tools/trace_resources.sh --model mobilenetv3_large > $o/mobilenetv3_large_launched_kernel_resources.txt 2>&1
bash tools/trace_resources.sh --model resnet50 --batch 64 > $o/resnet50_launched_kernel_resources.txt 2>&1
bash tools/trace_overlap.sh "se_slice_kernel<" --model mobilenetv3_large > $o/mobilenetv3_large_gate_backward_overlap.txt 2>&1
# HBM traffic (separate --pmc passes)
bash tools/pmc_traffic.sh $o/hbm_traffic_pmc.json $commit > $o/pmc_traffic.log 2>&1
tail -2 $o/pmc_traffic.log
tail -1 $o/bench_bf16.json | cut -c1-400
