#!/usr/bin/env python3
"""Headline benchmark: regression-train crops/sec, MobileNetV2 9-class, 224x224 crops, per-GPU batch 256,
bf16 activation storage (fp32 accumulate, fp32 master weights), on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

One "step" = one full train iteration of the hot path over one synthetic batch that is already resident
in HBM: forward (train-mode BatchNorm) -> fused losses l1 + 0.1*add_loss + 0.2*cross_entropy and their
gradients -> hand-derived backward -> (N > 1: bucketed RCCL all-reduce overlapped with the backward) ->
AdamW update.  Prints ONE JSON line (rank 0).  Also reported on the same line:
  roofline      the dominant kernel family (by device time, measured with HIP events on the launch
                stream inside the timed steps): algorithmic HBM bytes / measured time vs 8 TB/s
  cpu_baseline  the CPU oracle (oracle/, a torch-CPU restatement pinned to the reference) running the same
                train step on this host's cores, on a bounded sample (rank 0, N == 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]

import torch  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
MNV2_TRAIN_MB_PER_CROP = 80.66   # algorithmic bytes, bf16, fwd + dgrad + wgrad
CONV_KERNELS = ('t3d_dwconv_fwd', 't3d_dwconv_bwd', 't3d_pwconv_fwd', 't3d_pwconv_dgrad', 't3d_pwconv_wgrad',
                't3d_pwconv_dgrad_yfree', 't3d_pwconv_wgrad_yfree')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--model', default='mobilenetv2')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=16)
    ap.add_argument('--cpu-steps', type=int, default=60)
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--eval', action='store_true', help='time the inference forward (Evaluator.val_step: running BatchNorm '
                    'statistics, values-only losses + metrics) instead of the train step')
    ap.add_argument('--per-launch', action='store_true', help='print every conv launch of one step (stderr)')
    ap.add_argument('--profile-all', action='store_true', help='time every kernel family, print a table to stderr')
    return ap.parse_args()


def cpu_baseline(model, size, batch, steps, budget_s=20.0):
    """The oracle's train step (fwd + losses + autograd bwd + AdamW) on the host cores, on a BOUNDED sample:
    at most `steps` steps and ~`budget_s` seconds.  Threads are capped at 32: with every hardware thread of a
    256-thread host the small convolutions of this network oversubscribe and run ~50x slower."""
    from oracle import losses as OL
    from oracle import model as OMod
    from oracle.weights import make_inputs, make_state_dict
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = make_state_dict(model, 9)
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in sd.items()}
    opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=1e-3, weight_decay=1e-4)
    lm = OL.LossManager(OL.build(['l1', 'add_loss', 'cross_entropy']), ([1., .1], [.2]))
    imgs, gt, cats = make_inputs(batch, size, size, 9)
    mask = (torch.rand(batch, 1280) >= 0.5).float() * 2

    def step():
        kp, tg = OMod.forward(params, model, imgs, cats, train=True, num_classes=9, dropout_mask=mask)
        loss = lm.parse_losses(kp, gt, tg, cats, 0)
        opt.zero_grad()
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()                                   # warm-up (also the only sample if the host is very slow)
    warm = time.perf_counter() - t0
    done, dt = 0, 0.0
    t0 = time.perf_counter()
    while done < steps and warm + dt + (dt / done if done else warm) < budget_s:
        step()
        done += 1
        dt = time.perf_counter() - t0
    if done == 0:
        done, dt, note = 1, warm, 'the warm-up step itself (host too slow for more inside the time budget)'
    else:
        note = f'{done} train steps after 1 warm-up step'
    return dict(value=round(batch * done / dt, 2), unit='crops/s', cores=cores, kind='port',
                sample=f'{note}: {model} at batch {batch}, {size}x{size}, fp32 '
                       f'({warm + dt:.1f} s of CPU work, torch {torch.__version__} CPU kernels, {cores} threads of '
                       f'{os.cpu_count()})')


def cpu_baseline_guarded(args):
    """Runs the CPU leg in a child process with a hard wall-clock limit, so that a slow or oversubscribed host can
    never stall the benchmark (the child never touches the GPU)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--model', args.model, '--size', str(args.size),
           '--cpu-batch', str(args.cpu_batch), '--cpu-steps', str(args.cpu_steps)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=150)
        return json.loads(out.stdout.strip().split('\n')[-1])
    except Exception as e:  # noqa: BLE001  (timeout, crash, unparsable output)
        return dict(value=None, unit='crops/s', cores=min(os.cpu_count() or 1, 32), kind='port',
                    sample=f'CPU leg did not finish within its 150 s limit ({type(e).__name__})')


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.model, args.size, args.cpu_batch, args.cpu_steps)), flush=True)
        return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 or 'RANK' in os.environ:      # launched by torch.distributed.run (also with one rank)
        dist.init_process_group('nccl', device_id=dev)

    from torchdet3d import _native as N
    from torchdet3d.models.engine import Net
    from torchdet3d.parallel import GradSync

    B, S = args.batch, args.size
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    net = Net(args.model, 9, dev, dtype)
    net.reset_parameters(seed=5)
    sync = GradSync(net.gflat)
    sync.broadcast([net.flat] + [b for b in net.buffers.values()])
    if world > 1 or sync.force:
        net.grad_hook = sync.ready
    flat = torch.nn.Parameter(net.flat)          # one fused AdamW update over the flat master weights
    flat.grad = net.gflat
    try:
        opt = torch.optim.AdamW([flat], lr=1e-3, weight_decay=1e-4, fused=True)
    except Exception:                            # noqa: BLE001
        opt = torch.optim.AdamW([flat], lr=1e-3, weight_decay=1e-4)

    g = torch.Generator(device=dev).manual_seed(5 + rank)
    nb = 2                                       # synthetic batches resident in HBM, cycled
    imgs = [torch.randn(B, 3, S, S, device=dev, generator=g) for _ in range(nb)]
    gts = [torch.rand(B, 18, device=dev, generator=g) for _ in range(nb)]
    cats = [torch.randint(0, 9, (B,), device=dev, generator=g) for _ in range(nb)]
    cfg = N.LossCfg()
    cfg.c_l1, cfg.c_add, cfg.c_ce = 1.0, 0.1, 0.2
    cfg.smoothl1_beta, cfg.wing_w, cfg.wing_eps, cfg.lam_reg, cfg.lam_cls = 0.2, 5.18, 1.0, 1.0, 1.0
    out = torch.zeros(16, device=dev)
    dkp, dlg = torch.empty(B, 18, device=dev), torch.empty(B, 9, device=dev)

    def eval_step(i):
        j = i % nb
        kp, lg = net.forward(imgs[j], cats[j], train=False)
        N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp), N.ptr(gts[j]), N.ptr(lg), N.ptr(cats[j]), N.ptr(out), None, None, B, 9,
               N.stream())

    def step(i):
        if args.eval:
            return eval_step(i)
        j = i % nb
        kp, lg = net.forward(imgs[j], cats[j], train=True)
        N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp), N.ptr(gts[j]), N.ptr(lg), N.ptr(cats[j]), N.ptr(out), N.ptr(dkp),
               N.ptr(dlg), B, 9, N.stream())
        sync.start()
        net.backward(dkp, dlg)
        sync.finish()
        opt.step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up; the first half also finds the dominant kernel family: all families timed, with the weight gradients
    # on the main stream for these steps (event pairs on the second stream would also count the time a launch waits
    # for the main stream's persistent kernels to free registers, not just the kernel)
    N.timer = N.KernelTimer(None)
    side, net._side = net._side, None
    for i in range(args.warmup):
        if i == max(1, args.warmup // 2):
            fam = N.timer.summary()
            N.timer = None
            net._side = side
        step(i)
    if N.timer is not None:
        fam = N.timer.summary()
        N.timer = None
    net._side = side
    # host cost of a step: time to ENQUEUE two steps into an empty queue (inside the timed region the host runs ahead
    # until HIP's queue back-pressure stalls it, which would be measured instead)
    barrier()
    t0 = time.perf_counter()
    step(0)
    step(1)
    t_issue = (time.perf_counter() - t0) / 2
    conv = {k: v for k, v in fam.items() if k in CONV_KERNELS}
    dominant = max(conv, key=lambda k: conv[k]['ms']) if conv else None
    if args.profile_all and rank == 0:
        tot = sum(v['ms'] for v in fam.values())
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms']):
            bw = v['bytes'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] > 0 and v['bytes'] else 0
            print(f'  {k:22s} {v["launches"]:5d} launches {v["ms"]:9.3f} ms {100 * v["ms"] / tot:5.1f}%  '
                  f'{bw:6.2f} TB/s algorithmic', file=sys.stderr)

    if args.per_launch and rank == 0:
        N.timer = N.KernelTimer(set(CONV_KERNELS))
        step(0)
        for n, sg, ms, nby in N.timer.per_launch():
            print(f'  {n:20s} {str(sg):44s} {ms * 1e3:8.1f} us  {(nby or 0) / 1e6:8.1f} MB  {(nby or 0) / ms / 1e9:6.2f} TB/s',
                  file=sys.stderr)
        N.timer = None

    # ---- timed region: exactly K steps, only the dominant family carries event pairs
    N.timer = N.KernelTimer({dominant}) if dominant else None
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    tsum = N.timer.summary() if N.timer else {}
    N.timer = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = out[0].item()
    assert loss == loss, 'loss is NaN'

    if rank == 0:
        crops = B * world * args.steps / dt
        res = {
            'metric': f'regression {"eval" if args.eval else "train"} crops/sec @{S}^2 bs{B} ' + {'mobilenetv2': 'MobileNetV2', 'mobilenetv3_large': 'MobileNetV3-large', 'mobilenetv3_small': 'MobileNetV3-small'}.get(args.model, args.model), 'value': round(crops, 1), 'unit': 'crops/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{args.model} 9-class Objectron keypoint regression, '
                                   + ('inference forward (running BatchNorm statistics) + loss / metric values'
                                      if args.eval else
                                      f'train step (fwd + l1/add/CE losses + bwd + AdamW{" + RCCL grad all-reduce" if world > 1 else ""})')
                                   + f', {S}x{S} crops, per-GPU batch {B}', 'global_batch': B * world, 'parallelism': f'dp{world}',
                       'final_loss': round(loss, 5), 'host_issue_ms_per_step': round(t_issue * 1e3, 3)},
        }
        if dominant and dominant in tsum:
            d = tsum[dominant]
            ach = d['bytes'] / (d['ms'] * 1e-3) / 1e9
            # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction),
            # valid for the default workload only; not measurable from inside this process
            traffic = None
            tf = os.path.join(ROOT, 'profiles', 'r1_f_hbm_traffic_pmc.json')
            if os.path.exists(tf) and args.model == 'mobilenetv2' and S == 224 and B == 256 and args.dtype == 'bf16':
                fam_t = json.load(open(tf))['families'].get(dominant)
                if fam_t:
                    traffic = round(fam_t['hbm_bytes_per_step'] / (d['launches'] // args.steps))
            res['roofline'] = {'bound': 'hbm', 'kernel': dominant, 'achieved': round(ach, 1), 'peak': HBM_PEAK / 1e9,
                               'unit': 'GB/s', 'frac': round(ach * 1e9 / HBM_PEAK, 4), 'traffic': traffic,
                               'launches_per_step': d['launches'] // args.steps,
                               'avg_launch_us': round(1e3 * d['ms'] / d['launches'], 2),
                               'algorithmic_MB_per_step': round(d['bytes'] / args.steps / 1e6, 1)}
        if args.model == 'mobilenetv2' and S == 224 and args.dtype == 'bf16':
            per_crop = 26.89 if args.eval else MNV2_TRAIN_MB_PER_CROP     # forward / train MB per crop
            res['config']['step_hbm_roofline_frac'] = round(crops / world * per_crop * 1e6 / HBM_PEAK, 4)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline_guarded(args)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
