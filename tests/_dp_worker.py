"""Worker of tests/test_gpu_data_parallel.py (one rank; launched by `python -m torch.distributed.run`).

mode 'engine2': TWO ranks that share GPU 0 (backend gloo: RCCL refuses two ranks on one device; GradSync stages the
  collectives through the host there).  Each rank builds the model through the API (`build_model(cfg).to('cuda')` -- the
  gradient exchange attaches itself because WORLD_SIZE > 1), runs ONE train step of its own shard through
  `Trainer.train_step`, and checks ("oracle for N ranks = CPU restatement run per shard with
  averaged grads"): the exchanged fp32 gradient == mean over ranks of the ORACLE's per-shard gradient (per-rank BatchNorm
  statistics, mean losses), and all ranks hold identical weights after the optimizer step.
mode 'main1': ONE rank over RCCL with T3D_FORCE_SYNC=1 replays scripts/main.py's flow (tests/test_boundary_main.py
  `_replay_main`, unchanged) -- `build_model` joins the process group and attaches the exchange by itself.
mode 'main2': TWO ranks sharing GPU 0 (gloo) replay scripts/main.py's flow unchanged (VERDICT r3 missing #1 / ADVICE): every
  rank must train on its OWN share of each global batch (main.py:60-61 scatters one batch over the replicas), rank 0 alone
  writes the checkpoint / log file / scalars, `Evaluator.val` returns the SAME all-reduced metrics on both ranks and those
  equal a single-process pass over the whole validation set, and the ranks end with identical weights."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402


def engine2():
    dist.init_process_group('gloo')          # before build_model: it then leaves the group alone
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    from oracle import losses as OL
    from oracle import model as OMod
    from oracle.weights import make_inputs, make_state_dict
    from test_host_logic import _cfg
    from torchdet3d.builders import build_loss, build_model, build_optimizer
    from torchdet3d.losses import LossManager
    from torchdet3d.trainer import Trainer
    name, B, HW, nc = 'mobilenetv2', 8, 96, 9
    cfg = _cfg(name)
    cfg.optim.lr = 1e-3
    sd = make_state_dict(name, nc, seed=3 + rank)            # ranks start DIFFERENT: the broadcast must fix that
    model = build_model(cfg)
    model.load_state_dict(sd)
    model.to('cuda')
    assert model.grad_sync is not None and model.grad_sync.world == 2 and model.grad_sync.staged
    sd0 = make_state_dict(name, nc, seed=3)                  # what rank 0 holds = what everyone must hold now
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), sd0[k]), f'rank {rank}: {k} was not taken from rank 0'
    opt = build_optimizer(cfg, model)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    tr = Trainer(model, None, opt, None, lm, None, 1, '', device='cuda', save_chkpt=False)
    model.train()
    mask = torch.full((B, 1280), 2.0)
    shards = [make_inputs(B, HW, HW, nc, seed=10 + r) for r in range(world)]
    imgs, gt, cats = shards[rank]
    # one step through the API, with the dropout mask pinned (the oracle needs the same one)
    kp, tg = model(imgs.cuda(), cats.cuda(), dropout_mask=mask.cuda())
    loss = lm.parse_losses(kp, gt.cuda(), tg, cats.cuda(), 0)
    opt.zero_grad()
    loss.backward()
    g = model.net.gflat.clone()                              # averaged over the two ranks by GradSync.finish()
    opt.step()
    torch.cuda.synchronize()
    # ---- the oracle, per shard, on the host
    gref = None
    for r in range(world):
        params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in sd0.items()}
        im, gk, ct = shards[r]
        kpo, tgo = OMod.forward(params, name, im, ct, train=True, num_classes=nc, dropout_mask=mask)
        lo = OL.LossManager(OL.build(['l1', 'add_loss', 'cross_entropy']), ([1., .1], [.2])).parse_losses(kpo, gk, tgo, ct, 0)
        lo.backward()
        gr = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items() if p.requires_grad}
        gref = gr if gref is None else {k: gref[k] + gr[k] for k in gr}
    # This network at initialisation amplifies 1e-7 differences (summation order of the BatchNorm sums) by activation-kink
    # flips, and with 8 crops @96^2 per rank a single flip moves one tensor's gradient by several per cent of its maximum
    # (tests/test_gpu_golden.py holds fixtures this small to 5e-2 per tensor for the same reason, and sits at that edge):
    # the exchange is judged on the WHOLE gradient (relative L2), each tensor only against a gross-error bound.
    worst, num, den = 0.0, 0.0, 0.0
    for k, v in gref.items():
        if k not in model.net.g:
            continue
        ref = v / world
        got = model.net.g[k].cpu()       # views gflat, which the optimizer step does not modify
        num += (got - ref).pow(2).sum().item()
        den += ref.pow(2).sum().item()
        err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-3)
        worst = max(worst, err)
        assert err < 0.15, f'rank {rank}: exchanged gradient of {k} differs from the shard-averaged oracle: {err}'
    rel = (num / den) ** 0.5
    assert rel < 2e-2, f'rank {rank}: exchanged gradient differs from the shard-averaged oracle: relative L2 {rel}'
    assert torch.equal(g, model.net.gflat)
    w = model.net.flat.cpu()
    gathered = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    assert torch.equal(gathered[0], gathered[1]), 'ranks hold different weights after the step'
    if rank == 0:
        print(f'DP2_OK rel_l2={rel:.2e} worst_tensor={worst:.2e}')
    dist.destroy_process_group()


def main1():
    import test_boundary_main as T
    out = sys.argv[2]
    os.makedirs(out, exist_ok=True)
    cfgp = os.path.join(out, 'cfg_train.py')
    open(cfgp, 'w').write(T.CONFIG % (out, 'training'))
    cfg, writer, net = T._replay_main(cfgp)
    assert dist.is_initialized() and dist.get_backend() == 'nccl', 'build_model did not join the RCCL group'
    assert net.grad_sync is not None and net.grad_sync.force and net.net.grad_hook is not None
    assert 'snap_1.pth' in os.listdir(out)
    assert all(v == v for _, v, _ in writer.scalars)
    print('MAIN1_OK')
    dist.destroy_process_group()


def main2():
    dist.init_process_group('gloo')          # (two ranks on one device: RCCL refuses that; build_model leaves the group alone)
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    import test_boundary_main as T
    from torchdet3d.builders.loader_builder import SyntheticCrops
    from torchdet3d.evaluation import Evaluator, compute_metrics_per_cls
    from torchdet3d.trainer import Trainer
    seen, vals, saves = [], [], []
    ts, ev, tsave = Trainer.train_step, Evaluator.val, torch.save

    def rec_step(self, imgs, gt_kp, gt_cats, it=0):
        seen.append((int(imgs.shape[0]), float(imgs.double().sum())))
        return ts(self, imgs, gt_kp, gt_cats, it)

    def rec_val(self, *a, **k):
        r = ev(self, *a, **k)
        vals.append(r)
        return r

    def rec_save(obj, f, *a, **k):
        saves.append(str(f))
        return tsave(obj, f, *a, **k)
    Trainer.train_step, Evaluator.val, torch.save = rec_step, rec_val, rec_save
    out = sys.argv[2]
    os.makedirs(out, exist_ok=True)
    cfgp = os.path.join(out, f'cfg_train_{rank}.py')
    open(cfgp, 'w').write(T.CONFIG % (out, 'training'))
    cfg, writer, net = T._replay_main(cfgp)
    Trainer.train_step, Evaluator.val, torch.save = ts, ev, tsave
    assert net.grad_sync is not None and net.grad_sync.world == 2
    # ---- every rank trained on ITS share: 8 crops per global batch -> 4 per rank, 32 crops / 8 = 4 iterations x 2 epochs
    assert len(seen) == 8 and all(n == 4 for n, _ in seen), seen
    both = [None, None]
    dist.all_gather_object(both, [v for _, v in seen])
    assert not set(both[0]) & set(both[1]), 'the two ranks saw the same batches'
    # ---- rank 0 alone wrote the checkpoint, the log file and the scalars
    allsaves = [None, None]
    dist.all_gather_object(allsaves, saves)
    # (save_freq = 10: epoch 0 and the last epoch are written, train.py:108-110)
    assert [os.path.basename(f) for f in allsaves[0]] == ['snap_0.pth', 'snap_1.pth'] and not allsaves[1], allsaves
    nsc = [None, None]
    dist.all_gather_object(nsc, len(writer.scalars))
    assert nsc[0] > 0 and nsc[1] == 0, nsc
    dist.barrier()
    files = os.listdir(out)
    assert sum(f.startswith('train.log-') for f in files) == 1 and 'snap_1.pth' in files, files
    # ---- validation: identical reduced metrics on both ranks ...
    allv = [None, None]
    dist.all_gather_object(allv, vals)
    assert len(allv[0]) == 2 and allv[0] == allv[1], allv
    # ... equal to one process walking the WHOLE validation set (overall meters are sample means, whatever the batching)
    ds = SyntheticCrops(32, (96, 96), 9, 2)
    net.eval()
    tot = dict(ADD=0.0, SADD=0.0, ACC=0.0, IOU=0.0)
    with torch.no_grad():
        for i in range(0, 32, 8):
            im, kp, ct = ds.imgs[i:i + 8].cuda(), ds.kp[i:i + 8].cuda(), ds.cats[i:i + 8].cuda()
            pk, pc = net(im, ct)
            _, a, s, io, ac = compute_metrics_per_cls(pk, kp, pc, ct, True)
            for k, v in zip(('ADD', 'SADD', 'IOU', 'ACC'), (a, s, io, ac)):
                tot[k] += v * 8 / 32
    for k in tot:
        assert abs(tot[k] - allv[0][-1][k]) < 1e-5, (k, tot[k], allv[0][-1][k])
    # ---- and the ranks hold the same weights
    w = net.net.flat.cpu()
    gathered = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    assert torch.equal(gathered[0], gathered[1]), 'ranks hold different weights after training'
    if rank == 0:
        print('MAIN2_OK', allv[0][-1])
    dist.destroy_process_group()


if __name__ == '__main__':
    {'engine2': engine2, 'main1': main1, 'main2': main2}[sys.argv[1]]()
