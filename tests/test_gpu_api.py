"""GPU tests through the reference-shaped Python API (build_model / build_loss / LossManager / Trainer /
Evaluator / metrics), mirroring the reference's tests/test_pipeline.py assertions and checking the autograd
bridge against the oracle."""
import os

import numpy as np
import pytest
import torch

from test_host_logic import _cfg

pytestmark = pytest.mark.gpu


def test_losses_like_reference_test_losses():
    """reference tests/test_pipeline.py:24-30: not NaN, backward-able."""
    from torchdet3d.losses import ADD_loss, DiagLoss, WingLoss
    g = torch.Generator().manual_seed(0)
    p = torch.sigmoid(torch.randn(512, 9, 2, generator=g)).cuda().requires_grad_(True)
    t = torch.sigmoid(torch.randn(512, 9, 2, generator=g)).cuda()
    for crit in (WingLoss(), ADD_loss(), DiagLoss()):
        v = crit(p, t)
        assert not torch.any(torch.isnan(v))
        v.backward()
        assert p.grad is not None and not torch.isnan(p.grad).any()


def test_random_inference_like_reference():
    """reference tests/test_pipeline.py:50-55 (shapes of the two outputs)."""
    from torchdet3d.builders import build_model
    m = build_model(_cfg('mobilenetv3_large')).to('cuda')
    m.eval()
    with torch.no_grad():
        kp, cat = m(torch.rand(16, 3, 224, 224).cuda(), torch.randint(0, 9, (16,)).cuda())
    assert kp.shape == (16, 9, 2) and cat.shape == (16, 9)
    assert (kp > 0).all() and (kp < 1).all()
    m1 = build_model(_cfg('mobilenetv3_large', nc=1)).to('cuda')
    cats = torch.randint(0, 1, (4,)).cuda()
    kp, tg = m1(torch.rand(4, 3, 96, 96).cuda(), cats)
    assert tg.shape == (4, 1) and tg.dtype == torch.int64          # model_builder.py:144
    exp = build_model(_cfg('mobilenetv3_large'), export_mode=True).to('cuda')
    okp, olg = exp(torch.rand(2, 3, 96, 96).cuda())
    assert okp.shape == (9, 2, 9, 2) and olg.shape == (2, 9)        # model_builder.py:112-124


def test_metrics_like_reference_test_metrics(golden_dir):
    """reference tests/test_pipeline.py:18-22 (ranges, structure) + values against the golden metrics."""
    from torchdet3d.evaluation import compute_accuracy, compute_average_distance, compute_metrics_per_cls
    g = torch.Generator().manual_seed(1)
    pk, gk = torch.rand(128, 9, 2, generator=g).cuda(), torch.rand(128, 9, 2, generator=g).cuda()
    pc, gc = torch.rand(128, 9, generator=g).cuda(), torch.randint(0, 9, (128,), generator=g).cuda()
    per, ADD, SADD, IOU, acc = compute_metrics_per_cls(pk, gk, pc, gc, compute_iou=True)
    for v in (ADD, SADD, IOU, acc):
        assert 0 <= v <= 1
    assert len(per) == 9 and all(len(r) == 5 for r in per)
    gold = np.load(os.path.join(golden_dir, 'metrics.npz'))
    p, t = torch.from_numpy(gold['p']).cuda(), torch.from_numpy(gold['t']).cuda()
    np.testing.assert_allclose(compute_average_distance(p, t), gold['add_mean'], rtol=3e-6)
    np.testing.assert_allclose(compute_average_distance(p, t, reduce_mean=False), gold['add_sum'], rtol=3e-6)
    lg, c = torch.from_numpy(gold['logits']).cuda(), torch.from_numpy(gold['cats']).cuda()
    assert compute_accuracy(lg, c) == pytest.approx(float(gold['acc_mean'])) and compute_accuracy(lg, c, False) == float(gold['acc_sum'])
    # per-class aggregation against the oracle's restatement of metrics.py:39-68
    from oracle import metrics as OM
    ref = OM.metrics_per_cls(pk.cpu(), gk.cpu(), pc.cpu(), gc.cpu(), compute_iou=True)
    np.testing.assert_allclose([ADD, SADD, IOU, acc], ref[1:], rtol=1e-5, atol=1e-6)
    for a, b in zip(per, ref[0]):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


def test_autograd_bridge_matches_oracle_and_optimizer_moves_weights():
    """model(imgs, cats) -> LossManager.parse_losses -> loss.backward() -> optimizer.step(), exactly the
    reference loop body (trainer/train.py:46-52), against the oracle's gradients."""
    from oracle import losses as OL
    from oracle import model as OMod
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d.builders import build_loss, build_model, build_optimizer
    from torchdet3d.losses import LossManager
    name, B, HW, nc = 'mobilenetv3_large', 4, 96, 9
    cfg = _cfg(name)
    sd = make_state_dict(name, nc)
    imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in sd.items()}
    ones = torch.ones(B, 1280)
    kp_o, tg_o = OMod.forward(params, name, imgs, cats, train=True, num_classes=nc, dropout_mask=ones)
    loss_o = OL.LossManager(OL.build(cfg.loss.names), cfg.loss.coeffs).parse_losses(kp_o, gt_kp, tg_o, cats, 0)
    loss_o.backward()

    m = build_model(cfg).to('cuda')
    m.load_state_dict(sd)
    m.train()
    opt = build_optimizer(cfg, m)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    kp, tg = m(imgs.cuda(), cats.cuda(), dropout_mask=ones.cuda())
    loss = lm.parse_losses(kp, gt_kp.cuda(), tg, cats.cuda(), 0)
    assert abs(loss.item() - loss_o.item()) < 1e-5
    opt.zero_grad()
    loss.backward()
    for k in ('conv.0.weight', 'classifier.0.weight', 'features.5.conv.5.fc.0.weight', 'features.0.0.weight'):
        ref = params[k].grad
        err = (m.net.g[k].cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-3)
        assert err < 2e-2, (k, err)
    assert m.flat.grad is not None and torch.equal(m.flat.grad, m.net.gflat)
    before = m.net.p['conv.0.weight'].clone()
    opt.step()
    assert (m.net.p['conv.0.weight'] - before).abs().max() > 0


def test_trainer_and_evaluator_on_synthetic_loader(tmp_path):
    from torchdet3d.builders import build_loader, build_loss, build_model, build_optimizer, build_scheduler
    from torchdet3d.evaluation import Evaluator
    from torchdet3d.losses import LossManager
    from torchdet3d.trainer import Trainer
    cfg = _cfg('mobilenetv2')
    cfg.data.update(root='synthetic', resize=(96, 96), train_batch_size=16, val_batch_size=16, synthetic_len=64)
    train_loader, val_loader, _ = build_loader(cfg)
    net = build_model(cfg).to('cuda')
    opt = build_optimizer(cfg, net)
    sched = build_scheduler(cfg, opt)

    class W:                      # SummaryWriter stand-in (tensorboard is not in the image)
        def __init__(self):
            self.rows = []

        def add_scalar(self, tag, v, global_step=None):
            self.rows.append((tag, float(v), global_step))

    w = W()
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    tr = Trainer(model=net, train_loader=train_loader, optimizer=opt, scheduler=sched, loss_manager=lm, writer=w,
                 max_epoch=3, log_path=str(tmp_path), device='cuda', save_chkpt=True, debug=False, save_freq=10,
                 print_freq=100, train_step=0)
    first = tr.train(0, False)['loss']
    tr.train(1, False)
    last = tr.train(2, True)['loss']
    assert last < first, (first, last)                         # it learns the synthetic set
    assert tr.global_step == 12 and os.path.exists(tmp_path / 'snap_2.pth') and os.path.exists(tmp_path / 'snap_0.pth')
    assert {r[0] for r in w.rows} == {'Train/loss', 'Train/ADD', 'Train/SADD', 'Train/ACC'}
    r = tr.train_step(*next(iter(train_loader)), 0)
    assert set(r) == {'loss', 'ADD', 'SADD', 'acc'} and all(np.isfinite(v) for v in r.values())
    # the lazily resolved mapping through the C-level consumers too (as a dict subclass these saw an empty dict)
    import copy
    import json
    import pickle
    for make in (lambda: dict(tr.train_step(*next(iter(train_loader)), 0)), lambda: {**tr.train_step(*next(iter(train_loader)), 0)},
                 lambda: json.loads(json.dumps(dict(tr.train_step(*next(iter(train_loader)), 0)))),
                 lambda: copy.copy(tr.train_step(*next(iter(train_loader)), 0)),
                 lambda: pickle.loads(pickle.dumps(tr.train_step(*next(iter(train_loader)), 0)))):
        d = make()
        assert set(d) == {'loss', 'ADD', 'SADD', 'acc'} and all(np.isfinite(v) for v in d.values()), d
    ev = Evaluator(model=net, val_loader=val_loader, cfg=cfg, writer=w, max_epoch=3, device='cuda')
    res = ev.val(epoch=2, compute_iou=True)
    assert 0 <= res['ADD'] <= 2 and 0 <= res['IOU'] <= 1 and 0 <= res['ACC'] <= 1
    assert any(r[0] == 'Val/IOU' for r in w.rows)
    per, *_ = ev.val_step(*next(iter(val_loader)))
    assert all(len(row) == 5 for row in per)
    # `val` keeps one batch in flight (val_enqueue / PendingMetrics.result, metric kernels on a second stream): the same numbers
    # as the batch-by-batch `val_step` sums, to the last bit
    tot = np.zeros(4)
    cnt = 0
    for imgs, gt_kp, gt_cats in val_loader:
        _, ADD, SADD, IOU, ACC = ev.val_step(imgs, gt_kp, gt_cats)
        tot += np.array([ADD, SADD, ACC, IOU]) * imgs.size(0)
        cnt += imgs.size(0)
    res2 = ev.val(compute_iou=True)
    assert [res2[k] for k in ('ADD', 'SADD', 'ACC', 'IOU')] == list(tot / cnt)
    p0 = ev.val_enqueue(*next(iter(val_loader)))
    p1 = ev.val_enqueue(*next(iter(val_loader)))                       # two batches in flight, collected out of order
    assert p1.result() == p0.result() == ev.val_step(*next(iter(val_loader)))


def test_train_mode_under_no_grad_uses_and_updates_batch_statistics_like_the_reference():
    """`model.train()` + `torch.no_grad()`: nn.BatchNorm2d still normalises with the batch statistics and moves the running
    estimates, nn.Dropout still drops (only autograd is off).  Against the oracle's train-mode forward on the same weights."""
    from oracle import model as OMod
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d.builders import build_model
    name, B, HW, nc = 'mobilenetv3_large', 8, 96, 9
    sd = make_state_dict(name, nc)
    imgs, _, cats = make_inputs(B, HW, HW, nc)
    m = build_model(_cfg(name))
    m.load_state_dict(sd)
    m.to('cuda')
    m.train()
    mask = torch.full((B, 1280), 2.0)
    with torch.no_grad():
        kp, lg = m(imgs.cuda(), cats.cuda(), dropout_mask=mask.cuda())
    ref = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        kp_o, lg_o = OMod.forward(ref, name, imgs, cats, train=True, num_classes=nc, dropout_mask=mask)
    np.testing.assert_allclose(kp.cpu().numpy(), kp_o.numpy(), atol=1e-4)
    np.testing.assert_allclose(lg.cpu().numpy(), lg_o.numpy(), atol=2e-4)
    got = m.state_dict()
    for k in ('features.3.conv.4.running_mean', 'features.3.conv.4.running_var', 'conv.1.running_mean'):
        assert not torch.equal(ref[k], sd[k])                           # the oracle moved them ...
        np.testing.assert_allclose(got[k].cpu().numpy(), ref[k].numpy(), rtol=1e-4, atol=1e-5)   # ... and so did the engine
    assert int(got['features.0.1.num_batches_tracked']) == int(sd['features.0.1.num_batches_tracked']) + 1
    assert not kp.requires_grad
    m.eval()
    with torch.no_grad():
        kp_e, _ = m(imgs.cuda(), cats.cuda())
    assert (kp_e - kp).abs().max() > 1e-4            # eval mode (running statistics) is a different function
