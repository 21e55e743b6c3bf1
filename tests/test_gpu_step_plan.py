"""The train iteration as ONE host call (csrc/plan.hip, torchdet3d/trainer/step_plan.py) against the reference-shaped eager
sequence of torchdet3d/trainer/train.py:44-55 (`model(...)`, `parse_losses`, `loss.backward()`, `optimizer.step()`):
the three forms -- eager through autograd, direct entry-point sequence, recorded plan replayed by `t3d_plan_run` -- must leave
bit-identical weights, BatchNorm buffers, optimizer state and metrics."""
import ctypes

import pytest
import torch

from test_host_logic import _cfg

pytestmark = pytest.mark.gpu


def _objects(name, dtype, nc=9, seed=3):
    from torchdet3d.builders import build_loss, build_model, build_optimizer
    from torchdet3d.losses import LossManager
    from torchdet3d.trainer import Trainer
    cfg = _cfg(name, nc=nc)
    cfg.model.storage_dtype = dtype
    if nc == 1:
        cfg.loss.names, cfg.loss.coeffs = ['l1', 'add_loss'], ([1., .1], [])
    torch.manual_seed(seed)
    model = build_model(cfg).to('cuda')
    model.net.reset_parameters(seed=seed)
    opt = build_optimizer(cfg, model)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    tr = Trainer(model, None, opt, None, lm, None, 1, '', device='cuda', save_chkpt=False)
    model.train()
    return model, opt, lm, tr


def _batches(B, S, nb=3, u8=False, seed=11, nc=9):
    g = torch.Generator(device='cuda').manual_seed(seed)
    if u8:
        imgs = [torch.randint(0, 256, (B, S, S, 3), device='cuda', generator=g, dtype=torch.uint8) for _ in range(nb)]
    else:
        imgs = [torch.randn(B, 3, S, S, device='cuda', generator=g) for _ in range(nb)]
    gts = [torch.rand(B, 9, 2, device='cuda', generator=g) for _ in range(nb)]
    cats = [torch.randint(0, nc, (B,), device='cuda', generator=g) for _ in range(nb)]
    return imgs, gts, cats


def _run(name, dtype, B, S, steps, mode, nc=9, u8=False, lr_at=None):
    """mode: 'eager' | 'direct' | 'replay'.  Returns (weights, buffers, optimizer moments, per-step metric dicts, Trainer)."""
    from torchdet3d.trainer import step_plan
    model, opt, lm, tr = _objects(name, dtype, nc)
    imgs, gts, cats = _batches(B, S, u8=u8, nc=nc)
    if mode == 'eager':
        tr._sp = None
    old = step_plan.REPLAY
    step_plan.REPLAY = mode == 'replay'
    try:
        res = []
        for i in range(steps):
            if lr_at is not None and i == lr_at:
                opt.param_groups[0]['lr'] = 3e-4           # what an LR scheduler does between two iterations
            j = i % len(imgs)
            res.append(dict(tr.train_step(imgs[j], gts[j], cats[j], i)))
    finally:
        step_plan.REPLAY = old
    torch.cuda.synchronize()
    st = opt.state[model.flat]
    return (model.net.flat.clone(), {k: v.clone() for k, v in model.net.buffers.items()},
            (st['step'], st['exp_avg'].clone(), st['exp_avg_sq'].clone()), res, tr)


def _same(a, b):
    wa, ba, oa, ra, _ = a
    wb, bb, ob, rb, _ = b
    assert torch.equal(wa, wb), f'weights differ: max {(wa - wb).abs().max().item():.3e}'
    for k in ba:
        assert torch.equal(ba[k], bb[k]), k
    assert oa[0] == ob[0] and torch.equal(oa[1], ob[1]) and torch.equal(oa[2], ob[2])
    assert ra == rb, (ra[-1], rb[-1])


@pytest.mark.parametrize('name,dtype,B,S', [('mobilenetv2', 'bf16', 16, 96), ('mobilenetv2', 'bf16', 32, 224),
                                            ('mobilenetv2', 'f32', 8, 96), ('mobilenetv3_large', 'bf16', 16, 96),
                                            ('mobilenetv3_small', 'bf16', 8, 96), ('resnet50', 'bf16', 8, 96)])
def test_three_forms_of_the_step_are_bit_identical(name, dtype, B, S):
    steps = 7
    eager = _run(name, dtype, B, S, steps, 'eager', lr_at=5)
    direct = _run(name, dtype, B, S, steps, 'direct', lr_at=5)
    replay = _run(name, dtype, B, S, steps, 'replay', lr_at=5)
    _same(eager, direct)
    _same(eager, replay)
    sp = replay[4]._sp
    assert sp is not None and sp.rec is not None and sp.replays == steps - 3      # two warm steps, one recorded, the rest replayed
    assert direct[4]._sp.replays == 0 and eager[4]._sp is None
    n_calls = len(sp.rec.calls)
    lib = __import__('torchdet3d._native', fromlist=['x']).lib()
    assert lib.t3d_plan_num_ops(sp.rec.plan, 0) == n_calls and n_calls > 50
    assert lib.t3d_plan_num_ops(sp.rec.plan, 2) == 1 and lib.t3d_plan_num_ops(sp.rec.plan, 3) == 1     # the metrics read-back


def test_uint8_crops_and_a_single_class_model_replay():
    a = _run('mobilenetv2', 'bf16', 8, 96, 6, 'eager', nc=1, u8=True)
    b = _run('mobilenetv2', 'bf16', 8, 96, 6, 'replay', nc=1, u8=True)
    _same(a, b)
    assert b[4]._sp.replays == 3


def test_plan_is_rerecorded_when_the_step_changes_and_eval_in_between_is_harmless():
    from torchdet3d.trainer import step_plan
    assert step_plan.REPLAY
    model, opt, lm, tr = _objects('mobilenetv2', 'bf16')
    model2, opt2, lm2, tr2 = _objects('mobilenetv2', 'bf16')
    tr2._sp = None                                                        # the eager twin
    imgs, gts, cats = _batches(8, 96)
    imgs_b, gts_b, cats_b = _batches(4, 96, seed=5)
    seq = [(imgs, gts, cats)] * 5 + [(imgs_b, gts_b, cats_b)] * 5 + [(imgs, gts, cats)] * 4
    ev = []
    for i, (I, G, C) in enumerate(seq):
        j = i % 3
        r1, r2 = dict(tr.train_step(I[j], G[j], C[j], i)), dict(tr2.train_step(I[j], G[j], C[j], i))
        assert r1 == r2, (i, r1, r2)
        if i in (3, 8):                      # a validation forward between two replayed steps (another engine, same parameters)
            for m in (model, model2):
                m.eval()
                with torch.no_grad():
                    ev.append(m(I[0], C[0])[0].clone())
                m.train()
    assert torch.equal(model.net.flat, model2.net.flat)
    assert torch.equal(ev[0], ev[1]) and torch.equal(ev[2], ev[3])
    assert tr._sp.replays == 2 + 2 + 1        # each change of the batch shape costs two warm steps and a recording


def test_what_the_plan_does_not_cover_takes_the_eager_form():
    from torchdet3d.trainer.step_plan import StepPlan
    model, opt, lm, tr = _objects('mobilenetv2', 'bf16')
    assert StepPlan.usable(model, lm, opt)
    lm.use_alwa = True
    assert not StepPlan.usable(model, lm, opt)
    lm.use_alwa = False
    assert not StepPlan.usable(model, lm, torch.optim.SGD(model.parameters(), lr=0.1))
    sp = StepPlan(model, lm, opt)
    imgs, gts, cats = _batches(4, 96, nb=1)
    assert sp.accepts(imgs[0], gts[0], cats[0])
    assert not sp.accepts(imgs[0].cpu(), gts[0], cats[0])
    assert not sp.accepts(imgs[0], gts[0], cats[0].int())
    model.eval()
    assert not sp.accepts(imgs[0], gts[0], cats[0])


def test_plan_entry_table_rejects_unknown_names_and_wrong_arity():
    from torchdet3d import _native as N
    lib = N.lib()
    plan = ctypes.c_void_p()
    assert lib.t3d_plan_create(ctypes.byref(plan)) == 0
    k, w, s = (ctypes.c_int * 4)(), (ctypes.c_ulonglong * 4)(), (ctypes.c_int * 4)()
    assert lib.t3d_plan_add_call(plan, b'no_such_entry', 0, k, w, s) == -3
    assert lib.t3d_plan_add_call(plan, b't3d_zero_batched', 2, k, w, s) == -1
    assert lib.t3d_plan_run(plan, 0, None, 0, None, 0) == -1          # no such segment
    assert lib.t3d_plan_run(plan, -1, None, 0, None, 0) == 0           # an empty plan
    lib.t3d_plan_destroy(plan)


def test_a_non_finite_activation_is_seen_in_the_batchnorm_sums():
    """the clamp form of ReLU6 (16-bit kernels) maps a NaN pre-activation to 0, so a diverged step can end in a
    finite loss; `Net.nonfinite()` reads it off the BatchNorm coefficients (the batch sums are taken on the raw conv outputs)."""
    model, opt, lm, tr = _objects('mobilenetv2', 'bf16')
    imgs, gts, cats = _batches(8, 96, nb=2)
    for i in range(4):
        r = dict(tr.train_step(imgs[i % 2], gts[i % 2], cats[i % 2], i))
    assert r['loss'] == r['loss'] and not model.net.nonfinite()
    bad = imgs[0].clone()
    bad[3, 1, 40:44, 40:44] = float('inf')
    assert opt.first_nonfinite_step() is None
    dict(tr.train_step(bad, gts[0], cats[0], 4))
    assert model.net.nonfinite()
    # ... and in the gradient the optimizer kernel reads: the FIRST diverged step is on record (t3d_set_grad_watch), replayed
    # steps included, so the trainer can report NaN from exactly that step on
    assert opt.first_nonfinite_step() == 5
    dict(tr.train_step(imgs[1], gts[1], cats[1], 5))
    assert opt.first_nonfinite_step() == 5


def test_trainer_reports_nan_from_the_diverging_step_on_and_not_before():
    """`Trainer.train` looks at its numbers every print_freq iterations; a divergence inside such a window must turn the loss of
    the diverging iteration and the later ones into NaN (what the reference's `loss.item()` shows, train.py:57) and leave the
    finite iterations before it alone."""
    from torchdet3d.trainer import Trainer
    model, opt, lm, _ = _objects('mobilenetv2', 'bf16')
    imgs, gts, cats = _batches(8, 96, nb=6)
    imgs[4] = imgs[4].clone()
    imgs[4][2, 0, 10:14, 10:14] = float('inf')

    class W:
        def __init__(self): self.rows = []
        def add_scalar(self, tag, v, global_step=None):
            if tag == 'Train/loss': self.rows.append((global_step, v))

    w = W()
    tr = Trainer(model, list(zip(imgs, gts, cats)), opt, None, lm, w, 1, '', device='cuda', save_chkpt=False, print_freq=100)
    tr.train(0, False)
    losses = [v for _, v in sorted(w.rows)]
    assert len(losses) == 6
    assert all(v == v for v in losses[:4]), losses
    assert all(v != v for v in losses[4:]), losses


@pytest.mark.parametrize('name,dtype,evdt', [('mobilenetv2', 'bf16', 'f32'), ('mobilenetv2', 'bf16', 'f16'), ('mobilenetv2', 'bf16', 'bf16'),
                                             ('mobilenetv3_large', 'f32', None), ('resnet50', 'bf16', 'f32')])
def test_eval_forward_replayed_from_a_plan_equals_the_launch_by_launch_forward(name, dtype, evdt):
    """`model.eval(); model(x, cats)` (validation / serving) records its launches once and replays them by one `t3d_plan_run`
    (trainer/step_plan.py: ForwardPlan): same outputs bit for bit, fresh output tensors, and the packed weights follow a
    training step in between."""
    from torchdet3d.builders import build_loss, build_model, build_optimizer
    from torchdet3d.losses import LossManager
    from torchdet3d.trainer import Trainer
    cfg = _cfg(name)
    cfg.model.storage_dtype = dtype
    cfg.model.eval_storage_dtype = evdt
    torch.manual_seed(1)
    m = build_model(cfg).to('cuda')
    m.net.reset_parameters(seed=1)
    opt = build_optimizer(cfg, m)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    tr = Trainer(m, None, opt, None, lm, None, 1, '', device='cuda', save_chkpt=False)
    B, S = (96, 96) if name == 'mobilenetv2' else (8, 96)          # (B >= 96: the fused 14x14 / 7x7 inference blocks are in the plan)
    imgs, gts, cats = _batches(B, S, nb=3)

    def eager(i):
        with torch.no_grad():
            kp, lg = m.net_eval.forward(imgs[i], cats[i], train=False)
        return kp.clone(), lg.clone()

    m.eval()
    outs = []
    with torch.no_grad():
        for i in range(6):
            kp, tg = m(imgs[i % 3], cats[i % 3])
            outs.append((kp, tg))
            ek, el = eager(i % 3)
            assert torch.equal(kp, ek) and torch.equal(tg, el), i
    fp = m._fplan
    assert fp.replays == 3 and outs[3][0].data_ptr() != outs[4][0].data_ptr()
    assert torch.equal(outs[0][0], outs[3][0])                       # batch 0 again, through the plan this time
    m.train()
    for i in range(4):
        dict(tr.train_step(imgs[i % 3], gts[i % 3], cats[i % 3], i))
    m.eval()
    with torch.no_grad():
        kp, tg = m(imgs[0], cats[0])
        ek, el = eager(0)
    assert torch.equal(kp, ek) and torch.equal(tg, el) and not torch.equal(kp, outs[0][0])      # new weights, same plan
    assert m._fplan.replays == 4
