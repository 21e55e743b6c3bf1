"""GPU parity of the round-2 bookkeeping kernels and API shims: pooling modes (model_builder.py:96-110), the hand-written
AdamW (optim_builder.py:10-12), the dropout-mask generator, weight re-packing after an optimizer step (eval must see the
updated weights), the one-forward-one-backward guard of the autograd bridge, and the reference's model attributes
(`extract_features`, `_glob_feature_vector`, `regressors`, `cls_fc`, `sigmoid`) composed by hand like
ModelWrapper.forward does."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_host_logic import _cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['avg', 'max', 'avg+max'])
@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('B,HW,C', [(5, 49, 960), (3, 9, 1280), (2, 100, 24)])
def test_pool_modes_fwd_bwd_vs_torch(mode, dt, B, HW, C):
    from oracle.model import act_fn
    from torchdet3d import _native as N
    dtype = torch.float32 if dt == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(B + HW + C)
    y = torch.randn(B, HW, C, generator=g).to(dtype)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    dp = torch.randn(B, C, generator=g)
    u = (y.float() * scale + shift).requires_grad_(True)
    a = act_fn(u, 'hswish')                                      # [B,HW,C]
    side = int(HW ** .5)
    a4 = a.view(B, side, side, C).permute(0, 3, 1, 2) if side * side == HW else a.permute(0, 2, 1).unsqueeze(-1)
    avg, mx = F.adaptive_avg_pool2d(a4, 1).flatten(1), F.adaptive_max_pool2d(a4, 1).flatten(1)
    ref = {'avg': avg, 'max': mx, 'avg+max': avg + mx}[mode]
    (ref * dp).sum().backward()
    yd, sc, sh, dpd = y.cuda(), scale.cuda(), shift.cuda(), dp.cuda()
    pro = N.prologue(sc, sh, None, 'hswish', False)
    pooled = torch.empty(B, C, device='cuda')
    amax = torch.full((B, C), -1, device='cuda', dtype=torch.int32)
    N.call('t3d_pool_fwd', N.dtype_code(yd), N.ptr(yd), pro, N.POOL[mode], N.ptr(pooled), N.ptr(amax), B, HW, C, N.stream())
    dz = torch.empty(B, HW, C, device='cuda', dtype=dtype)
    stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    N.call('t3d_pool_bwd', N.dtype_code(yd), N.ptr(dpd), N.ptr(yd), pro, N.POOL[mode], N.ptr(amax), N.ptr(dz), N.ptr(stats),
           B, HW, C, N.stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(pooled.cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=2e-6)
    if mode != 'avg':
        am_ref = a.detach().argmax(1)                            # first maximum in scan order, like PyTorch
        assert (amax.cpu().long() == am_ref).all()
    tol = 2e-6 if dt == 'f32' else 8e-3
    got = dz.float().cpu()
    np.testing.assert_allclose(got.numpy(), u.grad.numpy(), rtol=tol, atol=tol * u.grad.abs().max().item())
    st = stats.cpu().view(2, C)
    np.testing.assert_allclose(st[0].numpy(), got.double().sum((0, 1)).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(st[1].numpy(), (got.double() * y.double()).sum((0, 1)).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('mode', ['max', 'avg+max'])
def test_engine_pooling_mode_train_step_vs_torch_autograd(mode):
    """A whole train step of the engine with a non-default pooling mode: the pooled features and the gradient that
    reaches the last conv's BatchNorm are checked against torch autograd of the same tail (pool -> heads)."""
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d.models.engine import Net
    B, HW, nc = 4, 96, 9
    imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
    net = Net('mobilenetv2', nc, 'cuda', torch.float32, pooling_mode=mode)
    net.load_state_dict(make_state_dict('mobilenetv2', nc))
    mask = torch.full((B, 1280), 2.0, device='cuda')
    kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=mask)
    sv = net.saved
    yl, bnl = sv['yl'].clone(), net.bns['conv.1']
    feat = torch.clamp(yl * bnl.scale + bnl.shift, 0, 6).view(B, -1, 1280).cpu().requires_grad_(True)     # relu6(BN(y))
    pooled = {'max': feat.amax(1), 'avg+max': feat.mean(1) + feat.amax(1)}[mode]
    np.testing.assert_allclose(sv['pooled'].cpu().numpy(), pooled.detach().numpy(), rtol=1e-6, atol=1e-6)
    wr = torch.stack([net.p[f'regressors.{int(c)}.0.weight'].cpu() for c in cats])
    br = torch.stack([net.p[f'regressors.{int(c)}.0.bias'].cpu() for c in cats])
    kp_ref = torch.sigmoid(torch.einsum('bnf,bf->bn', wr, pooled) + br)
    lg_ref = (pooled * 2.0) @ net.p['cls_fc.1.weight'].cpu().t() + net.p['cls_fc.1.bias'].cpu()
    np.testing.assert_allclose(kp.view(B, 18).cpu().numpy(), kp_ref.detach().numpy(), atol=2e-6)
    dkp, dlg = torch.randn(B, 18), torch.randn(B, nc)
    ((kp_ref * dkp).sum() + (lg_ref * dlg).sum()).backward()
    net.backward(dkp.cuda(), dlg.cuda())
    torch.cuda.synchronize()
    # sum over pixels of the gradient at the activated feature map = BatchNorm dbeta of an identity-derivative region;
    # compare the full per-channel sums the pool backward emitted with autograd's
    inside = ((yl * bnl.scale + bnl.shift > 0) & (yl * bnl.scale + bnl.shift < 6)).view(B, -1, 1280).cpu()
    dbeta_ref = (feat.grad * inside).sum((0, 1))
    np.testing.assert_allclose(net.g['conv.1.bias'].cpu().numpy(), dbeta_ref.numpy(), rtol=1e-4, atol=1e-5)


def test_adamw_kernel_matches_torch_adamw():
    from torchdet3d.builders.optim_builder import FusedAdamW
    g = torch.Generator().manual_seed(0)
    n = 40004
    p0 = torch.randn(n, generator=g)
    pa = torch.nn.Parameter(p0.clone().cuda())
    pb = torch.nn.Parameter(p0.clone().cuda())
    oa = FusedAdamW([pa], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    ob = torch.optim.AdamW([pb], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    sched_a = torch.optim.lr_scheduler.StepLR(oa, 3, 0.5)
    sched_b = torch.optim.lr_scheduler.StepLR(ob, 3, 0.5)
    for it in range(12):
        gr = (torch.randn(n, generator=g) * (1 + it)).cuda()
        pa.grad, pb.grad = gr.clone(), gr.clone()
        v0 = pa._version
        oa.step()
        ob.step()
        assert pa._version > v0                                    # version-tracking users (engine._pack) see the update
        sched_a.step()
        sched_b.step()
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    sa, sb = oa.state_dict()['state'][0], ob.state_dict()['state'][0]
    # moments after 12 steps of gradients up to ~40 in magnitude: fp32 rounding of the lerp / fma forms (1e-6 relative
    # to the largest entries)
    for key in ('exp_avg', 'exp_avg_sq'):
        a, b = sa[key].cpu().numpy(), sb[key].cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=2e-7 * np.abs(b).max())
    assert int(sa['step']) == int(sb['step']) == 12
    # a torch.optim.AdamW checkpoint resumes into the kernel optimizer (save_snap / resume_from round trip)
    oc = FusedAdamW([torch.nn.Parameter(pb.detach().clone())], lr=1e-3)
    oc.load_state_dict(ob.state_dict())
    oc.param_groups[0]['params'][0].grad = torch.ones(n, device='cuda')
    oc.step()
    assert int(oc.state_dict()['state'][0]['step']) == 13


def test_dropout_mask_kernel_statistics_and_reproducibility():
    from torchdet3d import _native as N
    n = 256 * 1280
    a, b, c = (torch.empty(n, device='cuda') for _ in range(3))
    N.call('t3d_dropout_mask', N.ptr(a), n, 1234, 1, 0.5, N.stream())
    N.call('t3d_dropout_mask', N.ptr(b), n, 1234, 1, 0.5, N.stream())
    N.call('t3d_dropout_mask', N.ptr(c), n, 1234, 2, 0.5, N.stream())
    torch.cuda.synchronize()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert set(a.unique().tolist()) == {0.0, 2.0}
    keep = (a > 0).float().mean().item()
    assert abs(keep - 0.5) < 4 * 0.5 / n ** .5
    assert abs(((a > 0) & (c > 0)).float().mean().item() - 0.25) < 5e-3      # independent across offsets
    rows = (a.view(256, 1280) > 0).float().mean(1)
    assert rows.min() > 0.4 and rows.max() < 0.6


def test_eval_after_optimizer_step_sees_updated_weights():
    """after fwd -> bwd -> optimizer.step() an eval forward must use W_{t+1} everywhere; a fresh
    model loaded from state_dict() is the witness.  bf16 mode (packed 1x1 weight copies) and fp32 (stem copy)."""
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d.builders import build_loss, build_model, build_optimizer
    from torchdet3d.losses import LossManager
    imgs, gt_kp, cats = make_inputs(8, 96, 96, 9)
    im, gt, ca = imgs.cuda(), gt_kp.cuda(), cats.cuda()
    for sdt in ('bf16', 'f32'):
        cfg = _cfg('mobilenetv2')
        cfg.model.storage_dtype = sdt
        cfg.optim.lr = 0.05                                            # a step big enough to be visible in bf16
        m = build_model(cfg).to('cuda')
        m.load_state_dict(make_state_dict('mobilenetv2', 9))
        opt = build_optimizer(cfg, m)
        lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
        m.eval()
        with torch.no_grad():
            kp_before = m(im, ca)[0].clone()
        m.train()
        kp, tg = m(im, ca)
        loss = lm.parse_losses(kp, gt, tg, ca, 0)
        opt.zero_grad()
        loss.backward()
        opt.step()
        m.eval()
        with torch.no_grad():
            kp_after, lg_after = (t.clone() for t in m(im, ca))
        fresh = build_model(cfg).to('cuda')
        fresh.load_state_dict(m.state_dict())
        fresh.eval()
        with torch.no_grad():
            kp_f, lg_f = fresh(im, ca)
        assert torch.equal(kp_after, kp_f) and torch.equal(lg_after, lg_f), sdt
        assert (kp_after - kp_before).abs().max().item() > 1e-3, sdt


def test_backward_of_an_overwritten_forward_raises():
    from oracle.weights import make_inputs
    from torchdet3d.builders import build_model
    m = build_model(_cfg('mobilenetv2')).to('cuda')
    m.train()
    imgs, _, cats = make_inputs(4, 64, 64, 9)
    kp1, _ = m(imgs.cuda(), cats.cuda())
    kp2, _ = m(imgs.cuda().flip(0), cats.cuda())
    with pytest.raises(RuntimeError, match='overwritten'):
        kp1.sum().backward()
    kp2.sum().backward()                                                # the latest forward is fine
    with pytest.raises(RuntimeError):
        kp2.sum().backward()                                            # ... once


def test_reference_attributes_compose_like_model_wrapper_forward(golden_dir):
    """features = extract_features(x); pooled = _glob_feature_vector(features, 'avg'); kp = sigmoid(cat(regressors[c](f)))
    (model_builder.py:126-139) on MobileNetV2 (no `classifier` in between) equals model(x, cats)."""
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d.builders import build_model
    m = build_model(_cfg('mobilenetv2')).to('cuda')
    m.load_state_dict(make_state_dict('mobilenetv2', 9))
    m.eval()
    imgs, _, cats = make_inputs(6, 96, 96, 9)
    x, c = imgs.cuda(), cats.cuda()
    with torch.no_grad():
        kp, lg = m(x, c)
        feats = m.extract_features(x)
        assert feats.shape == (6, 1280, 3, 3) and feats.dtype == torch.float32
        pooled = m._glob_feature_vector(feats, mode='avg')
        assert m._glob_feature_vector(feats, 'max', reduce_dims=False).shape == (6, 1280, 1, 1)
        np.testing.assert_allclose(m._glob_feature_vector(feats, 'avg+max').cpu().numpy(),
                                   (feats.mean((2, 3)) + feats.amax((2, 3))).cpu().numpy(), rtol=1e-6, atol=1e-6)
        kp2 = torch.cat([m.regressors[int(ci)](s) for ci, s in zip(c, pooled)])
        kp2 = m.sigmoid(kp2).view(6, 9, 2)
        lg2 = m.cls_fc[1](pooled)
    np.testing.assert_allclose(kp2.cpu().numpy(), kp.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(lg2.cpu().numpy(), lg.cpu().numpy(), atol=1e-5)


def test_eval_storage_dtype_runs_validation_in_fp32_over_the_bf16_trained_parameters():
    """model.storage_dtype = 'bf16' + model.eval_storage_dtype = 'f32': train-mode forwards / backwards use the bf16
    engine, eval-mode forwards a second fp32 engine over the SAME master weights and BatchNorm buffers."""
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d.builders import build_loss, build_model, build_optimizer
    from torchdet3d.losses import LossManager
    imgs, gt_kp, cats = make_inputs(8, 96, 96, 9)
    im, gt, ca = imgs.cuda(), gt_kp.cuda(), cats.cuda()
    cfg = _cfg('mobilenetv2')
    cfg.model.storage_dtype, cfg.model.eval_storage_dtype = 'bf16', 'f32'
    m = build_model(cfg).to('cuda')
    m.load_state_dict(make_state_dict('mobilenetv2', 9))
    assert m.net_eval is not m.net and m.net_eval.flat.data_ptr() == m.net.flat.data_ptr()
    ref_cfg = _cfg('mobilenetv2')
    ref = build_model(ref_cfg).to('cuda')                       # plain fp32 model
    ref.load_state_dict(m.state_dict())
    m.eval(), ref.eval()
    with torch.no_grad():
        kp, lg = m(im, ca)
        kpr, lgr = ref(im, ca)
    assert torch.equal(kp, kpr) and torch.equal(lg, lgr)         # same engine, same weights, same kernels
    # one training step in bf16, then validation again: the fp32 eval engine must see the updated weights and buffers
    opt = build_optimizer(cfg, m)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    m.train()
    kpt, tg = m(im, ca)
    loss = lm.parse_losses(kpt, gt, tg, ca, 0)
    opt.zero_grad()
    loss.backward()
    opt.step()
    m.eval()
    ref.load_state_dict(m.state_dict())
    with torch.no_grad():
        kp2, _ = m(im, ca)
        kpr2, _ = ref(im, ca)
    assert torch.equal(kp2, kpr2) and not torch.equal(kp2, kp)


def test_launch_events_time_the_depthwise_kernel_itself():
    """t3d_set_launch_events (include/t3d.h; bench.py's roofline block): the pair rides on the kernel's own dispatch, so it
    reads the kernel's begin-to-end time -- positive, and no longer than a pair recorded around the same launch call."""
    from torchdet3d import _native as N
    B, H, W, C = 64, 56, 56, 144
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(B * H * W, C, device='cuda', generator=g).bfloat16()
    w = torch.randn(C, 9, device='cuda', generator=g) * 0.3
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.2
    pro = N.prologue(sc, sh, None, 'relu6', False)
    y = torch.empty_like(x)
    stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)

    def launch():
        N.call('t3d_dwconv_fwd', N.BF16, N.ptr(x), pro, N.ptr(w), N.ptr(y), N.ptr(stats), None, B, H, W, C, 3, 1, N.stream())
    for _ in range(3):
        launch()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        e.record()                                   # (torch creates the HIP event at its first record)
    torch.cuda.synchronize()
    inner, outer = [], []
    for _ in range(10):
        ev[0].record()
        N.call('t3d_set_launch_events', ev[2].cuda_event, ev[3].cuda_event)
        launch()
        ev[1].record()
        torch.cuda.synchronize()
        inner.append(ev[2].elapsed_time(ev[3]))
        outer.append(ev[0].elapsed_time(ev[1]))
    ref = y.clone()
    launch()                                          # the pair was consumed: a plain launch again, same result
    torch.cuda.synchronize()
    assert torch.equal(ref, y)
    mi, mo = sorted(inner)[len(inner) // 2], sorted(outer)[len(outer) // 2]
    assert 0.005 < mi <= mo * 1.02, (mi, mo)          # ms


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,nrep', [(256, 960, 4), (8, 72, 1), (100, 120, 16), (64, 672, 8)])
def test_se_bwd_affine_serves_a_fold_request(B, C, nrep):
    """t3d_se_bwd_affine with a pending BatchNorm-backward finalize request for `alpha` (include/t3d.h: t3d_fold_request) derives
    AND publishes alpha / beta / gamma / dgamma / dbeta itself: everything equals the standalone t3d_bn_bwd_finalize followed by
    the plain launch, bit for bit (same arithmetic in the same order, common.h)."""
    from torchdet3d import _native as N
    g = torch.Generator(device='cuda').manual_seed(B + C)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
    s, gg = torch.rand(B, C, device='cuda', generator=g), rnd(B, C) * 0.1
    gamma, mean, invstd = torch.rand(C, device='cuda', generator=g) + 0.5, rnd(C) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5
    count = float(B * 49)
    stride = 2 * C + 6                       # (replica stride larger than the row: as in the engine's shared statistics buffer)
    bst = torch.zeros(nrep, stride, device='cuda', dtype=torch.float64)
    bst[:, :2 * C] = torch.randn(nrep, 2 * C, device='cuda', generator=g, dtype=torch.float64) * 3
    res = {}
    for fold in (False, True):
        al, be, ga, dg, db = (torch.full((C,), float('nan'), device='cuda') for _ in range(5))
        aps, gps = torch.full((B, C), float('nan'), device='cuda'), torch.full((B, C), float('nan'), device='cuda')
        N.call('t3d_set_reduction_replicas', nrep, stride)
        try:
            if fold:
                f = N.BnFold()
                f.kind, f.C, f.count, f.nrep, f.rstride = 2, C, count, nrep, stride
                f.gamma, f.stats, f.mean, f.invstd = N.ptr(gamma), N.ptr(bst), N.ptr(mean), N.ptr(invstd)
                f.o0, f.o1, f.o2, f.o3, f.o4 = N.ptr(al), N.ptr(be), N.ptr(ga), N.ptr(dg), N.ptr(db)
                desc = torch.frombuffer(bytearray(bytes(f)), dtype=torch.uint8).cuda()
                N.call('t3d_fold_request', N.ptr(desc), N.ptr(al))
            else:
                N.call('t3d_bn_bwd_finalize', N.ptr(bst), C, count, N.ptr(gamma), N.ptr(mean), N.ptr(invstd), N.ptr(al), N.ptr(be),
                       N.ptr(ga), N.ptr(dg), N.ptr(db), N.stream())
            N.call('t3d_se_bwd_affine', N.ptr(s), N.ptr(gg), N.ptr(al), N.ptr(ga), N.ptr(aps), N.ptr(gps), B, C, N.stream())
            assert not N.lib().t3d_fold_pending()
        finally:
            N.call('t3d_set_reduction_replicas', 1, 0)
        torch.cuda.synchronize()
        res[fold] = (al, be, ga, dg, db, aps, gps)
    for name, a, b in zip(('alpha', 'beta', 'gamma', 'dgamma', 'dbeta', 'aps', 'gps'), res[False], res[True]):
        assert torch.isfinite(a).all(), name
        assert torch.equal(a, b), name
    # and against the definition
    al, _, ga = res[True][:3]
    assert torch.equal(res[True][5], s * al) and torch.allclose(res[True][6], ga + gg * al, rtol=0, atol=1e-7)
