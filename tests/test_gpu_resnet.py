"""GPU: the ResNet-50 regression model (BASELINE config 4; models/resnet.py, csrc/resnet.hip) against oracle/resnet.py --
the standard torchvision architecture inside the reference's ModelWrapper; the reference has no ResNet, so this parity is
unpinned w.r.t. the reference.  fp32 storage: outputs 1e-4 / arg-max exact; train step: loss, BatchNorm
running statistics, gradients (relative L2 per tensor); bf16 storage: metrics level."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(dtype, B, HW, train, name='resnet50'):
    from oracle import losses as OL
    from oracle import resnet as R
    from oracle.weights import make_inputs, make_state_dict
    from test_gpu_engine import _loss_cfg
    from torchdet3d import _native as N
    from torchdet3d.models.resnet import ResNetEngine
    nc = 9
    imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
    layers = R.TINY_LAYERS if name == 'resnet14' else None
    feat = 512 if name == 'resnet14' else 2048
    sd = make_state_dict(name, nc)
    net = ResNetEngine(name, nc, 'cuda', dtype)
    net.load_state_dict(sd)
    mask = torch.full((B, feat), 2.0)
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in sd.items()}
    kp_o, lg_o = R.forward(params, imgs, cats, train=train, num_classes=nc, dropout_mask=mask, layers=layers)
    kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=train, dropout_mask=mask.cuda() if train else None)
    return net, params, (kp, lg), (kp_o, lg_o), (imgs, gt_kp, cats), _loss_cfg, N, OL


def test_resnet50_eval_forward_matches_the_oracle():
    net, params, (kp, lg), (kp_o, lg_o), _, _, _, _ = _step(torch.float32, 4, 96, False)
    np.testing.assert_allclose(kp.cpu().numpy(), kp_o.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(lg.cpu().numpy(), lg_o.detach().numpy(), atol=1e-4)
    assert (lg.argmax(1).cpu() == lg_o.argmax(1)).all()


def test_resnet50_at_baseline_config_4_workload():
    """BASELINE config 4 at its own per-GPU workload (VERDICT r3 missing #4): ResNet-50, 224 x 224, 64 crops per GPU
    (global batch 512 on 8 GPUs).  fp32 storage: eval-mode keypoints / logits against oracle/resnet.py at 1e-4, class arg-max
    exact; bf16 storage (`eval_storage_dtype = 'bf16'` semantics: the engine's own bf16 inference): ADD / SADD / accuracy of
    the outputs against the oracle's, metric level (evaluation/metrics.py:10-37), 1e-3."""
    from oracle import metrics as OM
    net, params, (kp, lg), (kp_o, lg_o), (imgs, gt_kp, cats), _, _, _ = _step(torch.float32, 64, 224, False)
    np.testing.assert_allclose(kp.cpu().numpy(), kp_o.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(lg.cpu().numpy(), lg_o.detach().numpy(), atol=1e-4)
    assert (lg.argmax(1).cpu() == lg_o.argmax(1)).all()
    del net
    from oracle.weights import make_state_dict
    from torchdet3d.models.resnet import ResNetEngine
    nb = ResNetEngine('resnet50', 9, 'cuda', torch.bfloat16)
    nb.load_state_dict(make_state_dict('resnet50', 9))
    kpb, lgb = nb.forward(imgs.cuda(), cats.cuda(), train=False)
    add_o, sadd_o = OM.average_distance(kp_o.detach(), gt_kp)
    add_b, sadd_b = OM.average_distance(kpb.cpu(), gt_kp)
    acc_o, acc_b = OM.accuracy(lg_o.detach(), cats), OM.accuracy(lgb.cpu(), cats)
    print(f'   resnet50 64@224 bf16 eval: dADD {add_b - add_o:+.2e} dSADD {sadd_b - sadd_o:+.2e} acc {acc_b} / {acc_o}, '
          f'max keypoint deviation {(kpb.cpu() - kp_o.detach()).abs().max().item():.2e}')
    assert abs(add_b - add_o) < 1e-3 and abs(sadd_b - sadd_o) < 1e-3 and abs(acc_b - acc_o) <= 1 / 64 + 1e-9


def _grad_errors(name, B, HW, seed):
    from oracle import losses as OL
    from oracle import resnet as R
    from oracle.weights import make_inputs, make_state_dict
    from test_gpu_engine import _loss_cfg
    from torchdet3d import _native as N
    from torchdet3d.models.resnet import ResNetEngine
    nc = 9
    layers = R.TINY_LAYERS if name == 'resnet14' else None
    feat = 512 if name == 'resnet14' else 2048
    imgs, gt_kp, cats = make_inputs(B, HW, HW, nc, seed=seed)
    sd = make_state_dict(name, nc)
    net = ResNetEngine(name, nc, 'cuda', torch.float32)
    net.load_state_dict(sd)
    mask = torch.full((B, feat), 2.0)
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in sd.items()}
    kp_o, lg_o = R.forward(params, imgs, cats, train=True, num_classes=nc, dropout_mask=mask, layers=layers)
    kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=mask.cuda())
    np.testing.assert_allclose(kp.cpu().numpy(), kp_o.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(lg.cpu().numpy(), lg_o.detach().numpy(), atol=2e-4)
    lm = OL.LossManager(OL.build(['l1', 'add_loss', 'cross_entropy']), ([1., .1], [.2]))
    loss_o = lm.parse_losses(kp_o, gt_kp, lg_o, cats, 0)
    loss_o.backward()
    out = torch.zeros(16, device='cuda')
    dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
    gtd, cd = gt_kp.cuda().view(B, 18).contiguous(), cats.cuda()        # (kept alive: the call takes raw pointers)
    N.call('t3d_loss_fwd_bwd', _loss_cfg(['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2])), N.ptr(kp),
           N.ptr(gtd), N.ptr(lg), N.ptr(cd), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
    assert abs(out[0].item() - loss_o.item()) < 1e-4 * max(1.0, abs(loss_o.item()))
    net.backward(dkp, dlg)
    torch.cuda.synchronize()
    last = 'layer2.1.bn2' if name == 'resnet14' else 'layer4.2.bn2'
    for k in ('bn1', 'layer2.0.downsample.1', last):
        np.testing.assert_allclose(net.buffers[k + '.running_mean'].cpu().numpy(), params[k + '.running_mean'].numpy(), atol=2e-5)
        np.testing.assert_allclose(net.buffers[k + '.running_var'].cpu().numpy(), params[k + '.running_var'].numpy(), rtol=2e-4, atol=1e-6)
    gmax = max(p.grad.abs().max().item() for p in params.values() if p.requires_grad and p.grad is not None)
    errs = []
    for k, p in params.items():
        if not p.requires_grad or p.grad is None:
            continue
        ref, got = p.grad, net.g[k].cpu()
        errs.append((((got - ref).norm() / max(ref.norm().item(), 1e-3 * gmax * ref.numel() ** 0.5)).item(), k))
    errs.sort(reverse=True)
    return errs


def test_resnet_backward_matches_the_oracle_on_the_shallow_model():
    """resnet14 (test-only: ResNet-50's four bottleneck kinds -- projection / identity shortcut, stride 1 / 2 -- in 4 blocks).
    With ~8 M ReLU inputs per sample a handful of pre-activations differ in SIGN between the two implementations (|u| < 1e-6:
    the GEMMs sum in different orders), and each such element moves every gradient below it by a fraction of a per cent
    (found with tools/scratch/dbg_resnet.py: ONE element of d1 in the top block accounts for the 0.9 % on its BatchNorm's
    bias gradient).  That floor is a property of the comparison, not of the backward: three samples, every tensor within
    2e-2 relative L2 (2e-3 .. 9e-3 measured); the backward's arithmetic itself is pinned kernel by kernel against torch
    autograd at 1e-5 below."""
    res = [_grad_errors('resnet14', 16, 96, seed) for seed in (0, 1, 2)]
    for errs in res:
        print('   resnet14 worst:', [(round(a, 5), b) for a, b in errs[:3]], 'median', float(np.median([e[0] for e in errs])))
        assert errs[0][0] < 2e-2, errs[:6]


def test_resnet50_train_step_matches_the_oracle():
    """The full 53-layer model: forward, loss and BatchNorm running statistics to 1e-4 (inside `_grad_errors`).  At random
    initialisation it amplifies the top layers' round-off / kink flips by ~9 % per layer (the conditioning
    tests/test_gpu_bf16_gate.py documents for MobileNetV2), so its gradients are held to a gross bound only; the backward
    itself is pinned on the shallow model above and kernel by kernel below."""
    errs = _grad_errors('resnet50', 8, 96, 0)
    print('   resnet50 worst:', [(round(a, 5), b) for a, b in errs[:3]], 'median', float(np.median([e[0] for e in errs])))
    assert errs[0][0] < 8e-2, errs[:6]


def test_resnet50_through_the_api_in_bf16():
    """build_model('resnet50') -> LossManager -> backward -> optimizer in the throughput mode: finite, the loss goes down
    over a few steps on one batch, eval-mode outputs come from the fp32-storage engine."""
    from test_host_logic import _cfg
    from oracle.weights import make_inputs
    from torchdet3d.builders import build_loss, build_model, build_optimizer
    from torchdet3d.losses import LossManager
    cfg = _cfg('resnet50')
    cfg.model.storage_dtype = 'bf16'
    m = build_model(cfg).to('cuda')
    assert sum(p.numel() for p in m.parameters()) >= 23_800_000
    opt = build_optimizer(cfg, m)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    imgs, gt_kp, cats = make_inputs(16, 128, 128, 9)
    m.train()
    losses = []
    for it in range(6):
        kp, tg = m(imgs.cuda(), cats.cuda())
        loss = lm.parse_losses(kp, gt_kp.cuda(), tg, cats.cuda(), it)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    m.eval()
    with torch.no_grad():
        kp, tg = m(imgs.cuda(), cats.cuda())
    assert kp.shape == (16, 9, 2) and tg.shape == (16, 9) and torch.isfinite(kp).all()


@pytest.mark.parametrize('B,H,W,C,k,s,pad', [(2, 12, 12, 16, 3, 1, 1), (2, 13, 11, 8, 3, 2, 1), (1, 9, 9, 8, 7, 2, 3),
                                             (2, 12, 12, 12, 3, 1, 1),            # C % 8 != 0: the scalar forms
                                             (3, 14, 14, 64, 3, 1, 1), (2, 15, 15, 128, 3, 2, 1), (2, 10, 9, 512, 3, 1, 1),
                                             (5, 28, 28, 24, 3, 2, 1), (2, 7, 7, 520, 3, 1, 1)])  # 520: C / 8 > 64 -> scalar backward
def test_im2col_and_its_backward_match_torch_autograd(B, H, W, C, k, s, pad):
    """t3d_im2col (BatchNorm affine + ReLU on load, zero padding after the activation) and t3d_col2im_bwd (gather of the
    patch gradients, times relu', + the BatchNorm-backward sums) against torch autograd of the same function."""
    import torch.nn.functional as F
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(H * 31 + k)
    x = torch.randn(B, H, W, C, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    Kp = (k * k * C + 31) // 32 * 32
    xr = x.clone().requires_grad_(True)
    a = F.relu(xr * sc + sh).permute(0, 3, 1, 2)                           # NCHW
    cols = F.unfold(a, k, padding=pad, stride=s)                            # [B, C*k*k, L], row (c*k*k + t)
    cols = cols.view(B, C, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, k * k * C)     # column t*C + c
    dcol = torch.randn(B * Ho * Wo, Kp, generator=g)
    cols.backward(dcol[:, :k * k * C])
    xd, scd, shd = x.cuda(), sc.cuda(), sh.cuda()
    pro = N.prologue(scd, shd, None, 'relu', False)
    col = torch.full((B * Ho * Wo, Kp), 7.0, device='cuda')
    N.call('t3d_im2col', N.F32, N.ptr(xd), pro, N.ptr(col), B, H, W, C, k, s, pad, Kp, N.stream())
    np.testing.assert_allclose(col.cpu().numpy()[:, :k * k * C], cols.detach().numpy(), atol=1e-6)
    assert (col.cpu()[:, k * k * C:] == 0).all()
    dcd = dcol.cuda()
    dx = torch.empty(B, H, W, C, device='cuda')
    stats = torch.zeros(2 * C, dtype=torch.float64, device='cuda')
    N.call('t3d_col2im_bwd', N.F32, N.ptr(dcd), N.ptr(xd), pro, N.ptr(dx), N.ptr(stats), B, H, W, C, k, s, pad, Kp, N.stream())
    torch.cuda.synchronize()
    ref = xr.grad / sc            # autograd gives d/dx through the affine; the kernel reports the gradient at the affine's OUTPUT
    np.testing.assert_allclose(dx.cpu().numpy(), ref.numpy(), atol=2e-5)
    np.testing.assert_allclose(stats[:C].cpu().numpy(), ref.sum((0, 1, 2)).double().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(stats[C:].cpu().numpy(), (ref * x).sum((0, 1, 2)).double().numpy(), rtol=1e-4, atol=1e-4)
    # bf16 storage through the same entries: inputs rounded to bf16, results within bf16 rounding of the fp32 ones
    xb, dcb = xd.bfloat16(), dcd.bfloat16()
    colb = torch.full((B * Ho * Wo, Kp), 7.0, device='cuda', dtype=torch.bfloat16)
    N.call('t3d_im2col', N.BF16, N.ptr(xb), pro, N.ptr(colb), B, H, W, C, k, s, pad, Kp, N.stream())
    col32 = torch.empty_like(col)
    xb32 = xb.float()
    N.call('t3d_im2col', N.F32, N.ptr(xb32), pro, N.ptr(col32), B, H, W, C, k, s, pad, Kp, N.stream())
    np.testing.assert_allclose(colb.float().cpu().numpy(), col32.cpu().numpy(), rtol=8e-3, atol=1e-6)
    dxb = torch.empty(B, H, W, C, device='cuda', dtype=torch.bfloat16)
    stb = torch.zeros(2 * C, dtype=torch.float64, device='cuda')
    N.call('t3d_col2im_bwd', N.BF16, N.ptr(dcb), N.ptr(xb), pro, N.ptr(dxb), N.ptr(stb), B, H, W, C, k, s, pad, Kp, N.stream())
    dc32, dx32 = dcb.float(), torch.empty(B, H, W, C, device='cuda')
    N.call('t3d_col2im_bwd', N.F32, N.ptr(dc32), N.ptr(xb32), pro, N.ptr(dx32), None, B, H, W, C, k, s, pad, Kp, N.stream())
    np.testing.assert_allclose(dxb.float().cpu().numpy(), dx32.cpu().numpy(), rtol=8e-3, atol=1e-6)
    np.testing.assert_allclose(stb[:C].cpu().numpy(), dxb.double().sum((0, 1, 2)).cpu().numpy(), rtol=1e-4, atol=1e-3)


def test_stem_patch_gather_from_nchw_matches_unfold():
    """t3d_im2col_nchw (the 7x7 / stride-2 stem reads the reference's fp32 NCHW input contract directly)."""
    import torch.nn.functional as F
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(5)
    for (B, H, W, C, k, s, pad, Kp) in ((3, 32, 30, 3, 7, 2, 3, 160), (2, 9, 9, 3, 3, 1, 1, 32), (1, 8, 8, 5, 3, 1, 1, 45)):
        x = torch.randn(B, C, H, W, generator=g)
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        cols = F.unfold(x, k, padding=pad, stride=s).view(B, C, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, k * k * C)
        xd = x.cuda()
        col = torch.full((B * Ho * Wo, Kp), 7.0, device='cuda')
        N.call('t3d_im2col_nchw', N.F32, N.ptr(xd), N.ptr(col), B, H, W, C, k, s, pad, Kp, N.stream())
        assert torch.equal(col.cpu()[:, :k * k * C], cols)
        assert (col.cpu()[:, k * k * C:] == 0).all()
        colb = torch.full((B * Ho * Wo, Kp), 7.0, device='cuda', dtype=torch.bfloat16)
        N.call('t3d_im2col_nchw', N.BF16, N.ptr(xd), N.ptr(colb), B, H, W, C, k, s, pad, Kp, N.stream())
        assert torch.equal(colb.cpu()[:, :k * k * C], cols.bfloat16())


@pytest.mark.parametrize('B,H,W,C', [(2, 11, 14, 16), (2, 11, 14, 12), (3, 23, 20, 64), (2, 9, 9, 2048)])
def test_maxpool_res_relu_subsample_match_torch_autograd(B, H, W, C):
    import torch.nn.functional as F
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(3)
    y = torch.randn(B, H, W, C, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    yr = y.clone().requires_grad_(True)
    out_ref = F.max_pool2d(F.relu(yr * sc + sh).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    Ho, Wo = out_ref.shape[1], out_ref.shape[2]
    dout = torch.randn(B, Ho, Wo, C, generator=g)
    out_ref.backward(dout)
    yd, scd, shd = y.cuda(), sc.cuda(), sh.cuda()
    pro = N.prologue(scd, shd, None, 'relu', False)
    out = torch.empty(B, Ho, Wo, C, device='cuda')
    idx = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device='cuda')
    N.call('t3d_maxpool_fwd', N.F32, N.ptr(yd), pro, N.ptr(out), N.ptr(idx), B, H, W, C, N.stream())
    np.testing.assert_allclose(out.cpu().numpy(), out_ref.detach().numpy(), atol=1e-6)
    dy = torch.empty(B, H, W, C, device='cuda')
    stats = torch.zeros(2 * C, dtype=torch.float64, device='cuda')
    dd = dout.cuda()
    N.call('t3d_maxpool_bwd', N.F32, N.ptr(dd), N.ptr(idx), N.ptr(yd), pro, N.ptr(dy), N.ptr(stats), B, H, W, C, N.stream())
    ref = yr.grad / sc
    np.testing.assert_allclose(dy.cpu().numpy(), ref.numpy(), atol=2e-5)
    np.testing.assert_allclose(stats[:C].cpu().numpy(), ref.sum((0, 1, 2)).double().numpy(), rtol=1e-4, atol=1e-4)
    # bottleneck tail, projection shortcut
    M = 300
    y3, ydn = torch.randn(M, C, generator=g), torch.randn(M, C, generator=g)
    s3, t3, ss, ts = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5,
                      torch.randn(C, generator=g))
    zr = F.relu(y3 * s3 + t3 + ydn * ss + ts)
    dz = torch.randn(M, C, generator=g)
    gr = dz * (zr > 0)
    dev = [t.cuda() for t in (y3, ydn, s3, t3, ss, ts, dz)]
    z = torch.empty(M, C, device='cuda')
    N.call('t3d_res_relu_fwd', N.F32, N.ptr(dev[0]), N.prologue(dev[2], dev[3], None, 'none', False), N.ptr(dev[1]),
           N.prologue(dev[4], dev[5], None, 'none', False), N.ptr(z), M, C, N.stream())
    np.testing.assert_allclose(z.cpu().numpy(), zr.numpy(), atol=1e-6)
    gg = torch.empty(M, C, device='cuda')
    st3, std = torch.zeros(2 * C, dtype=torch.float64, device='cuda'), torch.zeros(2 * C, dtype=torch.float64, device='cuda')
    N.call('t3d_res_relu_bwd', N.F32, N.ptr(dev[6]), N.ptr(z), N.ptr(dev[0]), N.ptr(dev[1]), N.ptr(gg), N.ptr(st3), N.ptr(std), M, C,
           N.stream())
    np.testing.assert_allclose(gg.cpu().numpy(), gr.numpy(), atol=1e-6)
    np.testing.assert_allclose(st3.cpu().numpy(), torch.cat([gr.sum(0), (gr * y3).sum(0)]).double().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(std.cpu().numpy(), torch.cat([gr.sum(0), (gr * ydn).sum(0)]).double().numpy(), rtol=1e-5, atol=1e-4)
    # identity shortcut (no projection branch, no second set of sums)
    zi = F.relu(y3 * s3 + t3 + ydn)
    N.call('t3d_res_relu_fwd', N.F32, N.ptr(dev[0]), N.prologue(dev[2], dev[3], None, 'none', False), N.ptr(dev[1]), None, N.ptr(z), M, C,
           N.stream())
    np.testing.assert_allclose(z.cpu().numpy(), zi.numpy(), atol=1e-6)
    st3.zero_()
    N.call('t3d_res_relu_bwd', N.F32, N.ptr(dev[6]), N.ptr(z), N.ptr(dev[0]), None, N.ptr(gg), N.ptr(st3), None, M, C, N.stream())
    gi = dz * (zi > 0)
    np.testing.assert_allclose(gg.cpu().numpy(), gi.numpy(), atol=1e-6)
    np.testing.assert_allclose(st3.cpu().numpy(), torch.cat([gi.sum(0), (gi * y3).sum(0)]).double().numpy(), rtol=1e-5, atol=1e-4)
    # stride-2 sampling and its adjoint
    x = torch.randn(B, H, W, C, generator=g)
    xd = x.cuda()
    sm = torch.empty(B, (H + 1) // 2, (W + 1) // 2, C, device='cuda')
    N.call('t3d_subsample', N.F32, N.ptr(xd), N.ptr(sm), B, H, W, C, 2, 0, N.stream())
    assert torch.equal(sm.cpu(), x[:, ::2, ::2])
    up = torch.full((B, H, W, C), 7.0, device='cuda')
    N.call('t3d_subsample', N.F32, N.ptr(sm), N.ptr(up), B, H, W, C, 2, 1, N.stream())
    refu = torch.zeros_like(x)
    refu[:, ::2, ::2] = x[:, ::2, ::2]
    assert torch.equal(up.cpu(), refu)


@pytest.mark.parametrize('B,H,C,Nn,s', [(4, 12, 128, 128, 1), (2, 13, 64, 64, 2)])
def test_dense_3x3_conv_as_gather_plus_gemm_matches_torch(B, H, C, Nn, s):
    """The dense 3x3 conv of a bottleneck end to end -- t3d_pack_conv_weight, t3d_im2col, t3d_pwconv_fwd, and backward
    t3d_pwconv_dgrad (into the patch matrix) + t3d_col2im_bwd, t3d_pwconv_wgrad + t3d_unpack_conv_grad -- against
    torch.nn.functional.conv2d autograd (fp32 storage; BatchNorm coefficients set to the identity)."""
    import torch.nn.functional as F
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(C + s)
    W = H
    x = torch.randn(B, H, W, C, generator=g)
    w = torch.randn(Nn, C, 3, 3, generator=g) / (9 * C) ** 0.5
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(F.relu(xr).permute(0, 3, 1, 2), wr, None, s, 1).permute(0, 2, 3, 1)
    Ho = y_ref.shape[1]
    dy = torch.randn(B, Ho, Ho, Nn, generator=g)
    y_ref.backward(dy)
    Kp, M2 = 9 * C, B * Ho * Ho
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda().reshape(M2, Nn).contiguous()
    one, zero = torch.ones(max(C, Nn), device='cuda'), torch.zeros(max(C, Nn), device='cuda')
    pro = N.prologue(one, zero, None, 'relu', False)
    wp = torch.empty(Nn, Kp, device='cuda')
    N.call('t3d_pack_conv_weight', N.F32, N.ptr(wd), N.ptr(wp), Nn, C, 3, Kp, N.stream())
    wpt = wp.t().contiguous()
    col = torch.empty(M2, Kp, device='cuda')
    N.call('t3d_im2col', N.F32, N.ptr(xd), pro, N.ptr(col), B, H, W, C, 3, s, 1, Kp, N.stream())
    y = torch.empty(M2, Nn, device='cuda')
    N.call('t3d_pwconv_fwd', N.F32, N.ptr(col), None, N.ptr(wp), None, N.ptr(y), None, M2, Ho * Ho, Kp, Nn, N.stream())
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.detach().reshape(M2, Nn).numpy(), atol=2e-5)
    bb = N.bnbwd(one, zero, zero, False)
    dcol = torch.empty(M2, Kp, device='cuda')
    N.call('t3d_pwconv_dgrad', N.F32, N.ptr(dyd), N.ptr(y), bb, N.ptr(wpt), None, None, None, N.ptr(dcol), None, None, M2, Ho * Ho,
           Kp, Nn, N.stream())
    dx = torch.empty(B, H, W, C, device='cuda')
    stats = torch.zeros(2 * C, dtype=torch.float64, device='cuda')
    N.call('t3d_col2im_bwd', N.F32, N.ptr(dcol), N.ptr(xd), pro, N.ptr(dx), N.ptr(stats), B, H, W, C, 3, s, 1, Kp, N.stream())
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), atol=5e-5)
    np.testing.assert_allclose(stats[:C].cpu().numpy(), xr.grad.sum((0, 1, 2)).double().numpy(), rtol=1e-4, atol=1e-4)
    dwp = torch.zeros(Nn, Kp, device='cuda')
    N.call('t3d_pwconv_wgrad', N.F32, N.ptr(dyd), N.ptr(y), bb, N.ptr(col), None, N.ptr(dwp), M2, Ho * Ho, Kp, Nn, N.stream())
    dw = torch.empty(Nn, C, 3, 3, device='cuda')
    N.call('t3d_unpack_conv_grad', N.ptr(dwp), N.ptr(dw), Nn, C, 3, Kp, N.stream())
    np.testing.assert_allclose(dw.cpu().numpy(), wr.grad.numpy(), atol=1e-4, rtol=1e-4)


def test_resnet50_train_step_with_implicit_gemm_3x3_layers(monkeypatch):
    """The opt-in gather-form implicit GEMMs (csrc/conv3x3.hip, T3D_IMPLICIT3=1) against the default patch-matrix path of the same
    engine: one bf16 train step, same weights and crops -- same loss and gradients up to the bf16 rounding of two summation
    orders."""
    from oracle.weights import make_inputs, make_state_dict
    from test_gpu_engine import _loss_cfg
    from torchdet3d import _native as N
    from torchdet3d.models import resnet as RM
    B, HW, nc = 8, 96, 9
    imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
    sd = make_state_dict('resnet50', nc)
    mask = torch.full((B, 2048), 2.0).cuda()
    cfg = _loss_cfg(['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2]))
    res = {}
    for on in (False, True):
        monkeypatch.setattr(RM, 'IMPLICIT3', on)
        net = RM.ResNetEngine('resnet50', nc, 'cuda', torch.bfloat16)
        net.load_state_dict(sd)
        kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=mask)
        assert bool(net._conv3) == on
        out = torch.zeros(16, device='cuda')
        dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
        gtd, cd = gt_kp.cuda().view(B, 18).contiguous(), cats.cuda()     # (kept alive: raw pointers cross the boundary)
        N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp), N.ptr(gtd), N.ptr(lg), N.ptr(cd), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc,
               N.stream())
        net.backward(dkp, dlg)
        torch.cuda.synchronize()
        res[on] = (out[0].item(), net.gflat.clone())
        del net
    (l0, g0), (l1, g1) = res[False], res[True]
    rel = ((g0 - g1).double().norm() / g0.double().norm()).item()
    print(f'   implicit 3x3: loss {l1:.6f} vs {l0:.6f}, whole-gradient relative L2 {rel:.3e}')
    assert abs(l0 - l1) < 2e-2 and rel < 0.15
