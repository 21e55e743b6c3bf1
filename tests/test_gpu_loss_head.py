"""GPU parity of the loss / metric reduction kernel against the reference's golden vectors
(tests/golden/losses.npz, metrics.npz: produced by importing the real reference, oracle/gen_golden.py)
and of the head kernels against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TERMS = {  # golden name -> (cfg field, extra cfg)
    'l1': ('c_l1', {}), 'mse': ('c_mse', {}), 'smoothl1': ('c_smoothl1', dict(smoothl1_beta=0.2)),
    'add_loss': ('c_add', {}), 'diag_loss': ('c_diag', {}), 'wing_default': ('c_wing', dict(wing_w=0.05, wing_eps=2.)),
    'wing_cfg': ('c_wing', dict(wing_w=5.18, wing_eps=1.)), 'wing_quirk': ('c_wing', dict(wing_w=0.3, wing_eps=0.05))}


def _run(cfg, p, t, logits, cats):
    from torchdet3d import _native as N
    B = p.shape[0]
    nc = logits.shape[1] if logits is not None else 1
    out = torch.zeros(16, device='cuda')
    dkp = torch.empty(B, 18, device='cuda')
    dlg = torch.empty(B, nc, device='cuda') if logits is not None else None
    pd, td = p.cuda().view(B, 18).contiguous(), t.cuda().view(B, 18).contiguous()
    ld = logits.cuda().contiguous() if logits is not None else None
    cd = cats.cuda()                 # (kept alive: raw pointers cross the boundary)
    N.call('t3d_loss_fwd_bwd', cfg, N.ptr(pd), N.ptr(td), N.ptr(ld), N.ptr(cd), N.ptr(out), N.ptr(dkp),
           N.ptr(dlg), B, nc, N.stream())
    torch.cuda.synchronize()
    return out.cpu(), dkp.cpu().view(B, 9, 2), (dlg.cpu() if dlg is not None else None)


def _cfg(**kw):
    from torchdet3d import _native as N
    c = N.LossCfg()
    c.smoothl1_beta, c.wing_w, c.wing_eps, c.lam_reg, c.lam_cls = 0.2, 5.18, 1.0, 1.0, 1.0
    for k, v in kw.items():
        setattr(c, k, v)
    return c


@pytest.mark.parametrize('B', [256, 7])
def test_each_loss_term_matches_reference_golden(golden_dir, B):
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    p, t = torch.from_numpy(g[f'p{B}']), torch.from_numpy(g[f't{B}'])
    logits, cats = torch.from_numpy(g[f'logits{B}']), torch.from_numpy(g[f'cats{B}'])
    for name, (field, extra) in TERMS.items():
        out, dkp, _ = _run(_cfg(**{field: 1.0}, **extra), p, t, None, cats)
        np.testing.assert_allclose(out[0].item(), g[f'{name}:{B}:val'], rtol=3e-6, err_msg=name)
        np.testing.assert_allclose(dkp.numpy(), g[f'{name}:{B}:grad'], atol=1e-8, rtol=2e-5, err_msg=name)
    out, dkp, dlg = _run(_cfg(c_ce=1.0), p, t, logits, cats)
    np.testing.assert_allclose(out[0].item(), g[f'ce:{B}:val'], rtol=3e-6)
    np.testing.assert_allclose(dlg.numpy(), g[f'ce:{B}:grad'], atol=2e-8, rtol=2e-5)
    assert dkp.abs().max() == 0


def test_weighted_sum_and_lambdas(golden_dir):
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    B = 256
    p, t = torch.from_numpy(g[f'p{B}']), torch.from_numpy(g[f't{B}'])
    logits, cats = torch.from_numpy(g[f'logits{B}']), torch.from_numpy(g[f'cats{B}'])
    out, dkp, dlg = _run(_cfg(c_l1=1.0, c_add=0.1, c_ce=0.2, lam_reg=0.7, lam_cls=0.4), p, t, logits, cats)
    reg = g['l1:256:val'] + 0.1 * g['add_loss:256:val']
    np.testing.assert_allclose(out[1].item(), reg, rtol=3e-6)
    np.testing.assert_allclose(out[2].item(), 0.2 * g['ce:256:val'], rtol=3e-6)
    np.testing.assert_allclose(out[0].item(), 0.7 * reg + 0.4 * 0.2 * g['ce:256:val'], rtol=3e-6)
    np.testing.assert_allclose(dkp.numpy(), 0.7 * (g['l1:256:grad'] + 0.1 * g['add_loss:256:grad']), atol=1e-8, rtol=2e-5)
    np.testing.assert_allclose(dlg.numpy(), 0.4 * 0.2 * g['ce:256:grad'], atol=2e-8, rtol=2e-5)


def test_metrics_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'metrics.npz'))
    p, t = torch.from_numpy(g['p']), torch.from_numpy(g['t'])
    logits, cats = torch.from_numpy(g['logits']), torch.from_numpy(g['cats'])
    out, _, _ = _run(_cfg(c_l1=1.0), p, t, logits, cats)
    np.testing.assert_allclose(out[3:5].numpy(), g['add_mean'], rtol=3e-6)
    np.testing.assert_allclose(out[6:8].numpy(), g['add_sum'], rtol=3e-6)
    assert out[5].item() == pytest.approx(float(g['acc_mean'])) and out[8].item() == float(g['acc_sum'])  # bit-exact count


@pytest.mark.parametrize('B,F,nc,with_pro', [(5, 1280, 9, False), (12, 1024, 9, True), (3, 1280, 1, False)])
def test_head_fwd_bwd(B, F, nc, with_pro):
    from oracle.model import act_fn
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(B + F)
    f = torch.randn(B, F, generator=g)
    cats = torch.randint(0, 9, (B,), generator=g)
    wreg = (torch.randn(9, 18, F, generator=g) / F ** .5).requires_grad_(True)
    breg = torch.randn(9, 18, generator=g).requires_grad_(True)
    wcls = (torch.randn(nc, F, generator=g) / F ** .5).requires_grad_(True)
    bcls = torch.randn(nc, generator=g).requires_grad_(True)
    mask = (torch.rand(B, F, generator=g) >= .5).float() * 2
    scale, shift = torch.rand(F, generator=g) + .5, torch.randn(F, generator=g) * .3
    u = (f * scale + shift).requires_grad_(True) if with_pro else f.clone().requires_grad_(True)
    fa = act_fn(u, 'hswish') if with_pro else u
    kp = torch.sigmoid(torch.stack([wreg[c] @ fa[b] + breg[c] for b, c in enumerate(cats)]))
    lg = (fa * mask) @ wcls.t() + bcls
    dkp, dlg = torch.randn(B, 18, generator=g), torch.randn(B, nc, generator=g)
    ((kp * dkp).sum() + ((lg * dlg).sum() if nc > 1 else 0)).backward()

    d = lambda x: x.detach().cuda().contiguous()
    fd, cd, md = d(f), cats.cuda(), d(mask)
    keep = (d(scale), d(shift))
    pro = N.prologue(keep[0], keep[1], None, 'hswish', False) if with_pro else None
    kpd = torch.empty(B, 18, device='cuda')
    lgd = torch.empty(B, nc, device='cuda') if nc > 1 else None
    wr, br, wc, bc = d(wreg), d(breg), d(wcls), d(bcls)
    N.call('t3d_head_fwd', N.ptr(fd), pro, N.ptr(cd), N.ptr(wr), N.ptr(br), N.ptr(wc), N.ptr(bc), N.ptr(md),
           N.ptr(kpd), N.ptr(lgd), B, F, nc, N.stream())
    np.testing.assert_allclose(kpd.cpu().numpy(), kp.detach().numpy(), atol=2e-6)
    if nc > 1:
        np.testing.assert_allclose(lgd.cpu().numpy(), lg.detach().numpy(), atol=2e-5)
    dpre, df = torch.empty(B, 18, device='cuda'), torch.empty(B, F, device='cuda')
    stats = torch.zeros(2 * F, device='cuda', dtype=torch.float64)
    dwr, dbr = torch.zeros_like(wr), torch.zeros_like(br)      # accumulated into: the caller zeroes once per step
    dwc, dbc = torch.zeros_like(wc), torch.zeros_like(bc)
    dkpd, dlgd = d(dkp), d(dlg)     # keep the device copies alive across the call
    N.call('t3d_head_bwd', N.ptr(fd), pro, N.ptr(cd), N.ptr(wr), N.ptr(wc), N.ptr(md), N.ptr(kpd), N.ptr(dkpd),
           N.ptr(dlgd) if nc > 1 else None, N.ptr(dpre), N.ptr(df), N.ptr(stats) if with_pro else None, N.ptr(dwr),
           N.ptr(dbr), N.ptr(dwc), N.ptr(dbc), B, F, nc, N.stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(df.cpu().numpy(), u.grad.numpy(), atol=3e-6, rtol=1e-4)
    np.testing.assert_allclose(dwr.cpu().numpy(), wreg.grad.numpy(), atol=3e-6, rtol=1e-4)
    np.testing.assert_allclose(dbr.cpu().numpy(), breg.grad.numpy(), atol=3e-6, rtol=1e-4)
    if nc > 1:
        np.testing.assert_allclose(dwc.cpu().numpy(), wcls.grad.numpy(), atol=3e-6, rtol=1e-4)
        np.testing.assert_allclose(dbc.cpu().numpy(), bcls.grad.numpy(), atol=3e-6, rtol=1e-4)
    if with_pro:
        np.testing.assert_allclose(stats[:F].cpu().numpy(), u.grad.double().sum(0).numpy(), atol=1e-5)
        np.testing.assert_allclose(stats[F:].cpu().numpy(), (u.grad.double() * f.double()).sum(0).numpy(), atol=1e-5)


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('B,C,R,HW', [(256, 960, 240, 49), (5, 72, 24, 196), (70, 16, 8, 3136), (33, 672, 168, 196),
                                      (9, 1000, 1024, 1)])
def test_se_gate_fwd_bwd(B, C, R, HW, fused):
    """t3d_se_fwd / t3d_se_bwd (SELayer, mobilenetv3.py:92-107) and their one-launch forms (t3d_se_fwd_fused,
    t3d_se_bwd_data + t3d_se_bwd_weights: what the engine runs) against torch autograd of the same two FCs."""
    import torch.nn.functional as F
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(B + C)
    rnd = lambda *s: torch.randn(*s, generator=g)
    gap, scale, shift = rnd(B, C) * HW ** .5, torch.rand(C, generator=g) + .5, rnd(C) * .2
    w1, b1, w2, b2 = rnd(R, C) / C ** .5, rnd(R) * .1, rnd(C, R) / R ** .5, rnd(C) * .5
    ps = rnd(B, C, 2)
    w1r, b1r, w2r, b2r = (t.clone().requires_grad_(True) for t in (w1, b1, w2, b2))
    m = (scale * gap / HW + shift).requires_grad_(True)
    h = F.relu(F.linear(m, w1r, b1r))
    q = F.linear(h, w2r, b2r)
    s = F.relu6(q + 3) / 6
    ds = scale * ps[..., 1] + shift * ps[..., 0]
    (s * ds).sum().backward()
    d = lambda t: t.contiguous().cuda()
    gd, scd, shd, w1d, b1d, w2d, b2d, psd = map(d, (gap, scale, shift, w1, b1, w2, b2, ps))
    mo, ho, qo, so = (torch.empty(B, n, device='cuda') for n in (C, R, C, C))
    if fused:
        w1t, w2t = w1d.t().contiguous(), w2d.t().contiguous()
        N.call('t3d_se_fwd_fused', N.ptr(gd), N.ptr(scd), N.ptr(shd), N.ptr(w1t), N.ptr(b1d), N.ptr(w2t), N.ptr(b2d), N.ptr(mo),
               N.ptr(ho), N.ptr(qo), N.ptr(so), B, C, R, HW, N.stream())
    else:
        N.call('t3d_se_fwd', N.ptr(gd), N.ptr(scd), N.ptr(shd), N.ptr(w1d), N.ptr(b1d), N.ptr(w2d), N.ptr(b2d), N.ptr(mo),
               N.ptr(ho), N.ptr(qo), N.ptr(so), B, C, R, HW, N.stream())
    go, dq, dp = torch.empty(B, C, device='cuda'), torch.empty(B, C, device='cuda'), torch.empty(B, R, device='cuda')
    stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    dw1, db1, dw2, db2 = (torch.empty_like(t) for t in (w1d, b1d, w2d, b2d))
    if fused:
        N.call('t3d_se_bwd_data', N.ptr(psd), N.ptr(gd), N.ptr(scd), N.ptr(shd), N.ptr(w1d), N.ptr(w2d), N.ptr(ho),
               N.ptr(qo), N.ptr(so), N.ptr(go), N.ptr(dq), N.ptr(dp), N.ptr(stats), B, C, R, HW, N.stream())
        N.call('t3d_se_bwd_weights', N.ptr(mo), N.ptr(ho), N.ptr(dq), N.ptr(dp), N.ptr(dw1), N.ptr(db1), N.ptr(dw2),
               N.ptr(db2), B, C, R, N.stream())
    else:
        N.call('t3d_se_bwd', N.ptr(psd), N.ptr(gd), N.ptr(scd), N.ptr(shd), N.ptr(w1d), N.ptr(w2d), N.ptr(mo), N.ptr(ho),
               N.ptr(qo), N.ptr(so), N.ptr(go), N.ptr(dq), N.ptr(dp), N.ptr(stats), N.ptr(dw1), N.ptr(db1), N.ptr(dw2),
               N.ptr(db2), B, C, R, HW, N.stream())
    torch.cuda.synchronize()
    close = lambda a, b, tol=2e-5: np.testing.assert_allclose(a.cpu().numpy(), b.detach().numpy(), rtol=1e-4,
                                                              atol=tol * max(1., b.detach().abs().max().item()))
    close(mo, m); close(ho, h); close(qo, q); close(so, s)
    close(go, m.grad / HW); close(dw1, w1r.grad); close(db1, b1r.grad); close(dw2, w2r.grad); close(db2, b2r.grad)
    gref = (m.grad / HW).detach()
    st0 = (s.detach() * ps[..., 0] + HW * gref).double().sum(0)
    st1 = (s.detach() * ps[..., 1] + gref * gap).double().sum(0)
    np.testing.assert_allclose(stats[:C].cpu().numpy(), st0.numpy(), rtol=1e-4, atol=1e-3 * B ** .5)
    np.testing.assert_allclose(stats[C:].cpu().numpy(), st1.numpy(), rtol=1e-4, atol=1e-3 * B ** .5 * HW ** .5)
