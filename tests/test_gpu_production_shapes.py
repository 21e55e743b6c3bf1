"""GPU parity of the hot kernels AT THEIR PRODUCTION LAUNCH SHAPES (BASELINE config 2: MobileNetV2, 224x224 crops,
batch 256, bf16 storage): the persistent multi-item loops with carried accumulators of the streaming depthwise
kernels, the workspace-split pointwise weight gradient and its partial-tile reduce, and the y-free expand-layer pair
only run in this form at B = 256 -- the small-shape tests in test_gpu_dwconv.py / test_gpu_pwconv.py never reach them.

The operands are generated on the device (a 256 x 112 x 112 x 96 tensor is 616 MB in bf16); the checker is torch-CPU
in fp64 on a SUBSAMPLE that still pins every output class:
  * depthwise: 8 of the C channels (depthwise channels are independent, so y / dx / dW / the BatchNorm sums of those
    channels are checked over the WHOLE batch);
  * pointwise forward / data gradient: 8192 pixel rows x all channels; the per-channel sums against the kernel's own
    stored output reduced in fp64;
  * pointwise weight gradient: the whole contraction over the 3.2 M pixels, in row chunks.
Tolerances (written per check): a stored bf16 value carries 2^-9 relative rounding, sums are held to 1e-5 relative
(fp64 accumulation of fp32 partials)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
B = 256


def _relu6(x):
    return x.clamp(0, 6)


def _sel(C):
    return torch.tensor(sorted({0, 1, C // 3, C // 2 - 1, C // 2, C - 9, C - 2, C - 1}))


def _gen(shape, seed, scale=1.0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return (torch.randn(shape, device='cuda', generator=g) * scale).to(BF)


def _nchw64(t, sel):
    """device NHWC bf16 -> CPU NCHW fp64 of the selected channels."""
    return t[..., sel.cuda()].permute(0, 3, 1, 2).double().cpu().contiguous()


@pytest.mark.parametrize('C,H,s', [(32, 112, 1), (96, 112, 2), (144, 56, 1)])
def test_dw3_forward_production_shape(C, H, s):
    from torchdet3d import _native as N
    x = _gen((B, H, H, C), 1 + C)
    g = torch.Generator().manual_seed(C)
    w = torch.randn(C, 9, generator=g) * 0.3
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    Ho = (H + 2 - 3) // s + 1
    y = torch.empty(B, Ho, Ho, C, device='cuda', dtype=BF)
    NREP = 16
    stats = torch.zeros(NREP, 2 * C, device='cuda', dtype=torch.float64)
    wd, sc, sh = w.cuda(), scale.cuda(), shift.cuda()
    pro = N.prologue(sc, sh, None, 'relu6', False)
    N.call('t3d_set_reduction_replicas', NREP, 2 * C)
    try:
        N.call('t3d_dwconv_fwd', N.BF16, N.ptr(x), pro, N.ptr(wd), N.ptr(y), N.ptr(stats), None, B, H, H, C, 3, s, N.stream())
    finally:
        N.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    sel = _sel(C)
    a = _relu6(_nchw64(x, sel) * scale[sel].double().view(1, -1, 1, 1) + shift[sel].double().view(1, -1, 1, 1))
    ref = F.conv2d(a, w[sel].double().view(-1, 1, 3, 3), None, s, 1, 1, len(sel))
    got = _nchw64(y, sel)
    # stored bf16 output: 2^-8 relative + the fp32 9-tap accumulation
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=8e-3, atol=8e-3 * ref.abs().max().item())
    st = stats.sum(0).cpu().view(2, C)[:, sel]
    np.testing.assert_allclose(st[0].numpy(), got.sum(dim=(0, 2, 3)).numpy(), rtol=1e-5, atol=1e-5 * got.abs().sum().item() / len(sel))
    np.testing.assert_allclose(st[1].numpy(), (got ** 2).sum(dim=(0, 2, 3)).numpy(), rtol=1e-5)


@pytest.mark.parametrize('C,H,s', [(32, 112, 1), (96, 112, 2), (144, 56, 1), (144, 56, 2)])
def test_dw3_backward_production_shape(C, H, s):
    """dx, dW and the producer's BatchNorm-backward sums of the fused depthwise backward at B = 256."""
    from torchdet3d import _native as N
    Ho = (H + 2 - 3) // s + 1
    x = _gen((B, H, H, C), 11 + C)
    dz = _gen((B, Ho, Ho, C), 12 + C)
    y2 = _gen((B, Ho, Ho, C), 13 + C)
    g = torch.Generator().manual_seed(C + s)
    w = torch.randn(C, 9, generator=g) * 0.3
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    alpha, beta, gamma = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2, torch.randn(C, generator=g) * 0.1
    NREP = 16
    dx = torch.empty(B, H, H, C, device='cuda', dtype=BF)
    stats = torch.zeros(NREP, 2 * C, device='cuda', dtype=torch.float64)
    dw = torch.zeros(NREP, C, 9, device='cuda')
    keep = [t.cuda() for t in (alpha, beta, gamma, scale, shift, w)]
    bb = N.bnbwd(keep[0], keep[1], keep[2], False)
    pro = N.prologue(keep[3], keep[4], None, 'relu6', False)
    N.call('t3d_set_reduction_replicas', NREP, 2 * C)
    try:
        N.call('t3d_dwconv_bwd', N.BF16, N.ptr(dz), N.ptr(y2), bb, N.ptr(keep[5]), N.ptr(x), pro, None, N.ptr(dx),
               N.ptr(stats), N.ptr(dw), B, H, H, C, 3, s, N.stream())
    finally:
        N.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    sel = _sel(C)
    v = lambda t: t[sel].double().view(1, -1, 1, 1)
    xs = _nchw64(x, sel)
    u = (xs * v(scale) + v(shift)).requires_grad_(True)
    wr = w[sel].double().view(-1, 1, 3, 3).requires_grad_(True)
    yref = F.conv2d(_relu6(u), wr, None, s, 1, 1, len(sel))
    dy = v(alpha) * _nchw64(dz, sel) + v(beta) * _nchw64(y2, sel) + v(gamma)
    yref.backward(dy)
    got = _nchw64(dx, sel)
    np.testing.assert_allclose(got.numpy(), u.grad.numpy(), rtol=8e-3, atol=8e-3 * u.grad.abs().max().item())
    gw = dw.sum(0).double().cpu()[sel]
    # 3.2 M-term fp32 sums in per-thread / per-block partials: 2e-4 of the largest entry
    np.testing.assert_allclose(gw.numpy(), wr.grad.view(-1, 9).numpy(), rtol=2e-3, atol=2e-4 * wr.grad.abs().max().item())
    st = stats.sum(0).cpu().view(2, C)[:, sel]
    n = B * H * H
    np.testing.assert_allclose(st[0].numpy(), got.sum(dim=(0, 2, 3)).numpy(), rtol=1e-5, atol=1e-4 * n ** .5)
    np.testing.assert_allclose(st[1].numpy(), (got * xs).sum(dim=(0, 2, 3)).numpy(), rtol=1e-5, atol=1e-4 * n ** .5)


def _rows(M, n=8192):
    g = torch.Generator().manual_seed(M)
    return torch.randperm(M, generator=g)[:n].sort().values


@pytest.mark.parametrize('K,Nn,HW,pro_act', [(16, 96, 112 * 112, None), (96, 24, 56 * 56, 'relu6'), (32, 16, 112 * 112, 'relu6')])
def test_pw_forward_production_shape(K, Nn, HW, pro_act):
    from torchdet3d import _native as N
    from test_gpu_pwconv import _pack
    M = B * HW
    x = _gen((M, K), 21 + K)
    g = torch.Generator().manual_seed(K + Nn)
    w = torch.randn(Nn, K, generator=g) / K ** .5
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    wd = _pack(w, BF)
    y = torch.empty(M, Nn, device='cuda', dtype=BF)
    NREP = 16
    stats = torch.zeros(NREP, 2 * Nn, device='cuda', dtype=torch.float64)
    sc, sh = scale.cuda(), shift.cuda()
    pro = N.prologue(sc, sh, None, pro_act, False) if pro_act else None
    N.call('t3d_set_reduction_replicas', NREP, 2 * Nn)
    try:
        N.call('t3d_pwconv_fwd', N.BF16, N.ptr(x), pro, N.ptr(wd), None, N.ptr(y), N.ptr(stats), M, HW, K, Nn, N.stream())
    finally:
        N.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    r = _rows(M)
    a = x[r.cuda()].double().cpu()
    if pro_act:
        a = _relu6(a * scale.double() + shift.double()).to(BF).double()      # MFMA operand is the bf16-rounded activation
    ref = a @ wd.double().cpu().t()
    got = y[r.cuda()].double().cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=8e-3, atol=8e-3 * ref.abs().max().item())
    st = stats.sum(0).cpu().view(2, Nn)
    yd = y.double()
    np.testing.assert_allclose(st[0].numpy(), yd.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
    np.testing.assert_allclose(st[1].numpy(), (yd * yd).sum(0).cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize('K,Nn,HW,act', [(96, 24, 56 * 56, 'relu6'), (144, 24, 56 * 56, 'relu6'), (32, 16, 112 * 112, 'relu6')])
def test_pw_dgrad_and_split_wgrad_production_shape(K, Nn, HW, act):
    """Projection layer K -> Nn at B = 256: data gradient (BatchNorm-backward affine on load, act' and the producer's sums
    in the epilogue) and the workspace-split weight gradient + partial-tile reduce."""
    from torchdet3d import _native as N
    from test_gpu_pwconv import _pack
    M = B * HW
    dz, y = _gen((M, Nn), 31 + K), _gen((M, Nn), 32 + K)
    xraw = _gen((M, K), 33 + K)
    g = torch.Generator().manual_seed(K * 3 + Nn)
    w = torch.randn(Nn, K, generator=g) / Nn ** .5
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    alpha, beta, gamma = torch.rand(Nn, generator=g) + 0.5, torch.randn(Nn, generator=g) * 0.2, torch.randn(Nn, generator=g) * 0.1
    keep = [t.cuda() for t in (alpha, beta, gamma, scale, shift)]
    bb = N.bnbwd(keep[0], keep[1], keep[2], False)
    pin = N.prologue(keep[3], keep[4], None, act, False)
    wt = _pack(w, BF, transpose=True)
    dx = torch.empty(M, K, device='cuda', dtype=BF)
    NREP = 16
    stats = torch.zeros(NREP, 2 * K, device='cuda', dtype=torch.float64)
    dw = torch.zeros(Nn, K, device='cuda')
    ws = torch.empty(64 << 20, device='cuda', dtype=torch.uint8)
    N.call('t3d_set_reduction_replicas', NREP, 2 * K)
    N.call('t3d_set_workspace', N.ptr(ws), ws.numel())
    try:
        N.call('t3d_pwconv_dgrad', N.BF16, N.ptr(dz), N.ptr(y), bb, N.ptr(wt), N.ptr(xraw), pin, None, N.ptr(dx),
               N.ptr(stats), None, M, HW, K, Nn, N.stream())
        N.call('t3d_pwconv_wgrad', N.BF16, N.ptr(dz), N.ptr(y), bb, N.ptr(xraw), pin, N.ptr(dw), M, HW, K, Nn, N.stream())
    finally:
        N.call('t3d_set_reduction_replicas', 1, 0)
        N.call('t3d_set_workspace', None, 0)
    torch.cuda.synchronize()
    wq = wt.double().cpu().t()                                   # [Nn, K] as the kernels see it
    r = _rows(M)
    rc = r.cuda()
    dy = (alpha.double() * dz[rc].double().cpu() + beta.double() * y[rc].double().cpu() + gamma.double()).to(BF).double()
    u = xraw[rc].double().cpu() * scale.double() + shift.double()
    ref = (dy @ wq) * ((u > 0) & (u < 6)).double()
    got = dx[rc].double().cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=8e-3, atol=8e-3 * ref.abs().max().item())
    st = stats.sum(0).cpu().view(2, K)
    dxd = dx.double()
    np.testing.assert_allclose(st[0].numpy(), dxd.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
    np.testing.assert_allclose(st[1].numpy(), (dxd * xraw.double()).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
    del dxd
    # weight gradient over ALL pixels, fp64 on the CPU in row chunks (operands rounded to bf16 as the MFMA sees them)
    acc = torch.zeros(Nn, K, dtype=torch.float64)
    step = 1 << 19
    for i in range(0, M, step):
        dyc = (alpha.double() * dz[i:i + step].double().cpu() + beta.double() * y[i:i + step].double().cpu()
               + gamma.double()).to(BF).double()
        ac = _relu6(xraw[i:i + step].double().cpu() * scale.double() + shift.double()).to(BF).double()
        acc += dyc.t() @ ac
    np.testing.assert_allclose(dw.double().cpu().numpy(), acc.numpy(), rtol=2e-3, atol=2e-4 * acc.abs().max().item())


def test_pw_yfree_pair_production_shape():
    """The y-free expand-layer backward at the shape it exists for: 16 -> 96 @ 112^2, B = 256 (M*N = 308 M elements)."""
    from torchdet3d import _native as N
    K, Nn, HW = 16, 96, 112 * 112
    M = B * HW
    x, dz, res = _gen((M, K), 41), _gen((M, Nn), 42), _gen((M, K), 43)
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(Nn, K, generator=g) / K ** .5).to(BF)
    alpha, beta, gamma = torch.rand(Nn, generator=g) + .5, torch.randn(Nn, generator=g) * .2, torch.randn(Nn, generator=g) * .1
    wd = w.cuda()
    keep = [t.cuda() for t in (alpha, beta, gamma)]
    bb = N.bnbwd(keep[0], keep[1], keep[2], False)
    NP, KP = (Nn + 31) // 32 * 32, (K + 31) // 32 * 32
    wcat = torch.empty(K, NP + KP, device='cuda', dtype=BF)
    cvec = torch.empty(K, device='cuda')
    dx = torch.empty(M, K, device='cuda', dtype=BF)
    dw = torch.zeros(Nn, K, device='cuda')
    ws = torch.empty(64 << 20, device='cuda', dtype=torch.uint8)
    N.call('t3d_set_workspace', N.ptr(ws), ws.numel())
    try:
        wtd = wd.t().contiguous()
        N.call('t3d_pwconv_yfree_prep', N.ptr(wtd), bb, N.ptr(wcat), N.ptr(cvec), K, Nn, N.stream())
        N.call('t3d_pwconv_dgrad_yfree', N.ptr(dz), N.ptr(x), N.ptr(wcat), N.ptr(cvec), None, None, N.ptr(res), N.ptr(dx),
               None, M, HW, K, Nn, N.stream())
        N.call('t3d_pwconv_wgrad_yfree', N.ptr(dz), N.ptr(x), bb, N.ptr(wd), N.ptr(dw), M, HW, K, Nn, N.stream())
    finally:
        N.call('t3d_set_workspace', None, 0)
    torch.cuda.synchronize()
    w64 = w.double()
    r = _rows(M)
    rc = r.cuda()
    xs = x[rc].double().cpu()
    dy = alpha.double() * dz[rc].double().cpu() + beta.double() * (xs @ w64.t()) + gamma.double()
    ref = dy @ w64 + res[rc].double().cpu()
    got = dx[rc].double().cpu()
    # Wcat is rounded to bf16 once more than the regular path: 1.5e-2 of the largest entry (as test_pw_yfree_backward)
    assert (got - ref).abs().max().item() < 1.5e-2 * ref.abs().max().item()
    acc = torch.zeros(Nn, K, dtype=torch.float64)
    step = 1 << 19
    for i in range(0, M, step):
        xc = x[i:i + step].double().cpu()
        dyc = alpha.double() * dz[i:i + step].double().cpu() + beta.double() * (xc @ w64.t()) + gamma.double()
        acc += dyc.t() @ xc
    assert (dw.double().cpu() - acc).abs().max().item() < 1e-2 * acc.abs().max().item()
