"""GPU parity: depthwise conv forward + BN statistics through the C ABI vs the torch-CPU oracle ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to('cuda', dtype)


def _act(x, kind):
    from oracle.model import act_fn
    return act_fn(x, kind)


SHAPES = [  # B, C, H, W, k, s
    (2, 32, 24, 24, 3, 1), (2, 96, 24, 24, 3, 2), (3, 72, 13, 17, 5, 2), (2, 120, 12, 12, 5, 1),
    (2, 16, 48, 48, 3, 1), (4, 960, 3, 3, 5, 1), (2, 200, 6, 6, 3, 1), (1, 8, 7, 7, 3, 1), (2, 144, 56, 56, 3, 1),
    (2, 672, 14, 14, 5, 2), (3, 576, 6, 6, 3, 1), (2, 960, 3, 3, 3, 1), (2, 384, 12, 12, 3, 2), (5, 264, 9, 7, 3, 1),
    (3, 40, 13, 17, 3, 2), (2, 576, 7, 7, 3, 2), (1, 24, 2, 2, 3, 2), (2, 144, 56, 56, 3, 2),
    # 5x5 on 7x7 planes: a thread per (image, channel pair) holds the plane (csrc/dwconv5_plane7.hip); 72 channels = a partial slab
    (5, 960, 7, 7, 5, 1), (3, 72, 7, 7, 5, 1), (9, 576, 7, 7, 5, 1),
    # 5x5 stride 1 as register tiles (csrc/dwconv5_tile.hip): whole tiles, tiles hanging over both edges, a partial channel slab
    (2, 120, 28, 28, 5, 1), (3, 40, 14, 14, 5, 1), (2, 48, 9, 13, 5, 1), (2, 72, 56, 56, 5, 2), (3, 24, 10, 15, 5, 2)]


@pytest.mark.parametrize('B,C,H,W,k,s', SHAPES)
@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('mode', ['plain', 'bnact', 'se_pre', 'se_post'])
def test_dwconv_fwd(B, C, H, W, k, s, dt, mode):
    from torchdet3d import _native as N
    dtype = torch.float32 if dt == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(B * 1000 + C + k + s)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    scale = torch.rand(C, generator=g) + 0.5
    shift = torch.randn(C, generator=g) * 0.3
    se = torch.rand(B, C, generator=g)
    act = {'plain': 'none', 'bnact': 'hswish', 'se_pre': 'relu', 'se_post': 'relu6'}[mode]
    xq = x.to(dtype).float()                      # what the kernel actually reads
    if mode == 'plain':
        a = xq
    else:
        u = xq * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)
        if mode == 'se_pre':
            a = _act(u * se.view(B, C, 1, 1), act)
        elif mode == 'se_post':
            a = _act(u, act) * se.view(B, C, 1, 1)
        else:
            a = _act(u, act)
    if dt == 'bf16':
        a = a.to(dtype).float()                   # activated tile is parked in LDS in the storage dtype
    ref = F.conv2d(a, w, None, s, (k - 1) // 2, 1, C)
    Ho, Wo = ref.shape[2:]
    xd = _nhwc(x, dtype)
    wd = w.view(C, k * k).contiguous().cuda()
    y = torch.empty(B, Ho, Wo, C, device='cuda', dtype=dtype)
    stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    gap = torch.zeros(B, C, device='cuda')
    keep = [t.cuda() for t in (scale, shift, se)]
    p = None
    if mode != 'plain':
        p = N.prologue(keep[0], keep[1], keep[2] if mode.startswith('se') else None, act, mode == 'se_post')
    N.call('t3d_dwconv_fwd', N.dtype_code(xd), N.ptr(xd), p, N.ptr(wd), N.ptr(y), N.ptr(stats), N.ptr(gap),
           B, H, W, C, k, s, N.stream())
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    tol = 2e-5 if dt == 'f32' else 2e-2
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=tol * max(1.0, ref.abs().max().item()), rtol=tol)
    # statistics are those of the STORED values
    n = B * Ho * Wo
    st = stats.cpu().view(2, C)
    np.testing.assert_allclose(st[0].numpy(), got.double().sum(dim=(0, 2, 3)).numpy(), rtol=1e-5, atol=1e-4 * n ** .5)
    np.testing.assert_allclose(st[1].numpy(), (got.double() ** 2).sum(dim=(0, 2, 3)).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(gap.cpu().numpy(), got.sum(dim=(2, 3)).numpy(), rtol=1e-4, atol=1e-3)


def test_bn_finalize_matches_batchnorm():
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(3)
    C, n = 40, 4 * 6 * 6
    y = torch.randn(4, C, 6, 6, generator=g) * 2 + 0.7
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    ref = F.batch_norm(y, rm0, rv0, gamma, beta, True, 0.1, 1e-5)
    stats = torch.cat([y.double().sum(dim=(0, 2, 3)), (y.double() ** 2).sum(dim=(0, 2, 3))]).cuda()
    d = [t.cuda() for t in (gamma, beta, rm, rv)]
    nbt = torch.zeros((), dtype=torch.int64, device='cuda')
    out = [torch.empty(C, device='cuda') for _ in range(4)]
    N.call('t3d_bn_finalize', N.ptr(stats), C, float(n), N.ptr(d[0]), N.ptr(d[1]), N.ptr(d[2]), N.ptr(d[3]),
           N.ptr(nbt), 0.1, 1e-5, *[N.ptr(o) for o in out], N.stream())
    torch.cuda.synchronize()
    scale, shift = out[0].cpu(), out[1].cpu()
    got = y * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5)
    np.testing.assert_allclose(d[2].cpu().numpy(), rm0.numpy(), atol=1e-6)
    np.testing.assert_allclose(d[3].cpu().numpy(), rv0.numpy(), atol=1e-5)
    assert int(nbt) == 1
    N.call('t3d_bn_eval_affine', C, N.ptr(d[0]), N.ptr(d[1]), N.ptr(d[2]), N.ptr(d[3]), 1e-5,
           N.ptr(out[0]), N.ptr(out[1]), N.stream())
    torch.cuda.synchronize()
    ref_e = F.batch_norm(y, rm0, rv0, gamma, beta, False, 0.1, 1e-5)
    got_e = y * out[0].cpu().view(1, C, 1, 1) + out[1].cpu().view(1, C, 1, 1)
    np.testing.assert_allclose(got_e.numpy(), ref_e.numpy(), atol=2e-5)


@pytest.mark.parametrize('B,C,H,W,k,s', SHAPES)
@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('mode', ['plain_res', 'bnact', 'bnact_ps'])
def test_dwconv_bwd(B, C, H, W, k, s, dt, mode):
    """dx, dw and the BN-backward sums vs torch autograd of the same (pre-rounded) operands."""
    from torchdet3d import _native as N
    dtype = torch.float32 if dt == 'f32' else torch.bfloat16
    q = lambda t: t.to(dtype).float()
    g = torch.Generator().manual_seed(B * 100 + C + k * 7 + s)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    act = 'hswish' if mode != 'plain_res' else 'none'
    xq = q(x)
    u = (xq * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)) if mode != 'plain_res' else xq
    u = u.clone().requires_grad_(True)
    a1 = _act(u, act)
    tiled = k != 3    # the tiled kernel (k=5) stages a and dy in LDS at storage precision; the 3x3 streaming kernels keep fp32
    a1q = a1 + (q(a1) - a1).detach() if tiled else a1
    wr = w.clone().requires_grad_(True)
    yref = F.conv2d(a1q, wr, None, s, (k - 1) // 2, 1, C)
    Ho, Wo = yref.shape[2:]
    dz, y2 = torch.randn(B, C, Ho, Wo, generator=g), torch.randn(B, C, Ho, Wo, generator=g)
    ps = mode == 'bnact_ps'
    shp = (B, C, 1, 1) if ps else (1, C, 1, 1)
    alpha, gamma = torch.rand(shp, generator=g) + 0.5, torch.randn(shp, generator=g) * 0.1
    beta = torch.randn(1, C, 1, 1, generator=g) * 0.2
    dy = alpha * q(dz) + beta * q(y2) + gamma
    yref.backward(q(dy) if tiled else dy)
    res = torch.randn(B, C, H, W, generator=g)
    dx_ref = u.grad + (q(res) if mode == 'plain_res' else 0)
    d = lambda t: _nhwc(t, dtype)
    dzd, yd, xd, rd = d(dz), d(y2), d(x), d(res)
    keep = [t.reshape(-1, C).squeeze(0).contiguous().cuda() for t in (alpha, beta, gamma)] + [scale.cuda(), shift.cuda()]
    bb = N.bnbwd(keep[0], keep[1], keep[2], ps)
    pro = None if mode == 'plain_res' else N.prologue(keep[3], keep[4], None, act, False)
    wd = w.view(C, k * k).contiguous().cuda()
    dx = torch.empty(B, H, W, C, device='cuda', dtype=dtype)
    stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    dw = torch.zeros(C, k * k, device='cuda')
    N.call('t3d_dwconv_bwd', N.dtype_code(xd), N.ptr(dzd), N.ptr(yd), bb, N.ptr(wd), N.ptr(xd), pro,
           N.ptr(rd) if mode == 'plain_res' else None, N.ptr(dx), N.ptr(stats) if pro is not None else None,
           N.ptr(dw), B, H, W, C, k, s, N.stream())
    torch.cuda.synchronize()
    got = dx.float().cpu().permute(0, 3, 1, 2)
    tol = 3e-5 if dt == 'f32' else 2e-2
    np.testing.assert_allclose(got.numpy(), dx_ref.numpy(), atol=tol * max(1., dx_ref.abs().max().item()), rtol=tol)
    tolw = 3e-5 if dt == 'f32' else 3e-3
    np.testing.assert_allclose(dw.cpu().view(C, 1, k, k).numpy(), wr.grad.numpy(),
                               atol=tolw * max(1., wr.grad.abs().max().item()), rtol=tolw)
    if pro is not None:
        st = stats.cpu().view(2, C)
        n = B * H * W
        np.testing.assert_allclose(st[0].numpy(), got.double().sum(dim=(0, 2, 3)).numpy(), rtol=1e-5, atol=1e-4 * n ** .5)
        np.testing.assert_allclose(st[1].numpy(), (got.double() * xq.double()).sum(dim=(0, 2, 3)).numpy(), rtol=1e-5,
                                   atol=1e-4 * n ** .5)


@pytest.mark.parametrize('B,C,H,W,k,s', [(8, 32, 56, 56, 3, 1), (4, 144, 28, 28, 3, 1), (4, 192, 28, 28, 3, 2), (2, 960, 7, 7, 3, 1),
                                         (4, 120, 28, 28, 5, 1), (4, 240, 14, 14, 5, 2), (2, 48, 14, 14, -5, 1), (6, 960, 7, 7, 5, 1)])
def test_dwconv_bwd_weight_gradient_slots_are_exactly_reproducible(B, C, H, W, k, s, monkeypatch):
    """t3d_set_dw_slots (include/t3d.h): every workgroup stores its partial depthwise weight gradient into its own slot and
    t3d_sum_slots_batched adds the used slots in index order -- same result as the atomic replica form to fp32 rounding,
    bit-identical from launch to launch (the atomic form is not), `used` = the slots the launch filled (or the replica count
    for the kernels without slot support: k = -5 runs the 5x5 case through the LDS-tiled fallback, T3D_DW_TILED=1)."""
    from torchdet3d import _native as N
    tiled = k < 0
    k = abs(k)
    if tiled:
        monkeypatch.setenv('T3D_DW_TILED', '1')
    g = torch.Generator().manual_seed(C + k)
    dtype = torch.bfloat16
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    x, res = _nhwc(torch.randn(B, C, H, W, generator=g), dtype), None
    dz, y = _nhwc(torch.randn(B, C, Ho, Wo, generator=g), dtype), _nhwc(torch.randn(B, C, Ho, Wo, generator=g), dtype)
    w = (torch.randn(C, k * k, generator=g) * 0.3).cuda()
    keep = [t.cuda() for t in (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2, torch.randn(C, generator=g) * 0.1,
                               torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3)]
    bb, pro = N.bnbwd(keep[0], keep[1], keep[2], False), N.prologue(keep[3], keep[4], None, 'relu6', False)
    dx = torch.empty(B, H, W, C, device='cuda', dtype=dtype)

    def launch(dw, stats):
        N.call('t3d_dwconv_bwd', N.BF16, N.ptr(dz), N.ptr(y), bb, N.ptr(w), N.ptr(x), pro, None, N.ptr(dx), N.ptr(stats), N.ptr(dw),
               B, H, W, C, k, s, N.stream())
    ref, st_ref = torch.zeros(C, k * k, device='cuda'), torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    launch(ref, st_ref)                                           # plain form: one replica, atomics
    SLOTS = 512
    results = []
    try:
        for rep in range(3):
            slots = torch.full((SLOTS, C, k * k), float('nan'), device='cuda')
            slots[:16].zero_()                                    # what a launch without slot support adds into
            used = torch.zeros(1, dtype=torch.int32, device='cuda')
            st = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
            N.call('t3d_set_reduction_replicas', 16, 2 * C)
            N.call('t3d_set_dw_slots', SLOTS, N.ptr(used))
            st16 = torch.zeros(16, 2 * C, device='cuda', dtype=torch.float64)
            launch(slots, st16)
            out = torch.empty(C, k * k, device='cuda')
            desc = torch.tensor([[slots.data_ptr(), out.data_ptr(), C * k * k, used.data_ptr()]], dtype=torch.int64, device='cuda')
            N.call('t3d_sum_slots_batched', N.ptr(desc), 1, N.stream())
            torch.cuda.synchronize()
            results.append((out.clone(), int(used.item()), st16.sum(0)))
    finally:
        N.call('t3d_set_dw_slots', 0, None)
        N.call('t3d_set_reduction_replicas', 1, 0)
    out, nused, st = results[0]
    assert 1 <= nused <= SLOTS and (not tiled or nused == 16)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4 * ref.abs().max().item())
    np.testing.assert_allclose(st.cpu().numpy(), st_ref.cpu().numpy(), rtol=1e-9, atol=1e-9)
    if not tiled:
        for o, n, _ in results[1:]:
            assert n == nused and torch.equal(o, out)


@pytest.mark.parametrize('s', [1, 2])
def test_dwconv_bwd_refused_by_the_streaming_kernel_keeps_its_pending_finalize(s):
    """a tensor of >= 2 GB is refused by the streaming 3x3 backward (32-bit buffer offsets) and goes to the
    tiled kernel -- which must still find the pending BatchNorm-backward finalize request (round 4 consumed it in front of the
    refusal: the fallback then read alpha / beta / gamma before anyone had computed them).  bf16, 128 x 112 x 112 x 672 = 2.16 GB
    per tensor; compared with the same launch after a standalone t3d_bn_bwd_finalize: identical coefficients and gradients."""
    from torchdet3d import _native as N
    B, H, W, C = 128, 112, 112, 672
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    assert B * H * W * C * 2 >= 1 << 31
    g = torch.Generator(device='cuda').manual_seed(4)
    x = torch.randn(B * H * W, C, device='cuda', generator=g).to(torch.bfloat16)
    dz = (torch.randn(B * Ho * Wo, C, device='cuda', generator=g) * 0.1).to(torch.bfloat16)
    y = torch.randn(B * Ho * Wo, C, device='cuda', generator=g).to(torch.bfloat16)
    w = torch.randn(C, 9, device='cuda', generator=g) * 0.3
    sc, sh = torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.2
    pro = N.prologue(sc, sh, None, 'relu6', False)
    gamma, mean, invstd = torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5
    count = float(B * Ho * Wo)
    bst = torch.zeros(1, 2 * C, device='cuda', dtype=torch.float64)           # sum(dz), sum(dz * y) of the gradient's BatchNorm
    bst[0, :C] = dz.double().sum(0)
    bst[0, C:] = (dz.double() * y.double()).sum(0)
    res = {}
    for lazy in (False, True):
        al, be, ga = (torch.full((C,), float('nan'), device='cuda') for _ in range(3))
        dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        bb = N.bnbwd(al, be, ga, False)
        dx = torch.empty_like(x)
        dw = torch.zeros(C, 9, device='cuda')
        N.call('t3d_set_reduction_replicas', 1, 2 * C)
        try:
            if lazy:
                f = N.BnFold()
                f.kind, f.C, f.count, f.nrep, f.rstride = 2, C, count, 1, 2 * C
                f.gamma, f.stats, f.mean, f.invstd = N.ptr(gamma), N.ptr(bst), N.ptr(mean), N.ptr(invstd)
                f.o0, f.o1, f.o2, f.o3, f.o4 = N.ptr(al), N.ptr(be), N.ptr(ga), N.ptr(dg), N.ptr(db)
                desc = torch.frombuffer(bytearray(bytes(f)), dtype=torch.uint8).cuda()
                N.call('t3d_fold_request', N.ptr(desc), N.ptr(al))
            else:
                N.call('t3d_bn_bwd_finalize', N.ptr(bst), C, count, N.ptr(gamma), N.ptr(mean), N.ptr(invstd), N.ptr(al), N.ptr(be),
                       N.ptr(ga), N.ptr(dg), N.ptr(db), N.stream())
            N.call('t3d_dwconv_bwd', N.BF16, N.ptr(dz), N.ptr(y), bb, N.ptr(w), N.ptr(x), pro, None, N.ptr(dx), None, N.ptr(dw),
                   B, H, W, C, 3, s, N.stream())
            assert not N.lib().t3d_fold_pending()
        finally:
            N.call('t3d_set_reduction_replicas', 1, 0)
        torch.cuda.synchronize()
        res[lazy] = (al.clone(), be.clone(), ga.clone(), dx[:4 * H * W].clone(), dx[-H * W:].clone(), dw.clone())
        del dx
    for a, b in zip(res[False][:5], res[True][:5]):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    # (the weight gradient of the tiled kernel without slots is a sum of fp32 atomics: equal up to their order)
    assert torch.allclose(res[False][5], res[True][5], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('B,C,H,W,k,s', [(2, 32, 24, 24, 3, 1), (2, 96, 24, 24, 3, 2), (5, 264, 9, 7, 3, 1), (3, 40, 13, 17, 3, 2), (2, 960, 7, 7, 3, 1)])
@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_3x3_register_tiles_opt_in(B, C, H, W, k, s, dt, monkeypatch):
    """The 3x3 instantiation of the register-tile kernels (csrc/dwconv_tile.hip) is opt-in -- it measured slower than the
    row-walk kernels on every plane of MobileNetV2 -- but stays correct: the same forward / backward checks with
    T3D_DW3_TILE_MAX covering the plane."""
    monkeypatch.setenv('T3D_DW3_TILE_MAX', '64')
    test_dwconv_fwd(B, C, H, W, k, s, dt, 'bnact')
    test_dwconv_bwd(B, C, H, W, k, s, dt, 'bnact_ps')
    test_dwconv_bwd(B, C, H, W, k, s, dt, 'plain_res')
