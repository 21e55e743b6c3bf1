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
    (2, 672, 14, 14, 5, 2)]


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
