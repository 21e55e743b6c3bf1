"""GPU: the fused expand 1x1 + BatchNorm + activation + depthwise 3x3 forward (`t3d_expdw_fwd`, csrc/expdw_fwd.hip)
against torch-CPU fp64 on the same bf16 operands: the depthwise output, the optional raw expansion and the BatchNorm
sums, at small shapes (every stride / tile edge case) and at production shapes of MobileNetV2 @224 (sampled)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(z, w1, sc, sh, act, wdw, s, round_y1):
    """z [B,H,W,K] bf16, w1 [C,K] bf16 -> (y1 [B,H,W,C] fp64, y2 [B,Ho,Wo,C] fp64)."""
    y1 = torch.einsum('bhwk,ck->bhwc', z.double(), w1.double())
    y1n = y1.to(torch.bfloat16).double() if round_y1 else y1
    u = y1n * sc.double() + sh.double()
    a = {1: lambda t: t.clamp(min=0), 2: lambda t: t.clamp(0, 6)}[act](u)
    a = a.to(torch.bfloat16).double()                          # the LDS tile holds bf16
    y2 = F.conv2d(a.permute(0, 3, 1, 2), wdw.double().view(-1, 1, 3, 3), stride=s, padding=1, groups=a.shape[3])
    return y1, y2.permute(0, 2, 3, 1)


@pytest.mark.parametrize('B,H,W,K,C,s,act,store', [
    (2, 14, 14, 64, 384, 1, 2, True), (3, 7, 7, 160, 960, 1, 2, False), (2, 28, 28, 32, 192, 2, 2, True),
    (2, 23, 19, 24, 144, 1, 1, True), (2, 23, 19, 24, 144, 2, 2, False), (1, 56, 56, 24, 144, 1, 2, True),
    (2, 112, 112, 16, 96, 2, 2, True), (5, 9, 33, 8, 40, 2, 1, False)])
def test_expdw_fwd_matches_torch(B, H, W, K, C, s, act, store):
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(B * 1000 + H + C)
    z = torch.randn(B, H, W, K, generator=g).to(torch.bfloat16)
    w1 = (torch.randn(C, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
    wdw = torch.randn(C, 9, generator=g) * 0.3
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    zd, w1d, scd, shd, wd = z.cuda(), w1.cuda(), sc.cuda(), sh.cuda(), wdw.cuda()
    y1 = torch.full((B, H, W, C), 7.0, dtype=torch.bfloat16, device='cuda') if store else None
    y2 = torch.full((B, Ho, Wo, C), 7.0, dtype=torch.bfloat16, device='cuda')
    stats = torch.zeros(4, 2 * C, dtype=torch.float64, device='cuda')
    N.call('t3d_set_reduction_replicas', 4, 2 * C)
    try:
        N.call('t3d_expdw_fwd', N.ptr(zd), N.ptr(w1d), N.ptr(scd), N.ptr(shd), act, N.ptr(wd), N.ptr(y1), N.ptr(y2),
               N.ptr(stats), B, H, W, K, C, s, N.stream())
    finally:
        N.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    r1, r2 = _ref(z, w1, sc, sh, act, wdw, s, store)
    if store:
        assert torch.equal(y1.cpu(), r1.to(torch.bfloat16)) or (y1.cpu().double() - r1).abs().max() <= 2 ** -7 * r1.abs().max()
    got = y2.cpu().double()
    # one bf16 rounding of the output (2^-8 relative) + fp32 accumulation order + the occasional activation whose bf16
    # rounding in the LDS tile falls the other way than the fp64 reference's (one ulp of a value up to 6, times a weight)
    tol = 0.03 + 2 ** -7 * r2.abs()
    assert ((got - r2).abs() <= tol).all(), (got - r2).abs().max().item()
    st = stats.sum(0).cpu()
    assert torch.allclose(st[:C], got.sum((0, 1, 2)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(st[C:], (got * got).sum((0, 1, 2)), rtol=1e-5, atol=1e-3)


def test_expdw_fwd_production_shape_sampled():
    """B = 256, 112x112x16 -> 96, stride 2 (MobileNetV2 features.2 @224): 6 images checked in full."""
    from torchdet3d import _native as N
    B, H, W, K, C, s, act = 256, 112, 112, 16, 96, 2, 2
    g = torch.Generator().manual_seed(3)
    zs = torch.randn(6, H, W, K, generator=g).to(torch.bfloat16)
    z = zs.cuda().repeat(43, 1, 1, 1)[:B].contiguous()
    w1 = (torch.randn(C, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    sc, sh, wdw = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5, torch.randn(C, 9, generator=g) * 0.3
    y2 = torch.empty(B, 56, 56, C, dtype=torch.bfloat16, device='cuda')
    w1d, scd, shd, wd = w1.cuda(), sc.cuda(), sh.cuda(), wdw.cuda()
    N.call('t3d_expdw_fwd', N.ptr(z), N.ptr(w1d), N.ptr(scd), N.ptr(shd), act, N.ptr(wd), None,
           N.ptr(y2), None, B, H, W, K, C, s, N.stream())
    torch.cuda.synchronize()
    _, r2 = _ref(zs, w1, sc, sh, act, wdw, s, False)
    for b in (0, 5, 6 * 42 + 1, 255):
        got = y2[b].cpu().double()
        ref = r2[b % 6]
        assert ((got - ref).abs() <= 0.03 + 2 ** -7 * ref.abs()).all()
