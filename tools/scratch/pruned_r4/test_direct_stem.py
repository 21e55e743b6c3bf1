"""(pruned in round 4 with the direct stem kernels; needs t3d_stem_fwd / t3d_stem_wgrad back in the library)"""
"""GPU parity of the stem's input side: the patch gather from uint8 NHWC crops (`t3d_stem_im2col_u8`, normalisation fused)
against the fp32 NCHW gather, bit for bit; the opt-in direct stem kernels (`t3d_stem_fwd` / `t3d_stem_wgrad`: patches
gathered inside the GEMM kernels, no patch matrix) against torch-CPU fp64 of nn.Conv2d(3, C, 3, 2, 1)
(models/mobilenetv3.py:110-115) on the bf16-rounded operands, in both input formats, at odd sizes and at BASELINE config
2's shape; and the uint8 input path end to end against the fp32-input path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
MEAN, STD = [0.5931, 0.4690, 0.4229], [0.2471, 0.2214, 0.2157]


def _w32(w):
    """[C,3,3,3] fp32 -> [C,32] bf16 patch-row weights (device), like the engine's packing."""
    C = w.shape[0]
    out = torch.zeros(C, 32)
    out[:, :27] = w.reshape(C, 27)
    return out.to(BF).cuda()


@pytest.mark.parametrize('dtype', [torch.float32, BF])
@pytest.mark.parametrize('B,H,W', [(2, 33, 47), (1, 8, 8), (32, 224, 224)])
@pytest.mark.parametrize('B,H,W,C,fmt', [(3, 33, 47, 32, 0), (2, 96, 96, 16, 0), (2, 50, 31, 32, 1), (256, 224, 224, 32, 0),
                                         (64, 224, 224, 32, 1)])
def test_direct_stem_forward_and_weight_gradient(B, H, W, C, fmt):
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(B + H + C + fmt)
    w = torch.randn(C, 3, 3, 3, generator=g) * 0.2
    mean, istd = torch.tensor(MEAN), 1.0 / torch.tensor(STD)
    if fmt == 0:
        x = torch.randn(B, 3, H, W, generator=g)
        xd = x.cuda()
        xn = x
    else:
        u = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
        xd = u.cuda()
        xn = ((u.float() * (1.0 / 255.0) - mean) * istd).permute(0, 3, 1, 2).contiguous()
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    M = B * Ho * Wo
    wd, md, sd = _w32(w), mean.cuda(), istd.cuda()
    y = torch.empty(M, C, device='cuda', dtype=BF)
    stats = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
    N.call('t3d_stem_fwd', N.BF16, N.ptr(xd), fmt, N.ptr(md), N.ptr(sd), N.ptr(wd), N.ptr(y), N.ptr(stats), B, H, W, C, N.stream())
    dz, yb = torch.randn(M, C, generator=g).to(BF), torch.randn(M, C, generator=g).to(BF)
    alpha, beta, gamma = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2, torch.randn(C, generator=g) * 0.1
    keep = [t.cuda() for t in (alpha, beta, gamma, dz, yb)]
    bb = N.bnbwd(keep[0], keep[1], keep[2], False)
    dw = torch.zeros(C, 32, device='cuda')
    ws = torch.empty(16 << 20, device='cuda', dtype=torch.uint8)
    N.call('t3d_set_workspace', N.ptr(ws), ws.numel())
    try:
        N.call('t3d_stem_wgrad', N.BF16, N.ptr(keep[3]), N.ptr(keep[4]), bb, N.ptr(xd), fmt, N.ptr(md), N.ptr(sd), N.ptr(dw),
               B, H, W, C, N.stream())
    finally:
        N.call('t3d_set_workspace', None, 0)
    torch.cuda.synchronize()
    # reference on the operands the MFMA sees: patches and weights rounded to bf16
    sub = slice(0, min(B, 8))                                   # forward values: a slice of the batch is enough
    xq = xn.to(BF).double()
    wq = wd.double().cpu()[:, :27].reshape(C, 3, 3, 3)
    ref = F.conv2d(xq[sub], wq, None, 2, 1).permute(0, 2, 3, 1).reshape(-1, C)
    got = y.double().cpu()[:ref.shape[0]]
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=8e-3, atol=8e-3 * ref.abs().max().item())
    yd = y.double()
    st = stats.cpu().view(2, C)
    np.testing.assert_allclose(st[0].numpy(), yd.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
    np.testing.assert_allclose(st[1].numpy(), (yd * yd).sum(0).cpu().numpy(), rtol=1e-5)
    # weight gradient over the whole batch: dW[c][tap] = sum_pixels dy[p][c] * patch[p][tap]
    dy = (alpha.double() * dz.double() + beta.double() * yb.double() + gamma.double()).to(BF).double()
    cols = F.unfold(xq, 3, padding=1, stride=2)                 # [B, 27, Ho*Wo]
    ref_dw = torch.einsum('bkp,bpc->ck', cols, dy.view(B, Ho * Wo, C))
    gdw = dw.double().cpu()
    np.testing.assert_allclose(gdw[:, :27].numpy(), ref_dw.numpy(), rtol=3e-3, atol=3e-4 * ref_dw.abs().max().item())
    assert gdw[:, 27:].abs().max().item() == 0


