"""GPU: the implicit-GEMM dense 3x3 convolution (csrc/conv3x3.hip: t3d_conv3x3_fwd / _dgrad / _wgrad -- nn.Conv2d(C, N, 3, s, 1)
and its autograd inside a torchvision Bottleneck, BASELINE config 4's backbone) against torch fp64 autograd on the same bf16
operands, and against the patch-matrix path it replaces (t3d_im2col + 1x1 GEMMs + t3d_col2im_bwd)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 14, 14, 64, 64, 1), (2, 15, 13, 64, 128, 2), (1, 28, 28, 128, 128, 1), (2, 8, 8, 256, 256, 2),
          (3, 7, 7, 512, 512, 1), (2, 14, 14, 256, 256, 1), (1, 56, 56, 64, 64, 1), (2, 28, 28, 128, 128, 2)]


def _bf(t):
    return t.to(torch.bfloat16)


def _inputs(B, H, W, C, N, s, seed):
    g = torch.Generator().manual_seed(seed)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = _bf(torch.randn(B, H, W, C, generator=g))
    w = torch.randn(N, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    dz = _bf(torch.randn(B, Ho, Wo, N, generator=g) * 0.1)
    al, be, ga = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.05, torch.randn(N, generator=g) * 0.01
    return x, w, sc, sh, dz, al, be, ga, Ho, Wo


def _reference(x, w, sc, sh, dz, al, be, ga, s):
    """fp64 autograd with the roundings of the bf16 path: activated operand, weights, forward output, BatchNorm-backward operand."""
    a = _bf(torch.relu(x.double() * sc.double() + sh.double())).double().permute(0, 3, 1, 2).requires_grad_(True)
    wb = _bf(w).double().requires_grad_(True)
    y = F.conv2d(a, wb, stride=s, padding=1)
    yr = _bf(y.detach().permute(0, 2, 3, 1))                                   # raw output as stored
    dy = _bf(al.double() * dz.double() + be.double() * yr.double() + ga.double()).double()
    y.backward(dy.permute(0, 3, 1, 2))
    mask = ((x.double() * sc.double() + sh.double()) > 0).double()
    dx = a.grad.permute(0, 2, 3, 1) * mask
    return y.detach().permute(0, 2, 3, 1), yr, dx, wb.grad


def _pack(w, N, C):
    from torchdet3d import _native as Nt
    kp = 9 * C
    w32 = torch.empty(N, kp, device='cuda')
    Nt.call('t3d_pack_conv_weight', Nt.F32, Nt.ptr(w), Nt.ptr(w32), N, C, 3, kp, Nt.stream())
    wb = torch.empty(N, kp, device='cuda', dtype=torch.bfloat16)
    Nt.call('t3d_pack_weight', Nt.BF16, Nt.ptr(w32), Nt.ptr(wb), N, kp, 0, Nt.stream())
    wf = torch.zeros(Nt.lib().t3d_pwconv_frag_bytes(N, kp) // 2, device='cuda', dtype=torch.bfloat16)
    Nt.call('t3d_pwconv_pack_frag', Nt.ptr(wb), Nt.ptr(wf), N, kp, Nt.stream())
    wd = torch.empty(C, 9 * N, device='cuda', dtype=torch.bfloat16)
    Nt.call('t3d_pack_conv3x3_dgrad_weight', Nt.ptr(w), Nt.ptr(wd), N, C, Nt.stream())
    wdf = torch.zeros(Nt.lib().t3d_pwconv_frag_bytes(C, 9 * N) // 2, device='cuda', dtype=torch.bfloat16)
    Nt.call('t3d_pwconv_pack_frag', Nt.ptr(wd), Nt.ptr(wdf), C, 9 * N, Nt.stream())
    return wb, wf, wdf


@pytest.mark.parametrize('B,H,W,C,N,s', SHAPES)
def test_conv3x3_implicit_gemm_against_fp64_autograd(B, H, W, C, N, s):
    from torchdet3d import _native as Nt
    x, w, sc, sh, dz, al, be, ga, Ho, Wo = _inputs(B, H, W, C, N, s, B * 100 + C + s)
    y_ref, yr, dx_ref, dw_ref = _reference(x, w, sc, sh, dz, al, be, ga, s)
    xd, wd_, scd, shd, dzd = x.cuda(), w.cuda(), sc.cuda(), sh.cuda(), dz.cuda()
    ald, bed, gad = al.cuda(), be.cuda(), ga.cuda()
    wb, wf, wdf = _pack(wd_, N, C)
    pro = Nt.prologue(scd, shd, None, 'relu', False)
    bb = Nt.bnbwd(ald, bed, gad, False)
    M2, M1 = B * Ho * Wo, B * H * W
    nrep = 4
    y = torch.full((M2, N), 7.0, device='cuda', dtype=torch.bfloat16)
    st = torch.zeros(nrep, 2 * N, device='cuda', dtype=torch.float64)
    dx = torch.full((M1, C), 7.0, device='cuda', dtype=torch.bfloat16)
    stb = torch.zeros(nrep, 2 * C, device='cuda', dtype=torch.float64)
    dwp = torch.full((N, 9 * C), 3.0, device='cuda')              # (written, not accumulated into)
    ws = torch.empty(32 << 20, device='cuda', dtype=torch.uint8)
    try:
        Nt.call('t3d_set_reduction_replicas', nrep, 2 * N)
        Nt.call('t3d_conv3x3_fwd', Nt.BF16 | Nt.W_FRAG, Nt.ptr(xd), pro, Nt.ptr(wf), Nt.ptr(y), Nt.ptr(st), B, H, W, C, N, s, Nt.stream())
        yrd = yr.cuda().reshape(M2, N).contiguous()
        Nt.call('t3d_set_reduction_replicas', nrep, 2 * C)
        Nt.call('t3d_conv3x3_dgrad', Nt.BF16 | Nt.W_FRAG, Nt.ptr(dzd), Nt.ptr(yrd), bb, Nt.ptr(wdf), Nt.ptr(xd), pro, Nt.ptr(dx),
                Nt.ptr(stb), B, H, W, C, N, s, Nt.stream())
        Nt.call('t3d_set_workspace', Nt.ptr(ws), ws.numel())
        Nt.call('t3d_conv3x3_wgrad', Nt.BF16, Nt.ptr(dzd), Nt.ptr(yrd), bb, Nt.ptr(xd), pro, Nt.ptr(dwp), B, H, W, C, N, s, Nt.stream())
    finally:
        Nt.call('t3d_set_workspace', None, 0)
        Nt.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    got = y.cpu().double().view(B, Ho, Wo, N)
    assert ((got - y_ref).abs() <= 2 ** -7 * y_ref.abs() + 2e-3).all(), (got - y_ref).abs().max().item()
    s1 = st.sum(0).cpu()
    assert torch.allclose(s1[:N], got.sum((0, 1, 2)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s1[N:], (got * got).sum((0, 1, 2)), rtol=1e-5, atol=1e-3)
    gdx = dx.cpu().double().view(B, H, W, C)
    scale = dx_ref.abs().max().item()
    assert ((gdx - dx_ref).abs() <= 2 ** -7 * dx_ref.abs() + 2e-3 * scale).all(), ((gdx - dx_ref).abs().max().item(), scale)
    s2 = stb.sum(0).cpu()
    assert torch.allclose(s2[:C], gdx.sum((0, 1, 2)), rtol=1e-4, atol=1e-3 * scale * M1 ** 0.5)
    assert torch.allclose(s2[C:], (gdx * x.double()).sum((0, 1, 2)), rtol=1e-4, atol=1e-3 * scale * M1 ** 0.5)
    gdw = dwp.cpu().double().view(N, 9, C).permute(0, 2, 1).reshape(N, C, 3, 3)
    wscale = dw_ref.abs().max().item()
    assert (gdw - dw_ref).abs().max().item() <= 2e-3 * wscale, ((gdw - dw_ref).abs().max().item(), wscale)


@pytest.mark.parametrize('B,H,W,C,N,s', [(4, 28, 28, 128, 128, 1), (4, 28, 28, 128, 128, 2)])
def test_conv3x3_implicit_gemm_equals_the_patch_matrix_path(B, H, W, C, N, s):
    """Same operands through t3d_im2col + t3d_pwconv_fwd: the same products in the same k order -> the raw outputs agree to the
    output rounding (the two GEMM kernels tile differently, so the fp32 sums may differ in the last bit)."""
    from torchdet3d import _native as Nt
    x, w, sc, sh, dz, al, be, ga, Ho, Wo = _inputs(B, H, W, C, N, s, 5)
    xd, scd, shd = x.cuda(), sc.cuda(), sh.cuda()
    wb, wf, _ = _pack(w.cuda(), N, C)
    pro = Nt.prologue(scd, shd, None, 'relu', False)
    M2, kp = B * Ho * Wo, 9 * C
    y = torch.empty(M2, N, device='cuda', dtype=torch.bfloat16)
    Nt.call('t3d_conv3x3_fwd', Nt.BF16 | Nt.W_FRAG, Nt.ptr(xd), pro, Nt.ptr(wf), Nt.ptr(y), None, B, H, W, C, N, s, Nt.stream())
    col = torch.empty(M2, kp, device='cuda', dtype=torch.bfloat16)
    Nt.call('t3d_im2col', Nt.BF16, Nt.ptr(xd), pro, Nt.ptr(col), B, H, W, C, 3, s, 1, kp, Nt.stream())
    y2 = torch.empty(M2, N, device='cuda', dtype=torch.bfloat16)
    Nt.call('t3d_pwconv_fwd', Nt.BF16, Nt.ptr(col), None, Nt.ptr(wb), None, Nt.ptr(y2), None, M2, Ho * Wo, kp, N, Nt.stream())
    torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs()
    assert (d <= 2 ** -7 * y2.float().abs() + 1e-4).all(), d.max().item()
    assert (d > 0).float().mean().item() < 0.02
