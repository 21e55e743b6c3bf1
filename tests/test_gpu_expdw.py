"""GPU: the fused expand 1x1 + BatchNorm + activation + depthwise 3x3 forward (`t3d_expdw_fwd`, csrc/expdw_fwd.hip;
models/mobilenetv3.py:146-153 of the reference) against torch-CPU fp64 on the same bf16 operands and against the two launches
it replaces: the depthwise output, the optional raw expansion and the BatchNorm sums, at small shapes (every stride / tile
edge case) and at production shapes of MobileNetV2 @224 (sampled)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(z, w1, sc, sh, act, wdw, s):
    """z [B,H,W,K] bf16, w1 [C,K] bf16 -> (y1 [B,H,W,C] fp64, y2 [B,Ho,Wo,C] fp64); the activation sees the bf16-rounded y1."""
    y1 = torch.einsum('bhwk,ck->bhwc', z.double(), w1.double())
    u = y1.to(torch.bfloat16).double() * sc.double() + sh.double()
    a = {1: lambda t: t.clamp(min=0), 2: lambda t: t.clamp(0, 6)}[act](u)
    y2 = F.conv2d(a.permute(0, 3, 1, 2), wdw.double().view(-1, 1, 3, 3), stride=s, padding=1, groups=a.shape[3])
    return y1, y2.permute(0, 2, 3, 1)


def _run(z, w1, sc, sh, act, wdw, s, store, nrep=4):
    from torchdet3d import _native as N
    B, H, W, K = z.shape
    C = w1.shape[0]
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    y1 = torch.full((B, H, W, C), 7.0, dtype=torch.bfloat16, device='cuda') if store else None
    y2 = torch.full((B, Ho, Wo, C), 7.0, dtype=torch.bfloat16, device='cuda')
    stats = torch.zeros(nrep, 2 * C, dtype=torch.float64, device='cuda')
    N.call('t3d_set_reduction_replicas', nrep, 2 * C)
    try:
        N.call('t3d_expdw_fwd', N.BF16, N.ptr(z), N.ptr(w1), N.ptr(sc), N.ptr(sh), act, N.ptr(wdw), N.ptr(y1), N.ptr(y2),
               N.ptr(stats), B, H, W, K, C, s, N.stream())
    finally:
        N.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    return y1, y2, stats.sum(0)


@pytest.mark.parametrize('B,H,W,K,C,s,act,store', [
    (2, 28, 28, 32, 192, 2, 2, True), (2, 28, 28, 32, 192, 1, 2, True), (2, 23, 19, 24, 144, 1, 1, True),
    (2, 23, 19, 24, 144, 2, 2, False), (3, 56, 56, 24, 144, 1, 2, True), (3, 56, 56, 24, 144, 2, 2, True),
    (2, 112, 112, 16, 96, 2, 2, True), (5, 9, 33, 8, 40, 2, 1, False), (9, 14, 14, 32, 64, 1, 2, True), (1, 8, 8, 16, 32, 1, 2, True)])
def test_expdw_fwd_matches_torch(B, H, W, K, C, s, act, store):
    g = torch.Generator().manual_seed(B * 1000 + H + C)
    z = torch.randn(B, H, W, K, generator=g).to(torch.bfloat16)
    w1 = (torch.randn(C, K, generator=g) / K ** 0.5).to(torch.bfloat16)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
    wdw = torch.randn(C, 9, generator=g) * 0.3
    y1, y2, st = _run(z.cuda(), w1.cuda(), sc.cuda(), sh.cuda(), act, wdw.cuda(), s, store)
    r1, r2 = _ref(z, w1, sc, sh, act, wdw, s)
    if store:
        # the MFMA's fp32 sum against fp64: equal after rounding except where the sum sits on a rounding boundary
        d = (y1.cpu().double() - r1).abs()
        assert (d <= 2 ** -8 * r1.abs() + 1e-6).all(), d.max().item()
    got = y2.cpu().double()
    # one bf16 rounding of the output (2^-8 relative), fp32 accumulation order, and the occasional y1 whose bf16 rounding falls the
    # other way than the fp64 reference's (one bf16 ulp of y1 through the BatchNorm scale and a stencil weight)
    tol = 0.03 + 2 ** -7 * r2.abs()
    assert ((got - r2).abs() <= tol).all(), (got - r2).abs().max().item()
    assert torch.allclose(st[:C].cpu(), got.sum((0, 1, 2)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(st[C:].cpu(), (got * got).sum((0, 1, 2)), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize('H,K,C,s', [(112, 16, 96, 2), (56, 24, 144, 1), (56, 24, 144, 2), (28, 32, 192, 1), (28, 32, 192, 2)])
def test_expdw_fwd_against_the_two_launches_at_production_shapes(H, K, C, s):
    """B = 256 at the block shapes of MobileNetV2 @224: same y1 bit for bit (the same MFMA on the same operands), y2 within the
    output rounding of the depthwise kernel it replaces, identical-to-1e-6 BatchNorm sums; and run twice: bit-identical."""
    from torchdet3d import _native as N
    B = 256
    g = torch.Generator(device='cuda').manual_seed(H + C)
    z = torch.randn(B, H, H, K, device='cuda', generator=g).to(torch.bfloat16)
    w1 = (torch.randn(C, K, device='cuda', generator=g) / K ** 0.5).to(torch.bfloat16)
    sc, sh = torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.5
    wdw = torch.randn(C, 9, device='cuda', generator=g) * 0.3
    y1, y2, st = _run(z, w1, sc, sh, 2, wdw, s, True, nrep=16)
    y1b, y2b, stb = _run(z, w1, sc, sh, 2, wdw, s, True, nrep=16)
    assert torch.equal(y1, y1b) and torch.equal(y2, y2b) and torch.equal(st, stb)
    Ho = (H - 1) // s + 1
    M = B * H * H
    p1 = torch.empty(B, H, H, C, device='cuda', dtype=torch.bfloat16)
    p2 = torch.empty(B, Ho, Ho, C, device='cuda', dtype=torch.bfloat16)
    stats = torch.zeros(16, 2 * C, device='cuda', dtype=torch.float64)
    N.call('t3d_set_reduction_replicas', 16, 2 * C)
    try:
        N.call('t3d_pwconv_fwd', N.BF16, N.ptr(z), None, N.ptr(w1), None, N.ptr(p1), None, M, H * H, K, C, N.stream())
        N.call('t3d_dwconv_fwd', N.BF16, N.ptr(p1), N.prologue(sc, sh, None, 'relu6', False), N.ptr(wdw), N.ptr(p2), N.ptr(stats),
               None, B, H, H, C, 3, s, N.stream())
    finally:
        N.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    assert torch.equal(y1, p1)
    d = (y2.float() - p2.float()).abs()
    assert (d <= 2 ** -7 * p2.float().abs() + 1e-3).all(), d.max().item()
    assert (d > 0).float().mean().item() < 0.05          # a different summation order flips few roundings
    sp = stats.sum(0)
    assert torch.allclose(st, sp, rtol=2e-4, atol=1.0)


def test_wide_inputs_take_the_two_launches_instead_of_failing():
    """the kernel refuses input rows wider than its fragment registers / LDS rows hold (W > 213: MobileNetV2's
    second block on 448 ... 512-pixel crops).  `t3d_expdw_supported` answers exactly what `t3d_expdw_fwd` would, the engine asks
    it, and a 16-bit inference forward at 448 px runs -- on the two-launch path for that block -- and agrees with the fp32
    engine like the 224-px forward does."""
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d import _native as N
    from torchdet3d.models.engine import Net
    sup = N.lib().t3d_expdw_supported
    assert sup(N.BF16, N.ACT['relu6'], 2, 112, 112, 16, 96, 2) == 1
    assert sup(N.BF16, N.ACT['relu6'], 2, 224, 213, 16, 96, 2) == 1
    assert sup(N.BF16, N.ACT['relu6'], 2, 224, 224, 16, 96, 2) == 0          # too wide
    assert sup(N.BF16, N.ACT['relu6'], 2, 56, 56, 40, 240, 1) == 0            # K > 32
    assert sup(N.F32, N.ACT['relu6'], 2, 56, 56, 24, 144, 1) == 0
    assert sup(N.BF16, N.ACT['hswish'], 2, 56, 56, 24, 144, 1) == 0
    # the query and the launch agree on both sides of the width limit
    for W in (213, 214):
        z = torch.randn(1, 8, W, 16, device='cuda').to(torch.bfloat16)
        w1 = torch.randn(96, 16, device='cuda').to(torch.bfloat16)
        sc, sh, wdw = torch.ones(96, device='cuda'), torch.zeros(96, device='cuda'), torch.randn(96, 9, device='cuda')
        y2 = torch.empty(1, 4, (W - 1) // 2 + 1, 96, device='cuda', dtype=torch.bfloat16)
        rc = N.lib().t3d_expdw_fwd(N.BF16, N.ptr(z), N.ptr(w1), N.ptr(sc), N.ptr(sh), N.ACT['relu6'], N.ptr(wdw), None, N.ptr(y2),
                                   None, 1, 8, W, 16, 96, 2, N.stream())
        torch.cuda.synchronize()
        assert (rc == 0) == bool(sup(N.BF16, N.ACT['relu6'], 1, 8, W, 16, 96, 2)), (W, rc)
    B, HW = 4, 448
    imgs, _, cats = make_inputs(B, HW, HW, 9)
    sd = make_state_dict('mobilenetv2', 9)
    outs = {}
    for dt in (torch.bfloat16, torch.float32):
        net = Net('mobilenetv2', 9, 'cuda', dt)
        net.load_state_dict(sd)
        with torch.no_grad():
            kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=False)
        outs[dt] = (kp.float().cpu(), lg.float().cpu())
        del net
    assert (outs[torch.bfloat16][0] - outs[torch.float32][0]).abs().max().item() < 5e-3
    assert torch.isfinite(outs[torch.bfloat16][1]).all()


@pytest.mark.parametrize('K,C,M,res', [(16, 96, 40000, False), (16, 96, 5001, True), (8, 48, 3000, True)])
def test_gram_statistics_of_the_expansion_match_batchnorm_of_the_product(K, C, M, res):
    """`t3d_bn_apply_gram` + `t3d_gram_bn_finalize` (csrc/gram.hip): the block input is materialised exactly as `t3d_bn_apply` does
    it, and the BatchNorm coefficients of W1 z derived from the K x K Gram sums equal what `F.batch_norm` computes on the product
    itself (fp64 reference on the same bf16 operands), running statistics included."""
    from torchdet3d import _native as N
    g = torch.Generator(device='cuda').manual_seed(K + C + M)
    y = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16)
    sc, sh = torch.rand(K, device='cuda', generator=g) + 0.5, torch.randn(K, device='cuda', generator=g) * 0.3
    r = torch.randn(M, K, device='cuda', generator=g).to(torch.bfloat16) if res else None
    w = (torch.randn(C, K, device='cuda', generator=g) / K ** .5).to(torch.bfloat16)
    gamma, beta = torch.rand(C, device='cuda', generator=g) + 0.5, torch.randn(C, device='cuda', generator=g) * 0.1
    rm, rv = torch.randn(C, device='cuda', generator=g) * 0.1, torch.rand(C, device='cuda', generator=g) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    nbt = torch.zeros(1, dtype=torch.int64, device='cuda')
    z = torch.empty(M, K, device='cuda', dtype=torch.bfloat16)
    z2 = torch.empty_like(z)
    gram = torch.zeros(16, K * (K + 1) // 2 + K, device='cuda', dtype=torch.float64)      # 16 reduction replicas
    pro = N.prologue(sc, sh, None, 'none', False)
    N.call('t3d_bn_apply_gram', N.BF16, N.ptr(y), pro, N.ptr(r), N.ptr(z), N.ptr(gram), M, K, N.stream())
    N.call('t3d_bn_apply', N.BF16, N.ptr(y), pro, N.ptr(r), N.ptr(z2), M, K, N.stream())
    scale, shift, mean, invstd = (torch.empty(C, device='cuda') for _ in range(4))
    N.call('t3d_gram_bn_finalize', N.ptr(gram), N.ptr(w), C, K, float(M), N.ptr(gamma), N.ptr(beta), N.ptr(rm), N.ptr(rv), N.ptr(nbt),
           0.1, 1e-5, N.ptr(scale), N.ptr(shift), N.ptr(mean), N.ptr(invstd), N.stream())
    torch.cuda.synchronize()
    assert torch.equal(z, z2)                                          # the same materialised tensor, bit for bit
    zd = z.double().cpu()
    iu = torch.triu_indices(K, K)
    G = (zd.t() @ zd)[iu[0], iu[1]]
    np_ = gram.sum(0).cpu()
    assert torch.allclose(np_[:G.numel()], G, rtol=1e-6, atol=1e-6 * M)
    assert torch.allclose(np_[G.numel():], zd.sum(0), rtol=1e-6, atol=1e-6 * M)
    y1 = zd @ w.double().cpu().t()                                     # the exact products the MFMA accumulates
    m_ref, v_ref = y1.mean(0), y1.var(0, unbiased=False)
    assert torch.allclose(mean.double().cpu(), m_ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(invstd.double().cpu(), 1 / torch.sqrt(v_ref + 1e-5), rtol=1e-5)
    assert torch.allclose(scale.double().cpu(), gamma.double().cpu() / torch.sqrt(v_ref + 1e-5), rtol=1e-5)
    assert torch.allclose(shift.double().cpu(), beta.double().cpu() - m_ref * gamma.double().cpu() / torch.sqrt(v_ref + 1e-5), rtol=1e-4, atol=1e-5)
    assert torch.allclose(rm.double().cpu(), 0.9 * rm0.double().cpu() + 0.1 * m_ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv.double().cpu(), 0.9 * rv0.double().cpu() + 0.1 * v_ref * M / (M - 1), rtol=1e-5)
    assert int(nbt) == 1
    # an already finished tensor: statistics only, nothing written
    gram2 = torch.zeros_like(gram)
    N.call('t3d_bn_apply_gram', N.BF16, N.ptr(z), None, None, None, N.ptr(gram2), M, K, N.stream())
    torch.cuda.synchronize()
    assert torch.equal(gram2, gram)                                    # snapped sums: the same bits in any order, replica by replica


def test_training_step_with_gram_statistics_matches_the_two_launch_path(monkeypatch):
    """Engine level (DESIGN.md finding 55): MobileNetV2's first expanded block through [materialise + Gram] -> finalize ->
    fused expand + depthwise with the expansion stored, against the two launches with the statistics from the conv's epilogue:
    the same train step to bf16 rounding (the Gram statistics are those of the exact products, the epilogue's those of their
    bf16 rounding), bit-reproducible run to run, the 1x1 + depthwise launches of that block gone."""
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d import _native as N
    from torchdet3d.models import engine as E
    B, HW = 16, 96
    imgs, gt_kp, cats = make_inputs(B, HW, HW, 9)
    sd = make_state_dict('mobilenetv2', 9)
    ones = torch.ones(B, 1280, device='cuda')

    def run(min_hw):
        monkeypatch.setattr(E, 'GRAM_FWD_MIN_HW', min_hw)
        net = E.Net('mobilenetv2', 9, 'cuda', torch.bfloat16)
        net.load_state_dict(sd)
        calls = []
        real = N.call
        monkeypatch.setattr(N, 'call', lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1])
        kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=ones)
        dkp, dlg = torch.randn(B, 18, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1)) * 1e-2, torch.zeros(B, 9, device='cuda')
        net.backward(dkp, dlg)
        torch.cuda.synchronize()
        monkeypatch.setattr(N, 'call', real)
        bn1 = net.bns['features.2.conv.1']
        y1b = next(v for k, v in net._bufs.items() if k[0] == 'y1:1').float().cpu().clone()
        y2b = next(v for k, v in net._bufs.items() if k[0] == 'y2:1').float().cpu().clone()
        out = (kp.float().cpu(), {k: v.float().cpu().clone() for k, v in net.g.items()}, {k: v.float().cpu().clone() for k, v in net.buffers.items()}, calls,
               (bn1.scale.cpu().clone(), bn1.shift.cpu().clone(), bn1.mean.cpu().clone(), bn1.invstd.cpu().clone()), y1b, y2b)
        del net
        return out

    kp0, g0, b0, c0, co0, y1_0, y2_0 = run(0)
    kp1, g1, b1, c1, co1, y1_1, y2_1 = run(1)
    kp2, g2, b2, c2, _, _, _ = run(1)
    # the block itself: the raw expansion is the same bits (same MFMA, same operands), the depthwise output the same to its rounding
    assert torch.equal(y1_0, y1_1)
    assert ((y2_1 - y2_0).norm() / y2_0.norm()).item() < 5e-3
    # the expansion's BatchNorm coefficients themselves: Gram-derived against epilogue sums of the rounded products
    for u, v in zip(co0, co1):
        assert torch.allclose(u, v, rtol=2e-3, atol=2e-4), (u - v).abs().max()
    assert 't3d_bn_apply_gram' not in c0 and c1.count('t3d_bn_apply_gram') >= 1 and c1.count('t3d_expdw_fwd') == c1.count('t3d_bn_apply_gram')
    assert c1.count('t3d_dwconv_fwd') == c0.count('t3d_dwconv_fwd') - c1.count('t3d_expdw_fwd')
    assert torch.equal(kp1, kp2) and all(torch.equal(g1[k], g2[k]) for k in g1)            # run-to-run bit identity
    print('[gram path] max keypoint difference against the two-launch path:', (kp1 - kp0).abs().max().item())
    assert (kp1 - kp0).abs().max().item() < 6e-2           # (a randomly initialised bf16 network amplifies a 1e-6 change of one BatchNorm: section 2)
    for k in ('features.2.conv.1.running_mean', 'features.2.conv.1.running_var'):
        assert torch.allclose(b1[k], b0[k], rtol=2e-3, atol=2e-4), k
    # (whole-network gradients of two bf16 steps that round at different points are not comparable at random initialisation --
    #  the difference of one rounding grows ~9 % per layer -- so the block-level identities above are the test)
