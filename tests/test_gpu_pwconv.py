"""GPU parity: pointwise-conv GEMM (forward + data gradient) through the C ABI vs torch-CPU math."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dt(dt):
    return torch.float32 if dt == 'f32' else torch.bfloat16


def _q(x, dtype):
    return x.to(dtype).float()


def _act(x, kind):
    from oracle.model import act_fn
    return act_fn(x, kind)


def _act_grad(u, kind):
    u = u.clone().requires_grad_(True)
    _act(u, kind).sum().backward()
    return u.grad


def _pack(w, dtype, transpose=False):
    from torchdet3d import _native as N
    wd = w.contiguous().cuda()
    r, c = w.shape
    out = torch.empty((c, r) if transpose else (r, c), device='cuda', dtype=dtype)
    N.call('t3d_pack_weight', N.dtype_code(out), N.ptr(wd), N.ptr(out), r, c, int(transpose), N.stream())
    return out


def _takes_frag(K, N, plain=True):
    """Shapes for which the 16-bit pointwise entry points accept T3D_W_FRAG (include/t3d.h): the streaming kernel's (its LDS chunk
    holds >= 2 tiles of every k-step) and, for launches without bias / gates / per-sample coefficients, the deep-contraction
    kernel's."""
    from torchdet3d import _native as Nt
    return K % 8 == 0 and N % 8 == 0 and ((K + 31) // 32 <= 60 or (plain and bool(Nt.lib().t3d_pwconv_wants_frag(K, N))))


def _frag(wd):
    """Fragment-order copy of a packed bf16 weight matrix (t3d_pwconv_pack_frag)."""
    from torchdet3d import _native as N
    r, c = wd.shape
    out = torch.zeros(N.lib().t3d_pwconv_frag_bytes(r, c) // 2, device='cuda', dtype=torch.bfloat16)
    N.call('t3d_pwconv_pack_frag', N.ptr(wd), N.ptr(out), r, c, N.stream())
    return out


def test_fragment_order_weight_layout():
    """t3d_pwconv_pack_frag and the frag / frag_t outputs of t3d_pack_weights_batched against the definition in include/t3d.h,
    ragged rows and columns included; the flag is refused where the deep-contraction kernel does not apply."""
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(5)
    for r, c in [(160, 960), (40, 72), (520, 1032)]:
        w = torch.randn(r, c, generator=g)
        wq = w.to('cuda', torch.bfloat16)
        KS, NP = (c + 31) // 32, (r + 31) // 32
        wp = torch.zeros(NP * 32, KS * 32, dtype=torch.bfloat16)
        wp[:r, :c] = wq.cpu()
        # row(T, lc) = 32 (T >> 1) + 8 (lc >> 2) + 4 (T & 1) + (lc & 3): [pair][lc >> 2][T & 1][lc & 3] -> [pair][T & 1][lc >> 2][lc & 3]
        rows = wp.view(NP, 4, 2, 4, KS * 32).permute(0, 2, 1, 3, 4).reshape(NP * 2, 16, KS, 4, 8)      # [T][lc][ks][lg][j]
        ref = rows.permute(0, 2, 3, 1, 4).contiguous().view(-1)                                         # [T][ks][lg][lc][j]
        got = _frag(wq)
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), ref)
        # batched form: plain, transposed and both fragment-order copies in one launch
        wf = w.cuda()
        out, out_t = torch.empty(r, c, device='cuda', dtype=torch.bfloat16), torch.empty(c, r, device='cuda', dtype=torch.bfloat16)
        fr = torch.zeros_like(got)
        frt = torch.zeros(N.lib().t3d_pwconv_frag_bytes(c, r) // 2, device='cuda', dtype=torch.bfloat16)
        desc = torch.tensor([[wf.data_ptr(), out.data_ptr(), out_t.data_ptr(), r, c, fr.data_ptr(), frt.data_ptr()]],
                            dtype=torch.int64, device='cuda')
        N.call('t3d_pack_weights_batched', N.BF16, N.ptr(desc), 1, N.stream())
        torch.cuda.synchronize()
        assert torch.equal(out, wq) and torch.equal(out_t, wq.t().contiguous())
        assert torch.equal(fr.cpu(), ref) and torch.equal(frt, _frag(out_t))
    x = torch.zeros(64, 96, device='cuda', dtype=torch.bfloat16)
    y = torch.zeros(64, 24, device='cuda', dtype=torch.bfloat16)
    assert N.lib().t3d_pwconv_wants_frag(96, 24) == 0 and N.lib().t3d_pwconv_wants_frag(960, 160) == 1
    assert N.lib().t3d_pwconv_wants_frag(2048, 8) == 1 and N.lib().t3d_pwconv_wants_frag(960, 64) == 0
    rc = N.lib().t3d_pwconv_fwd(N.F32 | N.W_FRAG, N.ptr(x), None, N.ptr(y), None, N.ptr(y), None, 64, 64, 96, 24, N.stream())
    assert rc == -1          # T3D_ERR_ARG: 16-bit storage only


SHAPES = [  # B, HW, K, N
    (2, 36, 16, 96), (3, 49, 96, 24), (2, 100, 24, 144), (5, 9, 160, 960), (2, 144, 672, 112),
    (7, 1, 960, 1280), (2, 333, 72, 40), (1, 130, 200, 80), (2, 64, 320, 1280), (2, 200, 8, 8)]


# deep contractions with outputs wider than the streaming kernel's LDS weight chunk: pwconv_deep.hip (operand resident in LDS,
# weights streamed), incl. ragged K (not a multiple of 32), a ragged last 32-channel block and several output chunks
DEEP_FWD = [(8, 49, 960, 160), (3, 49, 960, 320), (5, 49, 576, 160), (2, 37, 1032, 520), (1, 70, 2048, 512)]
DEEP_BWD = [(8, 49, 160, 960), (3, 49, 320, 1280), (2, 37, 520, 1032), (1, 70, 512, 2048)]


@pytest.mark.parametrize('B,HW,K,N', SHAPES + DEEP_FWD)
@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('mode', ['plain', 'bnact', 'se_pre', 'se_post'])
def test_pwconv_fwd(B, HW, K, N, dt, mode):
    from torchdet3d import _native as Nt
    dtype = _dt(dt)
    g = torch.Generator().manual_seed(B + HW + K + N)
    M = B * HW
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) if mode == 'plain' else None
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    se = torch.rand(B, K, generator=g)
    act = {'plain': 'none', 'bnact': 'relu6', 'se_pre': 'hswish', 'se_post': 'relu'}[mode]
    xq = _q(x, dtype)
    sev = se.repeat_interleave(HW, 0)
    if mode == 'plain':
        a = xq
    else:
        u = xq * scale + shift
        a = _act(u * sev, act) if mode == 'se_pre' else (_act(u, act) * sev if mode == 'se_post' else _act(u, act))
    a = _q(a, dtype)
    ref = a.double() @ _q(w, dtype).double().t()
    if bias is not None:
        ref = ref + bias.double()
    xd = x.to('cuda', dtype)
    wd = _pack(w, dtype)
    y = torch.empty(M, N, device='cuda', dtype=dtype)
    stats = torch.zeros(2 * N, device='cuda', dtype=torch.float64)
    keep = [t.cuda() for t in (scale, shift, se)]
    bd = bias.cuda() if bias is not None else None
    p = None if mode == 'plain' else Nt.prologue(keep[0], keep[1], keep[2] if mode.startswith('se') else None, act,
                                                 mode == 'se_post')
    # second pass: the same call with the fragment-order copy of the weights and T3D_W_FRAG (streaming kernel: linear weight
    # staging; deep-contraction kernel, csrc/pwconv_deep.hip, for its shapes when there are no gates)
    variants = [(Nt.dtype_code(xd), wd)]
    if dt == 'bf16' and _takes_frag(K, N, mode == 'bnact'):
        variants.append((Nt.BF16 | Nt.W_FRAG, _frag(wd)))
    for code, wv in variants:
        y.zero_(); stats.zero_()
        Nt.call('t3d_pwconv_fwd', code, Nt.ptr(xd), p, Nt.ptr(wv), Nt.ptr(bd), Nt.ptr(y), Nt.ptr(stats),
                M, HW, K, N, Nt.stream())
        torch.cuda.synchronize()
        got = y.float().cpu()
        tol = 1e-5 if dt == 'f32' else 1e-2
        np.testing.assert_allclose(got.numpy(), ref.float().numpy(), atol=tol * max(1., ref.abs().max().item()), rtol=tol)
        st = stats.cpu().view(2, N)
        np.testing.assert_allclose(st[0].numpy(), got.double().sum(0).numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
        np.testing.assert_allclose(st[1].numpy(), (got.double() ** 2).sum(0).numpy(), rtol=1e-5, atol=1e-4)


# fp32 storage, inference forward (no statistics) of layers with >= 1024 pixels: the fp32 streaming kernel
# (csrc/pwconv_f32_stream.hip) -- MobileNetV2's layer shapes, ragged contraction (24, 40, 72: not multiples of 16), ragged pixel
# counts, several output chunks (960 -> 160: two tiles per chunk), bias; 1e-5 against fp64 like the tiled kernel it replaces
@pytest.mark.parametrize('M,K,N', [(4096, 16, 96), (3001, 24, 144), (2048, 144, 24), (1500, 32, 192), (1024, 192, 64),
                                   (1031, 384, 96), (1200, 576, 160), (1024, 960, 160), (1024, 960, 320), (1100, 160, 960),
                                   (1024, 320, 1280), (5000, 72, 40), (1111, 40, 240), (2000, 8, 8), (1024, 1280, 8)])
@pytest.mark.parametrize('mode', ['plain', 'bias', 'relu6', 'hswish'])
def test_pwconv_fwd_f32_inference_streaming_kernel(M, K, N, mode):
    from torchdet3d import _native as Nt
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) if mode == 'bias' else None
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    a = x if mode in ('plain', 'bias') else _act(x * scale + shift, mode)
    ref = a.double() @ w.double().t()
    if bias is not None:
        ref = ref + bias.double()
    xd, wd = x.cuda(), w.cuda()
    keep = [scale.cuda(), shift.cuda()]
    bd = bias.cuda() if bias is not None else None
    p = None if mode in ('plain', 'bias') else Nt.prologue(keep[0], keep[1], None, mode, False)
    y = torch.full((M + 1, N), 7.0, device='cuda')          # (one guard row past the output)
    n0 = Nt.launch_count()
    Nt.call('t3d_pwconv_fwd', Nt.F32, Nt.ptr(xd), p, Nt.ptr(wd), Nt.ptr(bd), Nt.ptr(y), None, M, 1, K, N, Nt.stream())
    torch.cuda.synchronize()
    assert Nt.launch_count() - n0 == 1
    assert torch.all(y[M] == 7.0)
    np.testing.assert_allclose(y[:M].cpu().numpy(), ref.float().numpy(), atol=1e-5 * max(1., ref.abs().max().item()), rtol=1e-5)


# fp32 storage, TRAINING forward (BatchNorm sums) of layers with >= 1024 pixels: the register-operand kernel's persistent variant
# -- output as in inference, the sums = column sums of the stored output (fp32 partials over a wave's 16 R pixels, added in fp64:
# exact adds, so the order does not matter), spread over the reduction replicas, bit-identical from run to run
@pytest.mark.parametrize('M,K,N', [(4096, 16, 96), (3001, 24, 144), (2048, 144, 24), (1031, 384, 96), (1024, 960, 160),
                                   (1100, 160, 960), (100000, 32, 16), (1111, 40, 240), (2000, 8, 8)])
@pytest.mark.parametrize('mode', ['plain', 'relu6', 'bias', 'hswish'])
@pytest.mark.parametrize('nrep', [1, 16])
def test_pwconv_fwd_f32_training_forward_register_kernel(M, K, N, mode, nrep):
    from torchdet3d import _native as Nt
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    sc, sh = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3).cuda()
    p = None if mode in ('plain', 'bias') else Nt.prologue(sc, sh, None, mode, False)
    a = x if mode in ('plain', 'bias') else _act(x * sc + sh, mode)
    bias = (torch.randn(N, generator=g) * 3).cuda() if mode == 'bias' else None
    ref = a.double() @ w.double().t() + (bias.double() if bias is not None else 0.)
    runs = []
    for _ in range(2):
        y = torch.full((M + 1, N), 7.0, device='cuda')
        stats = torch.zeros(nrep, 2 * N, device='cuda', dtype=torch.float64)
        Nt.call('t3d_set_reduction_replicas', nrep, 2 * N)
        n0 = Nt.launch_count()
        try:
            Nt.call('t3d_pwconv_fwd', Nt.F32, Nt.ptr(x), p, Nt.ptr(w), Nt.ptr(bias), Nt.ptr(y), Nt.ptr(stats), M, 1, K, N, Nt.stream())
        finally:
            Nt.call('t3d_set_reduction_replicas', 1, 0)
        torch.cuda.synchronize()
        assert Nt.launch_count() - n0 == 1 and torch.all(y[M] == 7.0)
        runs.append((y[:M].clone(), stats.sum(0).view(2, N).clone()))
        if nrep > 1 and M >= 4096:
            assert (stats.abs().sum(1) > 0).sum() > 1          # more than one replica received sums
    (y, st), (y2, st2) = runs
    assert torch.equal(y, y2) and torch.equal(st, st2)
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().cpu().numpy(), atol=1e-5 * max(1., ref.abs().max().item()), rtol=1e-5)
    np.testing.assert_allclose(st[0].cpu().numpy(), y.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
    np.testing.assert_allclose(st[1].cpu().numpy(), (y.double() ** 2).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)


# fp32 storage, squeeze-excite gated operand (MobileNetV3's projection convs) on the register-operand kernel (variants 7 / 8): gate
# before the activation (hard-swish) or after it (ReLU), inference and training forward (BatchNorm sums); against fp64
@pytest.mark.parametrize('B,HW,K,N', [(8, 196, 72, 40), (9, 130, 120, 40), (16, 100, 480, 112), (5, 333, 960, 160), (256, 49, 672, 160)])
@pytest.mark.parametrize('mode', ['se_pre', 'se_post'])
@pytest.mark.parametrize('train', [False, True])
def test_pwconv_fwd_f32_gated_register_kernel(B, HW, K, N, mode, train):
    from torchdet3d import _native as Nt
    g = torch.Generator().manual_seed(B + HW + K + N + 1)
    M = B * HW
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    sc, sh = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3).cuda()
    se = torch.rand(B, K, generator=g).cuda()
    sev = se.repeat_interleave(HW, 0)
    act = 'hswish' if mode == 'se_pre' else 'relu'
    u = x * sc + sh
    a = _act(u * sev, act) if mode == 'se_pre' else _act(u, act) * sev
    ref = a.double() @ w.double().t()
    p = Nt.prologue(sc, sh, se, act, mode == 'se_post')
    y = torch.full((M + 1, N), 7.0, device='cuda')
    stats = torch.zeros(4, 2 * N, device='cuda', dtype=torch.float64) if train else None
    if train:
        Nt.call('t3d_set_reduction_replicas', 4, 2 * N)
    n0 = Nt.launch_count()
    try:
        Nt.call('t3d_pwconv_fwd', Nt.F32, Nt.ptr(x), p, Nt.ptr(w), None, Nt.ptr(y), Nt.ptr(stats), M, HW, K, N, Nt.stream())
    finally:
        Nt.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    assert Nt.launch_count() - n0 == 1 and torch.all(y[M] == 7.0)
    np.testing.assert_allclose(y[:M].cpu().numpy(), ref.float().cpu().numpy(), atol=1e-5 * max(1., ref.abs().max().item()), rtol=1e-5)
    if train:
        st = stats.sum(0).view(2, N)
        np.testing.assert_allclose(st[0].cpu().numpy(), y[:M].double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
        np.testing.assert_allclose(st[1].cpu().numpy(), (y[:M].double() ** 2).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)


# fp32 storage, inference: the materialising forward in ONE launch (csrc/pwconv_f32_reg.hip, variants 3 / 4): z is bit-equal to
# t3d_bn_apply's, y within 1e-5 of fp64; ragged contraction (24, 40: the store and the operand past K), ragged pixel counts
@pytest.mark.parametrize('M,K,N', [(4096, 16, 96), (3001, 24, 144), (1500, 32, 192), (1031, 96, 576), (1100, 160, 960),
                                   (1024, 320, 1280), (1111, 40, 240), (2000, 8, 8)])
@pytest.mark.parametrize('res', [False, True])
@pytest.mark.parametrize('train', [False, True])
def test_pwconv_fwd_mat_f32_inference_one_launch(M, K, N, res, train):
    from torchdet3d import _native as Nt
    g = torch.Generator().manual_seed(M + K + N + res)
    y3 = torch.randn(M, K, generator=g).cuda()
    r = torch.randn(M, K, generator=g).cuda() if res else None
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    sc, sh = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3).cuda()
    p = Nt.prologue(sc, sh, None, 'none', False)
    z_ref = torch.empty(M, K, device='cuda')
    Nt.call('t3d_bn_apply', Nt.F32, Nt.ptr(y3), p, Nt.ptr(r), Nt.ptr(z_ref), M, K, Nt.stream())
    z = torch.full((M + 1, K), 7.0, device='cuda')
    y = torch.full((M + 1, N), 7.0, device='cuda')
    # train: the training forward (BatchNorm sums of y into reduction replicas) in the same single launch
    stats = torch.zeros(8, 2 * N, device='cuda', dtype=torch.float64) if train else None
    if train:
        Nt.call('t3d_set_reduction_replicas', 8, 2 * N)
    n0 = Nt.launch_count()
    try:
        Nt.call('t3d_pwconv_fwd_mat', Nt.F32, Nt.ptr(y3), p, Nt.ptr(r), Nt.ptr(z), Nt.ptr(w), Nt.ptr(y), Nt.ptr(stats), M, 1, K, N, Nt.stream())
    finally:
        Nt.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    assert Nt.launch_count() - n0 == 1
    assert torch.equal(z[:M], z_ref) and torch.all(z[M] == 7.0) and torch.all(y[M] == 7.0)
    ref = z_ref.double() @ w.double().t()
    np.testing.assert_allclose(y[:M].cpu().numpy(), ref.float().cpu().numpy(), atol=1e-5 * max(1., ref.abs().max().item()), rtol=1e-5)
    if train:
        st = stats.sum(0).view(2, N)
        np.testing.assert_allclose(st[0].cpu().numpy(), y[:M].double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
        np.testing.assert_allclose(st[1].cpu().numpy(), (y[:M].double() ** 2).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4)


# bf16 materialising forward of the small planes' expansions with FRAGMENT-ORDER weights: csrc/pwconv_wide.hip (operand staged once,
# all output channels per workgroup) where the shape is one of its own, else the streaming kernel on the same layout; against
# t3d_bn_apply (z, bit for bit) and an fp64 product of the stored z; ragged pixel counts, contractions that are not a multiple of 32
# (80, 112, 200 outputs that end inside a 32-channel pair), statistics into replicas with a stride wider than the row
@pytest.mark.parametrize('M,K,N', [(1568, 160, 960), (1000, 64, 384), (777, 96, 576), (640, 320, 1280), (900, 80, 200), (1031, 112, 672),
                                   (513, 40, 120), (300, 24, 144), (2048, 16, 96)])
@pytest.mark.parametrize('res', [False, True])
@pytest.mark.parametrize('act', ['none', 'relu6', 'hswish'])
def test_pwconv_fwd_mat_bf16_fragment_order_weights(M, K, N, res, act, monkeypatch):
    from torchdet3d import _native as Nt
    monkeypatch.setenv('T3D_PW_WIDE', '1')      # (opt-in kernel, csrc/pwconv_wide.hip; read per call)
    g = torch.Generator().manual_seed(M + K + N + res)
    bf = torch.bfloat16
    y3 = torch.randn(M, K, generator=g).cuda().to(bf)
    r = torch.randn(M, K, generator=g).cuda().to(bf) if res else None
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    wq = _pack(w, bf)
    sc, sh = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3).cuda()
    p = Nt.prologue(sc, sh, None, act, False)
    z_ref = torch.empty(M, K, device='cuda', dtype=bf)
    Nt.call('t3d_bn_apply', Nt.BF16, Nt.ptr(y3), p, Nt.ptr(r), Nt.ptr(z_ref), M, K, Nt.stream())
    z = torch.full((M + 1, K), 7.0, device='cuda', dtype=bf)
    y = torch.full((M + 1, N), 7.0, device='cuda', dtype=bf)
    nrep, stride = 4, 2 * N + 10
    stats = torch.zeros(nrep, stride, device='cuda', dtype=torch.float64)
    wf = _frag(wq)
    Nt.call('t3d_set_reduction_replicas', nrep, stride)
    n0 = Nt.launch_count()
    try:
        Nt.call('t3d_pwconv_fwd_mat', Nt.BF16 | Nt.W_FRAG, Nt.ptr(y3), p, Nt.ptr(r), Nt.ptr(z), Nt.ptr(wf), Nt.ptr(y), Nt.ptr(stats), M, 1, K, N,
                Nt.stream())
    finally:
        Nt.call('t3d_set_reduction_replicas', 1, 0)
    torch.cuda.synchronize()
    assert Nt.launch_count() - n0 == 1
    assert torch.equal(z[:M], z_ref) and torch.all(z[M] == 7.0) and torch.all(y[M] == 7.0)
    ref = z_ref.double() @ wq.double().t()
    np.testing.assert_allclose(y[:M].float().cpu().numpy(), ref.float().cpu().numpy(), atol=8e-3 * max(1., ref.abs().max().item()), rtol=8e-3)
    st = stats[:, :2 * N].sum(0).view(2, N)
    assert torch.all(stats[:, 2 * N:] == 0)
    yq = y[:M].double()
    # the sums are those of the STORED (bf16-rounded) outputs, snapped onto the fixed grid of common.h (relative 2^-22 of the range)
    np.testing.assert_allclose(st[0].cpu().numpy(), yq.sum(0).cpu().numpy(), rtol=1e-5, atol=2e-3 * M ** .5)
    np.testing.assert_allclose(st[1].cpu().numpy(), (yq ** 2).sum(0).cpu().numpy(), rtol=1e-4, atol=1e-2)


# projection-conv data gradients of the 14x14 / 7x7 stages: contraction over 96 / 160 / 320 channels -> the deep-round
# variants with hoisted epilogue loads (pwconv_stream.hip: HOIST), incl. ragged widths
DEEP_DG = [(16, 196, 576, 96), (8, 49, 960, 160), (3, 49, 960, 320), (5, 100, 384, 96), (7, 33, 200, 88), (2, 49, 104, 152)]


@pytest.mark.parametrize('B,HW,K,N', SHAPES + DEEP_DG + DEEP_BWD)
@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('mode', ['input', 'input_res', 'bnact', 'se_pre', 'persample'])
def test_pwconv_dgrad(B, HW, K, N, dt, mode):
    """dx = (alpha*dz + beta*y + gamma) @ W, then the producer's activation derivative / residual."""
    from torchdet3d import _native as Nt
    dtype = _dt(dt)
    g = torch.Generator().manual_seed(B + HW + K + N + 7)
    M = B * HW
    dz, y = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    w = torch.randn(N, K, generator=g) / N ** 0.5
    xraw = torch.randn(M, K, generator=g)
    res = torch.randn(M, K, generator=g)
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    se = torch.rand(B, K, generator=g)
    ps = mode == 'persample'
    shp = (B, N) if ps else (N,)
    alpha, gamma = torch.rand(shp, generator=g) + 0.5, torch.randn(shp, generator=g) * 0.1
    beta = torch.randn(N, generator=g) * 0.2
    rep = (lambda t: t.repeat_interleave(HW, 0)) if ps else (lambda t: t)
    dy = rep(alpha) * _q(dz, dtype) + beta * _q(y, dtype) + rep(gamma)
    dy = _q(dy, dtype)
    gref = dy.double() @ _q(w, dtype).double()
    xq = _q(xraw, dtype)
    act = {'bnact': 'hswish', 'se_pre': 'relu6', 'persample': 'relu'}.get(mode, 'none')
    if mode in ('bnact', 'persample'):
        gref = gref * _act_grad(xq * scale + shift, act).double()
    elif mode == 'se_pre':
        sev = se.repeat_interleave(HW, 0)
        gref = gref * _act_grad((xq * scale + shift) * sev, act).double()
    if mode == 'input_res':
        gref = gref + _q(res, dtype).double()
    d = lambda t: t.to('cuda', dtype)
    dzd, yd, xd, rd = d(dz), d(y), d(xraw), d(res)
    wt = _pack(w, dtype, transpose=True)
    dx = torch.empty(M, K, device='cuda', dtype=dtype)
    stats = torch.zeros(2 * K, device='cuda', dtype=torch.float64)
    psst = torch.zeros(B, K, 2, device='cuda')
    keep = [t.cuda().contiguous() for t in (alpha, beta, gamma, scale, shift, se)]
    bb = Nt.bnbwd(keep[0], keep[1], keep[2], ps)
    has_act = mode in ('bnact', 'se_pre', 'persample')
    pin = Nt.prologue(keep[3], keep[4], keep[5] if mode == 'se_pre' else None, act, False) if has_act else None
    use_ps = mode == 'se_pre'
    variants = [(Nt.dtype_code(dzd), wt)]
    if dt == 'bf16' and _takes_frag(N, K, mode in ('input', 'input_res', 'bnact')):
        variants.append((Nt.BF16 | Nt.W_FRAG, _frag(wt)))      # fragment-order copy of wt
    for code, wv in variants:
        dx.zero_(); stats.zero_()
        Nt.call('t3d_pwconv_dgrad', code, Nt.ptr(dzd), Nt.ptr(yd), bb, Nt.ptr(wv),
                Nt.ptr(xd) if has_act else None, pin, Nt.ptr(rd) if mode == 'input_res' else None, Nt.ptr(dx),
                Nt.ptr(stats) if (has_act and not use_ps) else None, Nt.ptr(psst) if use_ps else None,
                M, HW, K, N, Nt.stream())
        torch.cuda.synchronize()
        got = dx.float().cpu()
        tol = 2e-5 if dt == 'f32' else 1.5e-2
        np.testing.assert_allclose(got.numpy(), gref.float().numpy(), atol=tol * max(1., gref.abs().max().item()), rtol=tol)
        if has_act and not use_ps:
            st = stats.cpu().view(2, K)
            np.testing.assert_allclose(st[0].numpy(), got.double().sum(0).numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
            np.testing.assert_allclose(st[1].numpy(), (got.double() * xq.double()).sum(0).numpy(), rtol=1e-5,
                                       atol=1e-4 * M ** .5)
    if use_ps:
        p1 = got.double().view(B, HW, K).sum(1)
        p2 = (got.double() * xq.double()).view(B, HW, K).sum(1)
        np.testing.assert_allclose(psst.cpu()[..., 0].numpy(), p1.numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(psst.cpu()[..., 1].numpy(), p2.numpy(), rtol=1e-4, atol=1e-3)


# fp32 storage, data gradient of layers with >= 1024 pixels: the register-operand kernel's variant 6 (csrc/pwconv_f32_reg.hip)
# against round 1's LDS-tiled kernel on the same inputs (T3D_F32_TILED=1) and against fp64 -- values, both BatchNorm-backward
# sums (sum dx, sum dx . x_raw), spread over the reduction replicas, bit-identical from run to run
@pytest.mark.parametrize('M,K,N', [(4096, 16, 96), (3001, 24, 144), (2048, 144, 24), (1031, 96, 576), (1024, 160, 960),
                                   (1100, 960, 160), (50000, 32, 16), (1111, 40, 240), (2000, 8, 8)])
@pytest.mark.parametrize('mode', ['input', 'input_res', 'relu6', 'relu6_res', 'none_stats'])
def test_pwconv_dgrad_f32_register_kernel(M, K, N, mode):
    import os
    from torchdet3d import _native as Nt
    g = torch.Generator().manual_seed(M + K + N + 3)
    dz, y = torch.randn(M, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    wt = (torch.randn(K, N, generator=g) / N ** 0.5).cuda()            # [fwd input channels][fwd output channels]
    xraw, res = torch.randn(M, K, generator=g).cuda() * 3, torch.randn(M, K, generator=g).cuda()
    scale, shift = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3 + 2).cuda()
    alpha, beta, gamma = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.2).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    bb = Nt.bnbwd(alpha, beta, gamma, False)
    has_x = mode in ('relu6', 'relu6_res', 'none_stats')
    act = 'relu6' if mode.startswith('relu6') else 'none'
    pin = Nt.prologue(scale, shift, None, act, False) if has_x else None
    with_res = mode.endswith('_res')
    dy = (alpha * dz + beta * y + gamma).double()
    ref = dy @ wt.double().t()
    if act == 'relu6':
        u = (xraw * scale + shift).double()
        ref = ref * ((u > 0) & (u < 6)).double()
    if with_res:
        ref = ref + res.double()
    outs = {}
    for tag in ('reg', 'reg2', 'tiled'):
        os.environ.pop('T3D_F32_TILED', None)
        if tag == 'tiled':
            os.environ['T3D_F32_TILED'] = '1'
        dx = torch.full((M + 1, K), 7.0, device='cuda')
        nrep = 1 if tag == 'tiled' else 8
        stats = torch.zeros(nrep, 2 * K, device='cuda', dtype=torch.float64)
        Nt.call('t3d_set_reduction_replicas', nrep, 2 * K)
        n0 = Nt.launch_count()
        try:
            Nt.call('t3d_pwconv_dgrad', Nt.F32, Nt.ptr(dz), Nt.ptr(y), bb, Nt.ptr(wt), Nt.ptr(xraw) if has_x else None, pin,
                    Nt.ptr(res) if with_res else None, Nt.ptr(dx), Nt.ptr(stats) if has_x else None, None, M, 1, K, N, Nt.stream())
        finally:
            Nt.call('t3d_set_reduction_replicas', 1, 0)
            os.environ.pop('T3D_F32_TILED', None)
        torch.cuda.synchronize()
        assert Nt.launch_count() - n0 == 1 and torch.all(dx[M] == 7.0)
        outs[tag] = (dx[:M].clone(), stats.sum(0).view(2, K).clone())
    (dx, st), (dx2, st2), (dxt, stt) = outs['reg'], outs['reg2'], outs['tiled']
    assert torch.equal(dx, dx2) and torch.equal(st, st2)
    scale_ = max(1., ref.abs().max().item())
    np.testing.assert_allclose(dx.cpu().numpy(), ref.float().cpu().numpy(), atol=2e-5 * scale_, rtol=2e-5)
    np.testing.assert_allclose(dx.cpu().numpy(), dxt.cpu().numpy(), atol=2e-5 * scale_, rtol=2e-5)
    if has_x:
        np.testing.assert_allclose(st[0].cpu().numpy(), dx.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
        np.testing.assert_allclose(st[1].cpu().numpy(), (dx.double() * xraw.double()).sum(0).cpu().numpy(), rtol=1e-5, atol=1e-4 * M ** .5)
        np.testing.assert_allclose(st.cpu().numpy(), stt.cpu().numpy(), rtol=1e-5, atol=1e-3 * M ** .5)


# (round 6: 'se_pre' / 'se_post' = a squeeze-excite gate on the operand WITHOUT per-sample coefficients -- the projection layers of
# MobileNetV3's gated blocks: the bf16 kernel stages the gate slice of a workgroup's samples in LDS; (168, 196, 672, 112) and
# (85, 784, 512, 128) reach the seven- / eight-column tiles at three row tiles per wave, (40, 196, 480, 112) the one-row-tile form)
@pytest.mark.parametrize('B,HW,K,N', SHAPES + [(16, 196, 96, 576), (64, 49, 160, 960), (2, 49, 320, 1280), (3, 49, 960, 160),
                                               (168, 196, 672, 112), (40, 196, 480, 112), (85, 784, 512, 128), (37, 49, 672, 160)])
@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('mode', ['plain', 'bnact', 'se_pre_ps', 'se_pre', 'se_post'])
def test_pwconv_wgrad(B, HW, K, N, dt, mode):
    from torchdet3d import _native as Nt
    dtype = _dt(dt)
    g = torch.Generator().manual_seed(B + HW + K + N + 11)
    M = B * HW
    dz, y = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    x = torch.randn(M, K, generator=g)
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    se = torch.rand(B, K, generator=g)
    ps = mode == 'se_pre_ps'
    shp = (B, N) if ps else (N,)
    alpha, gamma = torch.rand(shp, generator=g) + 0.5, torch.randn(shp, generator=g) * 0.1
    beta = torch.randn(N, generator=g) * 0.2
    rep = (lambda t: t.repeat_interleave(HW, 0)) if ps else (lambda t: t)
    dy = _q(rep(alpha) * _q(dz, dtype) + beta * _q(y, dtype) + rep(gamma), dtype)
    xq = _q(x, dtype)
    if mode == 'plain':
        a = xq
    elif mode == 'bnact':
        a = _act(xq * scale + shift, 'relu6')
    elif mode == 'se_post':
        a = _act(xq * scale + shift, 'hswish') * se.repeat_interleave(HW, 0)
    else:
        a = _act((xq * scale + shift) * se.repeat_interleave(HW, 0), 'hswish')
    if mode == 'bnact' and dt == 'bf16':
        # the bf16 kernel forms ReLU6 as 6 * clamp01((s/6) x + t/6) (one packed instruction with the clamp modifier) and rounds
        # THAT operand to bf16; the 6 is applied to the fp32 accumulators -- same accuracy, another rounding point
        sixth = torch.tensor(0.16666667163372040, dtype=torch.float32)
        a = _q((xq * (scale * sixth) + shift * sixth).clamp(0, 1), dtype) * 6
    else:
        a = _q(a, dtype)
    ref = dy.double().t() @ a.double()
    d = lambda t: t.to('cuda', dtype)
    dzd, yd, xd = d(dz), d(y), d(x)
    keep = [t.cuda().contiguous() for t in (alpha, beta, gamma, scale, shift, se)]
    bb = Nt.bnbwd(keep[0], keep[1], keep[2], ps)
    pro = None if mode == 'plain' else Nt.prologue(keep[3], keep[4], keep[5] if mode.startswith('se_') else None,
                                                   'relu6' if mode == 'bnact' else 'hswish', mode == 'se_post')
    dw = torch.zeros(N, K, device='cuda')
    Nt.call('t3d_pwconv_wgrad', Nt.dtype_code(dzd), Nt.ptr(dzd), Nt.ptr(yd), bb, Nt.ptr(xd), pro, Nt.ptr(dw),
            M, HW, K, N, Nt.stream())
    torch.cuda.synchronize()
    tol = 2e-5 if dt == 'f32' else 2e-3   # operands are pre-rounded to bf16 in the reference, accumulation is fp32
    np.testing.assert_allclose(dw.cpu().numpy(), ref.float().numpy(), atol=tol * max(1., ref.abs().max().item()), rtol=tol)


# fp32 storage, weight gradient of layers with >= 1024 pixels, with the caller's workspace set: the register-operand kernel
# (csrc/pwconv_f32_wgrad.hip) against fp64 and against round 1's LDS-tiled kernel (T3D_F32_TILED=1); accumulates into dw; ragged
# channel counts on both sides (24, 40, 144: blocks past the matrix edge), ragged pixel counts, bit-identical from run to run
@pytest.mark.parametrize('M,K,N', [(4096, 16, 96), (3001, 24, 144), (2048, 144, 24), (1031, 96, 576), (1024, 160, 960),
                                   (1100, 960, 160), (100000, 32, 16), (1111, 40, 240), (2000, 8, 8), (5000, 320, 1280)])
@pytest.mark.parametrize('mode', ['plain', 'relu6', 'hswish'])
def test_pwconv_wgrad_f32_register_kernel(M, K, N, mode):
    import os
    from torchdet3d import _native as Nt
    g = torch.Generator().manual_seed(M + K + N + 5)
    dz, y = torch.randn(M, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    x = torch.randn(M, K, generator=g).cuda() * 3
    scale, shift = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3 + 1).cuda()
    alpha, beta, gamma = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.2).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    bb = Nt.bnbwd(alpha, beta, gamma, False)
    pro = None if mode == 'plain' else Nt.prologue(scale, shift, None, mode, False)
    a = x if mode == 'plain' else _act(x * scale + shift, mode)
    ref = (alpha * dz + beta * y + gamma).double().t() @ a.double()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
    base = torch.randn(N, K, generator=g).cuda()
    outs = {}
    for tag in ('reg', 'reg2', 'tiled'):
        os.environ.pop('T3D_F32_TILED', None)
        if tag == 'tiled':
            os.environ['T3D_F32_TILED'] = '1'
        dw = base.clone()
        Nt.call('t3d_set_workspace', Nt.ptr(ws), ws.numel())
        n0 = Nt.launch_count()
        try:
            Nt.call('t3d_pwconv_wgrad', Nt.F32, Nt.ptr(dz), Nt.ptr(y), bb, Nt.ptr(x), pro, Nt.ptr(dw), M, 1, K, N, Nt.stream())
        finally:
            Nt.call('t3d_set_workspace', None, 0)
            os.environ.pop('T3D_F32_TILED', None)
        torch.cuda.synchronize()
        outs[tag] = (dw - base, Nt.launch_count() - n0)
    assert torch.equal(outs['reg'][0], outs['reg2'][0])
    tol = 2e-5 * max(1., ref.abs().max().item())
    np.testing.assert_allclose(outs['reg'][0].cpu().numpy(), ref.float().cpu().numpy(), atol=tol, rtol=2e-5)
    np.testing.assert_allclose(outs['reg'][0].cpu().numpy(), outs['tiled'][0].cpu().numpy(), atol=tol, rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('M,HW,K,N', [(4096, 64, 16, 96), (3000, 100, 24, 144), (2048, 64, 32, 192), (1024, 16, 64, 384),
                                      (520, 8, 16, 96), (2048, 64, 8, 48)])     # K=8, N=48: N+K+8 = 64, the pair only
@pytest.mark.parametrize('res', [False, True])
def test_pw_yfree_backward(M, HW, K, N, res):
    """y-free expand-layer backward (t3d_pwconv_*_yfree) vs fp64 autograd of conv1x1 + the BatchNorm-backward affine,
    and vs the regular kernels that read y: same sums, reassociated (bf16 storage)."""
    from torchdet3d import _native as N_
    g = torch.Generator().manual_seed(M + K)
    bf = torch.bfloat16
    x = torch.randn(M, K, generator=g).to(bf)
    w = (torch.randn(N, K, generator=g) / K ** .5).to(bf)
    y = (x.float() @ w.float().t()).to(bf)                     # what the forward stored
    dz = torch.randn(M, N, generator=g).to(bf)
    alpha, beta, gamma = torch.rand(N, generator=g) + .5, torch.randn(N, generator=g) * .2, torch.randn(N, generator=g) * .1
    r = torch.randn(M, K, generator=g).to(bf)
    # fp64 reference on the stored operands (y-free uses x W^T in place of the rounded y: bf16-rounding-level difference)
    dy = alpha.double() * dz.double() + beta.double() * (x.double() @ w.double().t()) + gamma.double()
    dx_ref = dy @ w.double() + (r.double() if res else 0)
    dw_ref = dy.t() @ x.double()
    dev = 'cuda'
    xd, wd, yd, dzd, rd = x.to(dev), w.to(dev), y.to(dev), dz.to(dev), r.to(dev)
    al, be, ga = alpha.to(dev), beta.to(dev), gamma.to(dev)
    bb = N_.bnbwd(al, be, ga, False)
    NP, KP = (N + 31) // 32 * 32, (K + 31) // 32 * 32
    wcat = torch.empty(K, NP + KP, device=dev, dtype=bf)
    cvec = torch.empty(K, device=dev)
    wtd = wd.t().contiguous()           # the transposed packed copy (engine: Net.wt)
    N_.call('t3d_pwconv_yfree_prep', N_.ptr(wtd), bb, N_.ptr(wcat), N_.ptr(cvec), K, N, N_.stream())
    # Wcat itself: [alpha.W | W^T diag(beta) W] and c = gamma^T W against fp64 on the stored weights (bf16 output)
    wq = w.double()
    np.testing.assert_allclose(wcat[:, :N].float().cpu().numpy(), (alpha.double()[:, None] * wq).t().numpy(), rtol=1e-2, atol=1e-3)
    Qref = wq.t() @ (beta.double()[:, None] * wq)
    np.testing.assert_allclose(wcat[:, NP:NP + K].float().cpu().numpy(), Qref.numpy(), rtol=2e-2, atol=2e-2 * float(Qref.abs().max()))
    assert not wcat[:, N:NP].float().abs().sum().item() and not wcat[:, NP + K:].float().abs().sum().item()      # zero padding
    np.testing.assert_allclose(cvec.cpu().numpy(), (gamma.double() @ wq).numpy(), rtol=1e-4, atol=1e-5)
    dx = torch.empty(M, K, device=dev, dtype=bf)
    yraw = torch.randn(M, K, generator=g).to(bf).to(dev)      # raw tensor of x's producer (BatchNorm-backward sums)
    st1 = torch.zeros(2 * K, device=dev, dtype=torch.float64)
    N_.call('t3d_pwconv_dgrad_yfree', N_.ptr(dzd), N_.ptr(xd), N_.ptr(wcat), N_.ptr(cvec), N_.ptr(yraw), None,
            N_.ptr(rd) if res else None, N_.ptr(dx), N_.ptr(st1), M, HW, K, N, N_.stream())
    ws = torch.empty(16 << 20, device=dev, dtype=torch.uint8)
    dw = torch.zeros(N, K, device=dev)
    N_.call('t3d_set_workspace', N_.ptr(ws), ws.numel())
    try:
        N_.call('t3d_pwconv_wgrad_yfree', N_.ptr(dzd), N_.ptr(xd), bb, N_.ptr(wd), N_.ptr(dw), M, HW, K, N, N_.stream())
        dw2 = torch.zeros(N, K, device=dev)
        N_.call('t3d_pwconv_wgrad', N_.BF16, N_.ptr(dzd), N_.ptr(yd), bb, N_.ptr(xd), None, N_.ptr(dw2), M, HW, K, N, N_.stream())
    finally:
        N_.call('t3d_set_workspace', None, 0)
    wt = wd.t().contiguous()
    dx2 = torch.empty(M, K, device=dev, dtype=bf)
    st2 = torch.zeros(2 * K, device=dev, dtype=torch.float64)
    N_.call('t3d_pwconv_dgrad', N_.BF16, N_.ptr(dzd), N_.ptr(yd), bb, N_.ptr(wt), N_.ptr(yraw), None,
            N_.ptr(rd) if res else None, N_.ptr(dx2), N_.ptr(st2), None, M, HW, K, N, N_.stream())
    torch.cuda.synchronize()
    for st_, d_ in ((st1, dx), (st2, dx2)):      # sums of the stored gradient, and of gradient * raw
        ref0, ref1 = d_.double().sum(0), (d_.double() * yraw.double()).sum(0)
        np.testing.assert_allclose(st_[:K].cpu().numpy(), ref0.cpu().numpy(), rtol=1e-4, atol=1e-3 * M ** .5)
        np.testing.assert_allclose(st_[K:].cpu().numpy(), ref1.cpu().numpy(), rtol=1e-4, atol=1e-3 * M ** .5)
    sx, sw = dx_ref.abs().max().item(), dw_ref.abs().max().item()
    ex, ex2 = (dx.double().cpu() - dx_ref).abs().max().item() / sx, (dx2.double().cpu() - dx_ref).abs().max().item() / sx
    ew, ew2 = (dw.double().cpu() - dw_ref).abs().max().item() / sw, (dw2.double().cpu() - dw_ref).abs().max().item() / sw
    assert ex < 1.5e-2 and ex < 2 * ex2 + 4e-3, (ex, ex2)          # bf16 output rounding dominates both
    assert ew < 1e-2 and ew < 2 * ew2 + 4e-3, (ew, ew2)
    # ---- the one-pass form (round 4): data gradient + weight-gradient products from the same staged rows
    need = N_.lib().t3d_pwconv_bwd_yfree_scratch(M, K, N)
    if N + K + 8 <= 64:
        assert need == 0      # prep2's 64-column rows are narrower than the kernel's narrowest staged tile: refused
    if need > 0:
        wdl = torch.zeros((K + 15) // 16 * 16, (N + K + 8 + 63) // 64 * 64, device=dev, dtype=bf)
        N_.call('t3d_pwconv_yfree_prep2', N_.ptr(wtd), bb, N_.ptr(wcat), N_.ptr(cvec), N_.ptr(wdl), K, N, N_.stream())
        scratch = torch.empty(need, device=dev, dtype=torch.uint8)
        dx3 = torch.empty(M, K, device=dev, dtype=bf)
        st3 = torch.zeros(2 * K, device=dev, dtype=torch.float64)
        dw3 = torch.zeros(N, K, device=dev)
        N_.call('t3d_set_reduction_replicas', 1, 0)
        N_.call('t3d_pwconv_bwd_yfree', N_.ptr(dzd), N_.ptr(xd), N_.ptr(wdl), N_.ptr(yraw), None, N_.ptr(rd) if res else None,
                N_.ptr(dx3), N_.ptr(st3), N_.ptr(scratch), need, M, HW, K, N, N_.stream())
        N_.call('t3d_pwconv_wgrad_yfree_finish', N_.ptr(scratch), bb, N_.ptr(wd), N_.ptr(dw3), M, K, N, N_.stream())
        torch.cuda.synchronize()
        ex3 = (dx3.double().cpu() - dx_ref).abs().max().item() / sx
        ew3 = (dw3.double().cpu() - dw_ref).abs().max().item() / sw
        assert ex3 < 1.5e-2 and ex3 < 2 * ex2 + 4e-3, (ex3, ex2)
        assert ew3 < 1e-2 and ew3 < 2 * ew2 + 4e-3, (ew3, ew2)
        ref0, ref1 = dx3.double().sum(0), (dx3.double() * yraw.double()).sum(0)
        np.testing.assert_allclose(st3[:K].cpu().numpy(), ref0.cpu().numpy(), rtol=1e-4, atol=1e-3 * M ** .5)
        np.testing.assert_allclose(st3[K:].cpu().numpy(), ref1.cpu().numpy(), rtol=1e-4, atol=1e-3 * M ** .5)
        # the same numbers as the pair up to the bf16 rounding of the bias c (it rides in the weight matrix here)
        assert (dx3.float() - dx.float()).abs().max().item() <= 2e-2 * sx
        # ---- round 5: the weight rows built in the launch's own prologue (t3d_pwconv_bwd_yfree_w) -- bit for bit what prep2 +
        # t3d_pwconv_bwd_yfree give: data gradient, BatchNorm-backward sums, partial weight-gradient tiles
        scratch4 = torch.empty(need, device=dev, dtype=torch.uint8)
        dx4 = torch.empty(M, K, device=dev, dtype=bf)
        st4 = torch.zeros(2 * K, device=dev, dtype=torch.float64)
        dw4 = torch.zeros(N, K, device=dev)
        N_.call('t3d_pwconv_bwd_yfree_w', N_.ptr(dzd), N_.ptr(xd), N_.ptr(wtd), bb, N_.ptr(yraw), None, N_.ptr(rd) if res else None,
                N_.ptr(dx4), N_.ptr(st4), N_.ptr(scratch4), need, M, HW, K, N, N_.stream())
        N_.call('t3d_pwconv_wgrad_yfree_finish', N_.ptr(scratch4), bb, N_.ptr(wd), N_.ptr(dw4), M, K, N, N_.stream())
        torch.cuda.synchronize()
        assert torch.equal(dx4, dx3) and torch.equal(st4, st3) and torch.equal(dw4, dw3)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('ratio', [3.0, 30.0])
def test_batch_variance_from_the_one_pass_sums_when_the_mean_dominates(dt, ratio):
    """The conv epilogues emit sum(y), sum(y^2) in one pass (fp32 per-lane partials, fp64 across lanes / blocks) and
    t3d_bn_finalize forms var = E[y^2] - E[y]^2 in fp64; PyTorch's BatchNorm is two-pass.  What that costs when
    |mean| >> std (a bias of `ratio` standard deviations on every output channel, production pixel count): the relative
    error of the variance stays below (1 + ratio^2) x the 1e-6 relative error of the sums."""
    from torchdet3d import _native as Nt
    dtype = _dt(dt)
    g = torch.Generator().manual_seed(17)
    M, HW, K, N = 256 * 196, 196, 64, 96
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.full((N,), ratio) * (torch.rand(N, generator=g) + 0.5)
    xd, wd, bd = x.to('cuda', dtype), _pack(w, dtype), bias.cuda()
    y = torch.empty(M, N, device='cuda', dtype=dtype)
    stats = torch.zeros(2 * N, device='cuda', dtype=torch.float64)
    Nt.call('t3d_pwconv_fwd', Nt.dtype_code(xd), Nt.ptr(xd), None, Nt.ptr(wd), Nt.ptr(bd), Nt.ptr(y), Nt.ptr(stats),
            M, HW, K, N, Nt.stream())
    torch.cuda.synchronize()
    yd = y.double()
    st = stats.view(2, N)
    var = (st[1] / M - (st[0] / M) ** 2).cpu()
    ref = yd.var(0, unbiased=False).cpu()          # two-pass, fp64, of the stored outputs
    rel = ((var - ref).abs() / ref).max().item()
    assert rel < 2e-6 * (1 + (1.5 * ratio) ** 2), rel
    print(f'   {dt} |mean| ~ {ratio} std: worst relative variance error {rel:.1e}')
