"""GPU: inference-mode inverted-residual blocks of the 14x14 / 7x7 stages as ONE launch (`t3d_ir_block_eval`, the
expanded tensors stay in LDS) against the launch-per-layer path on the same weights -- same rounding points, so the two
agree to a few bf16 steps -- and against the fp32 CPU oracle through the usual bf16 gate."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,B,HW', [('mobilenetv2', 128, 224), ('mobilenetv2', 100, 112), ('mobilenetv3_large', 96, 224)])
def test_fused_eval_blocks_match_the_layer_by_layer_path(name, B, HW):
    from oracle.weights import make_inputs, make_state_dict
    from torchdet3d.models.engine import Net
    imgs, _, cats = make_inputs(B, HW, HW, 9)
    sd = make_state_dict(name, 9)
    outs, launches = {}, {}
    for fused in (True, False):
        net = Net(name, 9, 'cuda', torch.bfloat16)
        net.load_state_dict(sd)
        net._fused_eval = fused
        with torch.no_grad():
            kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=False)
            feat = net.extract_features(imgs.cuda())
        outs[fused] = (kp.float().cpu(), lg.float().cpu(), feat.float().cpu())
        launches[fused] = sum(1 for i, blk in enumerate(net.arch.blocks) if blk.expand and not blk.se and blk.s == 1 and blk.k == 3)
        del net
    a, b = outs[True], outs[False]
    # keypoints are sigmoid outputs in (0, 1); a few bf16 steps of the feature map move them by < 1e-3
    assert (a[0] - b[0]).abs().max().item() < 2e-3, (a[0] - b[0]).abs().max().item()
    assert (a[1] - b[1]).abs().max().item() < 2e-2 * max(1.0, b[1].abs().max().item())
    rel = ((a[2] - b[2]).norm() / b[2].norm()).item()
    # (round 4: the launch-per-layer kernels form ReLU6 as 6 * clamp01(.) and round THAT operand to bf16, the fused block
    #  rounds the activated value itself: one more rounding point that differs between the two bf16 paths -- 1.1e-2 measured,
    #  8e-3 before)
    assert rel < 2e-2, rel
    if name == 'mobilenetv2' and HW == 224:
        assert torch.equal(a[1].argmax(1), b[1].argmax(1))


def test_fused_block_kernel_against_torch_on_the_rounded_operands():
    """One block (14x14, 64 -> 384 -> 64, ReLU6, residual) against torch fp64 with the kernel's rounding points."""
    import torch.nn.functional as F
    from torchdet3d import _native as N
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Ce, Cout = 5, 14, 14, 64, 384, 64
    q = lambda t: t.to(torch.bfloat16).double()
    x = torch.randn(B, H, W, Cin, generator=g)
    w1 = torch.randn(Ce, Cin, generator=g) / Cin ** .5
    wd = torch.randn(Ce, 9, generator=g) * 0.4
    w2 = torch.randn(Cout, Ce, generator=g) / Ce ** .5
    s1, h1 = torch.rand(Ce, generator=g) + .5, torch.randn(Ce, generator=g) * .3
    s2, h2 = torch.rand(Ce, generator=g) + .5, torch.randn(Ce, generator=g) * .3
    s3, h3 = torch.rand(Cout, generator=g) + .5, torch.randn(Cout, generator=g) * .3
    dev = lambda t, dt=torch.float32: t.to('cuda', dt).contiguous()
    z = torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16)
    keep = [dev(x, torch.bfloat16), dev(w1, torch.bfloat16), dev(s1), dev(h1), dev(wd), dev(s2), dev(h2), dev(w2, torch.bfloat16),
            dev(s3), dev(h3)]
    N.call('t3d_ir_block_eval', N.ptr(keep[0]), N.ptr(keep[1]), N.ptr(keep[2]), N.ptr(keep[3]), N.ACT['relu6'], N.ptr(keep[4]),
           N.ptr(keep[5]), N.ptr(keep[6]), N.ACT['relu6'], N.ptr(keep[7]), N.ptr(keep[8]), N.ptr(keep[9]), 1, N.ptr(z),
           B, H, W, Cin, Ce, Cout, N.stream())
    torch.cuda.synchronize()
    xq = q(x)
    e = q((xq.view(-1, Cin) @ q(w1).t()).float()).view(B, H, W, Ce)
    a1 = torch.clamp(e * s1.double() + h1.double(), 0, 6).float().double()
    d = F.conv2d(a1.permute(0, 3, 1, 2), wd.double().view(Ce, 1, 3, 3), None, 1, 1, 1, Ce).permute(0, 2, 3, 1)
    d = q(d.float())
    a2 = q(torch.clamp(d * s2.double() + h2.double(), 0, 6).float())
    y3 = q((a2.reshape(-1, Ce) @ q(w2).t()).float()).view(B, H, W, Cout)
    ref = q((y3 * s3.double() + h3.double() + xq).float())
    got = z.double().cpu()
    err = (got - ref).abs()
    # bf16 steps where an intermediate sat on a rounding boundary; nothing systematic
    assert err.max().item() <= 0.25 and (err > 0.05).double().mean().item() < 2e-3, (err.max().item(), (err > 0.05).double().mean().item())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=0.3, rtol=0.05)
