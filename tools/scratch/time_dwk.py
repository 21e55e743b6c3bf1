"""Isolated timing of MobileNetV3-large's 5x5 / squeeze-excite depthwise layers (csrc/dwconvk_stream.hip forward,
csrc/dwconv5_bwd_stream.hip backward) at batch 256, bf16 storage, per channels-per-thread setting (T3D_DWK_NC).
usage: python tools/time_dwk.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch
from torchdet3d import _native as N

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = 256
bf = torch.bfloat16
tot = 0.
for H, C, k, s, act in [(56, 72, 5, 2, 'relu'), (28, 120, 5, 1, 'relu'), (14, 480, 3, 1, 'hswish'), (14, 672, 3, 1, 'hswish'),
                        (14, 672, 5, 2, 'hswish'), (7, 960, 5, 1, 'hswish')]:
    pad = (k - 1) // 2
    Ho = (H + 2 * pad - k) // s + 1
    x = torch.randn(B * H * H, C, device='cuda').to(bf)
    w = torch.randn(C, k * k, device='cuda') / k
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.2
    pro = N.prologue(sc, sh, None, act, False)
    y = torch.empty(B * Ho * Ho, C, device='cuda', dtype=bf)
    stats = torch.zeros(16, 2 * C, device='cuda', dtype=torch.float64)
    gap = torch.zeros(B, C, device='cuda')
    N.call('t3d_set_reduction_replicas', 16, 2 * C)
    f = lambda: N.call('t3d_dwconv_fwd', N.BF16, N.ptr(x), pro, N.ptr(w), N.ptr(y), N.ptr(stats), N.ptr(gap), B, H, H, C, k, s, N.stream())
    t = timeit(f)
    N.call('t3d_set_reduction_replicas', 1, 0)
    mb = (B * H * H + B * Ho * Ho) * C * 2 / 1e6
    tot += t
    print(f'dwfwd {H:3d}^2 x{C:4d} k{k} s{s}: {t:7.1f} us  {mb / t / 1e0 / 1e6 * 1e6 / 1e6:.2f} TB/s', flush=True)
print(f'sum {tot:.0f} us')
