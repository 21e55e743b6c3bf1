"""Isolated timing of the fused expand + depthwise forward against the two launches it replaces, per MobileNetV2 block shape
(B = 256 @224).  usage: python tools/time_expdw.py [CS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch
from torchdet3d import _native as N

SHAPES = [(112, 16, 96, 2), (56, 24, 144, 1), (56, 24, 144, 2), (28, 32, 192, 1), (28, 32, 192, 2)]
B = 256


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


for H, K, C, s in SHAPES:
    Ho = (H - 1) // s + 1
    z = torch.randn(B, H, H, K, device='cuda').to(torch.bfloat16)
    w1 = (torch.randn(C, K, device='cuda') / K ** 0.5).to(torch.bfloat16)
    sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda') * 0.5
    wdw = torch.randn(C, 9, device='cuda') * 0.3
    y1 = torch.empty(B, H, H, C, device='cuda', dtype=torch.bfloat16)
    y2 = torch.empty(B, Ho, Ho, C, device='cuda', dtype=torch.bfloat16)
    stats = torch.zeros(16, 4 * C, device='cuda', dtype=torch.float64)
    N.call('t3d_set_reduction_replicas', 16, 4 * C)
    pro = N.prologue(sc, sh, None, 'relu6', False)
    M = B * H * H
    t_pw = timeit(lambda: N.call('t3d_pwconv_fwd', N.BF16, N.ptr(z), None, N.ptr(w1), None, N.ptr(y1), N.ptr(stats), M, H * H, K, C, N.stream()))
    t_st = timeit(lambda: N.call('t3d_pwconv_fwd', N.BF16, N.ptr(z), None, N.ptr(w1), None, None, N.ptr(stats), M, H * H, K, C, N.stream()))
    t_dw = timeit(lambda: N.call('t3d_dwconv_fwd', N.BF16, N.ptr(y1), pro, N.ptr(wdw), N.ptr(y2), (stats.data_ptr() + 16 * C), None, B, H, H, C, 3, s, N.stream()))
    t_f1 = timeit(lambda: N.call('t3d_expdw_fwd', N.BF16, N.ptr(z), N.ptr(w1), N.ptr(sc), N.ptr(sh), 2, N.ptr(wdw), N.ptr(y1), N.ptr(y2), (stats.data_ptr() + 16 * C), B, H, H, K, C, s, N.stream()))
    t_f0 = timeit(lambda: N.call('t3d_expdw_fwd', N.BF16, N.ptr(z), N.ptr(w1), N.ptr(sc), N.ptr(sh), 2, N.ptr(wdw), None, N.ptr(y2), (stats.data_ptr() + 16 * C), B, H, H, K, C, s, N.stream()))
    N.call('t3d_set_reduction_replicas', 1, 0)
    print(f'{H:4d}^2 {K:4d}->{C:4d} s{s}: pw {t_pw:7.1f}  dw {t_dw:7.1f}  (sum {t_pw + t_dw:7.1f}) | stats-only {t_st:6.1f}  fused+y1 {t_f1:7.1f}  fused {t_f0:7.1f} us')

