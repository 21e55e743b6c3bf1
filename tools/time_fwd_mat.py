"""Isolated timing of the materialising 1x1 conv (t3d_pwconv_fwd_mat) against the plain forward on the small-stage expand shapes.
usage: python tools/time_fwd_mat.py      (T3D_PW_WIDE=1 [T3D_PW_WIDE_MAX_M=...]: the last column through csrc/pwconv_wide.hip)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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

bf = torch.bfloat16
for M, HW, K, Nn in [(12544, 49, 160, 960), (12544, 49, 320, 1280), (50176, 196, 96, 576), (50176, 196, 64, 384), (50176, 196, 80, 200),
                     (50176, 196, 112, 672), (200704, 784, 32, 192), (200704, 784, 40, 120), (802816, 3136, 24, 144)]:
    y3 = torch.randn(M, K, device='cuda').to(bf)
    res = torch.randn(M, K, device='cuda').to(bf)
    z = torch.empty(M, K, device='cuda', dtype=bf)
    w = (torch.randn(Nn, K, device='cuda') / K ** .5).to(bf)
    sc, sh = torch.rand(K, device='cuda') + 0.5, torch.randn(K, device='cuda') * 0.2
    pro = N.prologue(sc, sh, None, 'none', False)
    y = torch.empty(M, Nn, device='cuda', dtype=bf)
    stats = torch.zeros(16, 2 * Nn, device='cuda', dtype=torch.float64)
    N.call('t3d_set_reduction_replicas', 8 if Nn > 160 else 16, 2 * Nn)
    t0 = timeit(lambda: N.call('t3d_pwconv_fwd', N.BF16, N.ptr(z), None, N.ptr(w), None, N.ptr(y), N.ptr(stats), M, HW, K, Nn, N.stream()))
    t1 = timeit(lambda: N.call('t3d_pwconv_fwd_mat', N.BF16, N.ptr(y3), pro, N.ptr(res), N.ptr(z), N.ptr(w), N.ptr(y), N.ptr(stats), M, HW, K, Nn, N.stream()))
    t2 = timeit(lambda: N.call('t3d_pwconv_fwd_mat', N.BF16, N.ptr(y3), pro, None, N.ptr(z), N.ptr(w), N.ptr(y), N.ptr(stats), M, HW, K, Nn, N.stream()))
    # fragment-order weights: the wide-output kernel (csrc/pwconv_wide.hip) where it takes the shape (T3D_PW_WIDE_MAX_M lifts its pixel limit)
    wf = torch.zeros(N.lib().t3d_pwconv_frag_bytes(Nn, K) // 2, device='cuda', dtype=bf)
    N.call('t3d_pwconv_pack_frag', N.ptr(w), N.ptr(wf), Nn, K, N.stream())
    t3 = timeit(lambda: N.call('t3d_pwconv_fwd_mat', N.BF16 | N.W_FRAG, N.ptr(y3), pro, N.ptr(res), N.ptr(z), N.ptr(wf), N.ptr(y), N.ptr(stats), M, HW, K, Nn, N.stream()))
    N.call('t3d_set_reduction_replicas', 1, 0)
    print(f'{M:6d} {K:4d}->{Nn:4d}: plain {t0:6.1f}  mat+res {t1:6.1f}  mat {t2:6.1f}  mat+res, fragment-order weights {t3:6.1f} us')
