"""Isolated timing of the fp32-storage TRAINING 1x1 kernels on MobileNetV2's layer shapes at batch 256: forward with BatchNorm sums,
data gradient, weight gradient -- the register-operand kernels (csrc/pwconv_f32_reg.hip, csrc/pwconv_f32_wgrad.hip) against round
1's LDS-tiled ones (T3D_F32_TILED=1).   usage: python tools/time_pw_f32_bwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch
from torchdet3d import _native as N

def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def both(fn):
    os.environ.pop('T3D_F32_TILED', None)
    t0 = timeit(fn)
    os.environ['T3D_F32_TILED'] = '1'
    t1 = timeit(fn)
    os.environ.pop('T3D_F32_TILED', None)
    return t0, t1

B = 256
ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
tot = {k: [0., 0.] for k in ('fwd', 'dgrad', 'wgrad')}
# (conv input channels K, output channels N): forward x [M,K] -> y [M,N]
for hw, K, Nn, cnt in [(112, 32, 16, 1), (112, 16, 96, 1), (56, 96, 24, 1), (56, 24, 144, 2), (56, 144, 24, 1), (28, 144, 32, 1),
                       (28, 32, 192, 3), (28, 192, 32, 2), (14, 192, 64, 1), (14, 64, 384, 4), (14, 384, 64, 3), (14, 384, 96, 1),
                       (14, 96, 576, 3), (14, 576, 96, 2), (7, 576, 160, 1), (7, 160, 960, 3), (7, 960, 160, 2), (7, 960, 320, 1),
                       (7, 320, 1280, 1)]:
    M = B * hw * hw
    x = torch.randn(M, K, device='cuda')
    y = torch.empty(M, Nn, device='cuda')
    dz = torch.randn(M, Nn, device='cuda')
    dx = torch.empty(M, K, device='cuda')
    w = torch.randn(Nn, K, device='cuda') / K ** .5
    wt = w.t().contiguous()
    dw = torch.zeros(Nn, K, device='cuda')
    sc, sh = torch.rand(K, device='cuda') + 0.5, torch.randn(K, device='cuda') * 0.2
    al, be, ga = torch.rand(Nn, device='cuda') + 0.5, torch.randn(Nn, device='cuda') * 0.2, torch.randn(Nn, device='cuda') * 0.1
    pro = N.prologue(sc, sh, None, 'relu6', False)
    bb = N.bnbwd(al, be, ga, False)
    stats = torch.zeros(16, 2 * Nn, device='cuda', dtype=torch.float64)
    bst = torch.zeros(16, 2 * K, device='cuda', dtype=torch.float64)
    def fwd():
        N.call('t3d_set_reduction_replicas', 16, 2 * Nn)
        N.call('t3d_pwconv_fwd', N.F32, N.ptr(x), pro, N.ptr(w), None, N.ptr(y), N.ptr(stats), M, hw * hw, K, Nn, N.stream())
    def dgrad():
        N.call('t3d_set_reduction_replicas', 16, 2 * K)
        N.call('t3d_pwconv_dgrad', N.F32, N.ptr(dz), N.ptr(y), bb, N.ptr(wt), N.ptr(x), pro, None, N.ptr(dx), N.ptr(bst), None,
               M, hw * hw, K, Nn, N.stream())
    def wgrad():
        N.call('t3d_set_workspace', N.ptr(ws), ws.numel())
        N.call('t3d_pwconv_wgrad', N.F32, N.ptr(dz), N.ptr(y), bb, N.ptr(x), pro, N.ptr(dw), M, hw * hw, K, Nn, N.stream())
    y.normal_()
    r = {'fwd': both(fwd), 'dgrad': both(dgrad), 'wgrad': both(wgrad)}
    N.call('t3d_set_reduction_replicas', 1, 0)
    N.call('t3d_set_workspace', None, 0)
    for k, (a, b) in r.items():
        tot[k][0] += cnt * a; tot[k][1] += cnt * b
    print(f'{hw:4d}^2 {K:4d}->{Nn:4d} x{cnt}: ' + '   '.join(f'{k} {a:6.1f} / {b:6.1f} us' for k, (a, b) in r.items()) + '   (register-operand / LDS-tiled)', flush=True)
print('per step: ' + ', '.join(f'{k} {a / 1e3:.2f} / {b / 1e3:.2f} ms' for k, (a, b) in tot.items()))
