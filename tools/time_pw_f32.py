"""Isolated timing of the fp32-storage inference 1x1 conv on MobileNetV2's layer shapes at batch 256: the register-operand kernel
(csrc/pwconv_f32_reg.hip) against round 1's LDS-tiled one (T3D_F32_TILED=1); --sweep: every task shape R x NT x V.
usage: python tools/time_pw_f32.py [--sweep]"""
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

B = 256
tot = [0., 0.]
for hw, K, Nn, cnt in [(112, 32, 16, 1), (112, 16, 96, 1), (56, 96, 24, 1), (56, 24, 144, 2), (56, 144, 24, 1), (28, 144, 32, 1),
                       (28, 32, 192, 3), (28, 192, 32, 2), (14, 192, 64, 1), (14, 64, 384, 4), (14, 384, 64, 3), (14, 384, 96, 1),
                       (14, 96, 576, 3), (14, 576, 96, 2), (7, 576, 160, 1), (7, 160, 960, 3), (7, 960, 160, 2), (7, 960, 320, 1),
                       (7, 320, 1280, 1)]:
    M = B * hw * hw
    x = torch.randn(M, K, device='cuda')
    w = torch.randn(Nn, K, device='cuda') / K ** .5
    sc, sh = torch.rand(K, device='cuda') + 0.5, torch.randn(K, device='cuda') * 0.2
    pro = N.prologue(sc, sh, None, 'relu6', False)
    y = torch.empty(M, Nn, device='cuda')
    f = lambda: N.call('t3d_pwconv_fwd', N.F32, N.ptr(x), pro, N.ptr(w), None, N.ptr(y), None, M, hw * hw, K, Nn, N.stream())
    os.environ.pop('T3D_F32_TILED', None)
    t0 = timeit(f)
    os.environ['T3D_F32_TILED'] = '1'
    t1 = timeit(f)
    os.environ.pop('T3D_F32_TILED', None)
    sweep = ''
    if '--sweep' in sys.argv:
        res = []
        for V in (1,):
            for R in (1, 2, 4):
                for NT in (2, 3, 4, 5, 6):
                    if R == 1:
                        continue
                    os.environ['T3D_F32_SHAPE'] = f'{R}{NT}'
                    res.append((timeit(f, 10), R, NT, V))
        os.environ.pop('T3D_F32_SHAPE', None)
        res.sort()
        sweep = '  best ' + ' '.join(f'{R}x{NT}x{V}:{t:.0f}' for t, R, NT, V in res[:5])
    gb = M * (K + Nn) * 4 / 1e3
    tot[0] += cnt * t0; tot[1] += cnt * t1
    print(f'{hw:4d}^2 {K:4d}->{Nn:4d} x{cnt}: reg {t0:6.1f} us {gb / t0:5.0f} GB/s {2e-6 * M * K * Nn / t0:5.1f} TF/s  tiled {t1:6.1f} us' + sweep)
print(f'per forward: reg {tot[0] / 1e3:.2f} ms, tiled {tot[1] / 1e3:.2f} ms')
