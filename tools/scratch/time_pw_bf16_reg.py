"""EXPERIMENT: bf16 inference 1x1 conv, register-operand kernel (csrc/pwconv_bf16_reg.hip, T3D_BF16_REG=1) against the streaming
kernel, MobileNetV2's layer shapes at batch 256; checks the two outputs against each other.   usage: python tools/time_pw_bf16_reg.py [--sweep]"""
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
bf = torch.bfloat16
tot = [0., 0.]
for hw, K, Nn, cnt in [(112, 32, 16, 1), (112, 16, 96, 1), (56, 96, 24, 1), (56, 24, 144, 2), (56, 144, 24, 1), (28, 144, 32, 1),
                       (28, 32, 192, 3), (28, 192, 32, 2), (14, 192, 64, 1), (14, 64, 384, 4), (14, 384, 64, 3), (14, 384, 96, 1),
                       (14, 96, 576, 3), (14, 576, 96, 2), (7, 576, 160, 1), (7, 160, 960, 3), (7, 960, 160, 2), (7, 960, 320, 1),
                       (7, 320, 1280, 1)]:
    M = B * hw * hw
    x = torch.randn(M, K, device='cuda').to(bf)
    w = (torch.randn(Nn, K, device='cuda') / K ** .5).to(bf)
    sc, sh = torch.rand(K, device='cuda') + 0.5, torch.randn(K, device='cuda') * 0.2
    pro = N.prologue(sc, sh, None, 'relu6', False)
    y = torch.empty(M, Nn, device='cuda', dtype=bf)
    f = lambda: N.call('t3d_pwconv_fwd', N.BF16, N.ptr(x), pro, N.ptr(w), None, N.ptr(y), None, M, hw * hw, K, Nn, N.stream())
    os.environ.pop('T3D_BF16_REG', None)
    t1 = timeit(f)
    y1 = y.clone()
    os.environ['T3D_BF16_REG'] = '1'
    y.zero_()
    t0 = timeit(f)
    err = (y.float() - y1.float()).abs().max().item()
    sweep = ''
    if '--sweep' in sys.argv:
        res = []
        for R in (1, 2, 4):
            for NP in (1, 2, 3):
                os.environ['T3D_BF16_SHAPE'] = f'{R}{NP}'
                res.append((timeit(f, 10), R, NP))
        os.environ.pop('T3D_BF16_SHAPE', None)
        res.sort()
        sweep = '  best ' + ' '.join(f'{R}x{NP}:{t:.0f}' for t, R, NP in res[:4])
    os.environ.pop('T3D_BF16_REG', None)
    gb = M * (K + Nn) * 2 / 1e3
    tot[0] += cnt * t0; tot[1] += cnt * t1
    print(f'{hw:4d}^2 {K:4d}->{Nn:4d} x{cnt}: reg {t0:6.1f} us {gb / t0:5.0f} GB/s  stream {t1:6.1f} us {gb / t1:5.0f} GB/s  maxdiff {err:.3g}' + sweep)
print(f'per forward: reg {tot[0] / 1e3:.2f} ms, stream {tot[1] / 1e3:.2f} ms')
