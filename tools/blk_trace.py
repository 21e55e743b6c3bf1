"""Phase times of the fused inference block kernel (needs HIPCC_EXTRA=-DT3D_BLK_TRACE build).  usage: blk_trace.py H Cin Ce Cout"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch
from torchdet3d import _native as N
H, Cin, Ce, Cout = [int(v) for v in sys.argv[1:5]]
B = 256
g = torch.Generator(device='cuda').manual_seed(0)
r = lambda *s: torch.randn(*s, device='cuda', generator=g)
x, w1, w2 = r(B, H, H, Cin).bfloat16(), (r(Ce, Cin) / Cin ** .5).bfloat16(), (r(Cout, Ce) / Ce ** .5).bfloat16()
wd, s1, h1, s2, h2, s3, h3 = r(Ce, 9) * .4, r(Ce).abs() + .5, r(Ce) * .3, r(Ce).abs() + .5, r(Ce) * .3, r(Cout).abs() + .5, r(Cout) * .3
z = torch.empty(B, H, H, Cout, device='cuda', dtype=torch.bfloat16)
fn = lambda: N.call('t3d_ir_block_eval', N.ptr(x), N.ptr(w1), N.ptr(s1), N.ptr(h1), 2, N.ptr(wd), N.ptr(s2), N.ptr(h2), 2, N.ptr(w2),
                    N.ptr(s3), N.ptr(h3), int(Cin == Cout), N.ptr(z), B, H, H, Cin, Ce, Cout, N.stream())
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print(f'block {H}x{H} {Cin}->{Ce}->{Cout}: {e0.elapsed_time(e1) * 50:.1f} us per launch')
lib = ctypes.CDLL(N.LIB_PATH)
if hasattr(lib, 't3d_debug_blk_trace'):
    buf = (ctypes.c_ulonglong * 8)()
    lib.t3d_debug_blk_trace(buf)
    names = ['plane load', 'weights', 'expand', 'depthwise', 'project', 'epilogue']
    print('   ' + ' | '.join(f'{n} {buf[i] * 0.01:.1f} us' for i, n in enumerate(names)))
