import os, sys, torch
sys.path[:0] = [os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), p) for p in ('', '3d-object-detection.pytorch_amd')]
from torchdet3d import _native as N
B, F, nc = 256, 1280, 9
g = torch.Generator(device='cuda').manual_seed(0)
f = torch.randn(B, F, device='cuda', generator=g)
cats = torch.randint(0, 9, (B,), device='cuda')
mask = torch.ones(B, F, device='cuda')
dpre = torch.randn(B, 18, device='cuda', generator=g)
dlog = torch.randn(B, nc, device='cuda', generator=g)
sc, sh = torch.rand(F, device='cuda') + .5, torch.randn(F, device='cuda') * .1
pro = N.prologue(sc, sh, None, 'relu6', False)
dwreg, dbreg = torch.zeros(9 * 18, F, device='cuda'), torch.zeros(9 * 18, device='cuda')
dwcls, dbcls = torch.zeros(nc, F, device='cuda'), torch.zeros(nc, device='cuda')
def run():
    N.call('t3d_head_bwd_weights', N.ptr(f), pro, N.ptr(cats), N.ptr(mask), N.ptr(dpre), N.ptr(dlog), N.ptr(dwreg), N.ptr(dbreg),
           N.ptr(dwcls), N.ptr(dbcls), B, F, nc, N.stream())
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print('head_bwd_weights %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
