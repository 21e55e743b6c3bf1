"""Which BatchNorm-backward coefficients differ between two backward passes on ONE saved forward (in backward order)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '3d-object-detection.pytorch_amd'))
from torchdet3d.models import engine as E
from torchdet3d import _native as N
from tests.test_gpu_engine import _loss_cfg
name, B, HW, nc = 'mobilenetv2', 32, 128, 9
gen = torch.Generator().manual_seed(0)
imgs, gt_kp = torch.randn(B, 3, HW, HW, generator=gen), torch.rand(B, 9, 2, generator=gen)
cats = torch.randint(0, nc, (B,), generator=gen)
net = E.Net(name, nc, 'cuda', torch.bfloat16)
net.reset_parameters(seed=11)
ones = torch.ones(B, net.arch.last_c, device='cuda')
kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=ones)
out = torch.zeros(16, device='cuda')
dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
cfg = _loss_cfg(['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2]))
N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp.view(B, 18)), N.ptr(gt_kp.cuda().view(B, 18).contiguous()), N.ptr(lg), N.ptr(cats.cuda()),
       N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
saved = net.saved
snaps = []
for r in range(3):
    net.saved = saved
    net._statbuf[:, net._statbuf.shape[1] // 2:].zero_()
    net.backward(dkp, dlg)
    torch.cuda.synchronize()
    snaps.append({k: (bn.alpha.clone(), bn.bbeta.clone(), bn.gammac.clone(), bn.bstats.clone()) for k, bn in net.bns.items()})
names = list(net.bns.keys())[::-1]
for k in names:
    a, b = snaps[0][k], snaps[1][k]
    c = snaps[2][k]
    eq = [bool(torch.equal(x, y)) and bool(torch.equal(x, z)) for x, y, z in zip(a, b, c)]
    rel = ((a[3] - b[3]).abs().max() / a[3].abs().max().clamp_min(1e-300)).item()
    print(f'{k:32s} alpha {eq[0]} beta {eq[1]} gamma {eq[2]} sums {eq[3]} (max rel diff of replica-0 sums {rel:.1e})')
