"""Repeat the backward pass on ONE saved forward and print how much the gradients move between repeats (should be the
atomics' summation-order noise only).  usage: python tools/debug_backward_determinism.py [model] [B] [HW] [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '3d-object-detection.pytorch_amd'))
from torchdet3d.models import engine as E          # noqa: E402
from torchdet3d import _native as N                # noqa: E402
from tests.test_gpu_engine import _loss_cfg        # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 96
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
nc = 9
gen = torch.Generator().manual_seed(0)
imgs, gt_kp = torch.randn(B, 3, HW, HW, generator=gen), torch.rand(B, 9, 2, generator=gen)
cats = torch.randint(0, nc, (B,), generator=gen)
net = E.Net(name, nc, 'cuda', torch.float32 if os.environ.get('DBG_F32') else torch.bfloat16)
net.reset_parameters(seed=11)
ones = torch.ones(B, net.arch.last_c, device='cuda')
kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=ones)
out = torch.zeros(16, device='cuda')
dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
cfg = _loss_cfg(['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2]))
N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp.view(B, 18)), N.ptr(gt_kp.cuda().view(B, 18).contiguous()), N.ptr(lg), N.ptr(cats.cuda()),
       N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
saved, first = net.saved, None
for r in range(reps):
    net.saved = saved
    net._statbuf[:, net._statbuf.shape[1] // 2:].zero_()
    net.backward(dkp, dlg)
    torch.cuda.synchronize()
    g = {k: v.detach().double().cpu().clone() for k, v in net.g.items()}
    if first is None:
        first = g
        rms = {k: v.norm().item() / v.numel() ** .5 for k, v in g.items()}
        med = sorted(rms.values())[len(rms) // 2]
        continue
    rows = sorted(((g[k] - first[k]).norm().item() / first[k].norm().item(), k) for k in g if rms[k] >= 1e-3 * med)
    nz = sum(1 for v, k in rows if v > 0)
    tot = sum((g[k] - first[k]).norm().item() ** 2 for k in g) ** .5 / sum(first[k].norm().item() ** 2 for k in g) ** .5
    print(r, f'total {tot:.2e} differing {nz}/{len(rows)} worst', [(f'{v:.2e}', k) for v, k in rows[-3:]], flush=True)
    if os.environ.get('DBG_NAMES'):
        print('   differing:', sorted(k for k in g if not torch.equal(g[k], first[k])))
    if os.environ.get('DBG_LIST') and tot > 1e-3:
        for k in g:
            d = (g[k] - first[k]).norm().item() / max(first[k].norm().item(), 1e-30)
            print(f'   {k:40s} {d:.2e} rms/med {rms[k] / med:.1e}')
        break
