"""Where does the bf16 backward decorrelate from the fp32 one?  Per gradient buffer (backward order) cosine and relative
L2 of bf16 vs fp32 engine on the same crops; T3D_YFREE_MIN=0 disables the y-free path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch
from oracle.weights import make_inputs, make_state_dict
from torchdet3d import _native as N
from torchdet3d.models.engine import Net

B, HW, nc = int(os.environ.get('DBG_B', 64)), int(os.environ.get('DBG_HW', 224)), 9
name = os.environ.get('DBG_MODEL', 'mobilenetv2')
imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
sd = make_state_dict(name, nc)
mask = ((torch.rand(B, 1280, generator=torch.Generator().manual_seed(2)) >= 0.5).float() * 2).cuda()
cfg = N.LossCfg()
cfg.c_l1, cfg.c_add, cfg.c_ce, cfg.lam_reg, cfg.lam_cls = 1.0, 0.1, 0.2, 1.0, 1.0
cfg.smoothl1_beta, cfg.wing_w, cfg.wing_eps = 0.2, 5.18, 1.0
gtd, cd, im = gt_kp.cuda().view(B, 18).contiguous(), cats.cuda(), imgs.cuda()
runs = {}
for tag, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
    net = Net(name, nc, 'cuda', dt)
    net.load_state_dict(sd)
    if os.environ.get('DBG_ROUND_W') and dt == torch.float32:       # fp32 engine on bf16-rounded 1x1 weights
        for k, v in net.p.items():
            if v.dim() == 4 and v.shape[2] == 1:
                v.copy_(v.to(torch.bfloat16).float())
    kp, lg = net.forward(im, cd, train=True, dropout_mask=mask)
    out = torch.zeros(16, device='cuda')
    dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
    N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp), N.ptr(gtd), N.ptr(lg), N.ptr(cd), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
    net.backward(dkp, dlg)
    torch.cuda.synchronize()
    bufs = {k[0]: v.float().clone() for k, v in net._bufs.items() if isinstance(k, tuple) and k[0].split(':')[0] in ('dz', 'dz1', 'dv2', 'dzin', 'y1', 'y2', 'y3', 'z')}
    runs[tag] = (out[0].item(), {k: v.clone() for k, v in net.g.items()}, bufs)
    del net
print('loss', runs['f32'][0], runs['bf16'][0])
def cmp(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a @ b) / (a.norm() * b.norm() + 1e-300)).item(), ((a - b).norm() / (b.norm() + 1e-300)).item()
def order(k):
    t, _, i = k.partition(':')
    i = i.split(':')[0]
    return (-(int(i) if i.isdigit() else 99), {'dv2': 0, 'dz1': 1, 'dzin': 2}.get(t, 3))
print('--- activations (forward order)')
for k in sorted([k for k in runs['f32'][2] if k[0] in 'yz'], key=lambda k: (int(k.split(':')[1]) if k.split(':')[1].isdigit() else 99, k)):
    c, l2 = cmp(runs['bf16'][2][k], runs['f32'][2][k])
    print(f'  {k:12s} cos {c:.5f} relL2 {l2:.4f}')
print('--- gradient buffers (backward order)')
for k in sorted([k for k in runs['f32'][2] if k[0] == 'd'], key=order):
    c, l2 = cmp(runs['bf16'][2][k], runs['f32'][2][k])
    print(f'  {k:12s} cos {c:.5f} relL2 {l2:.4f}')
print('--- parameter gradients')
for k in runs['f32'][1]:
    c, l2 = cmp(runs['bf16'][1][k], runs['f32'][1][k])
    print(f'  {k:34s} cos {c:.4f} relL2 {l2:.3f}  |g| {runs["f32"][1][k].norm().item():.3e}')
