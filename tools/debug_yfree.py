"""Engine gradients with the y-free expand backward on / off, and run-to-run (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch
from oracle.weights import make_inputs
from torchdet3d.models import engine as E
from torchdet3d import _native as N
name, B, HW, nc = 'mobilenetv2', 16, 96, 9
imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
def run(thr):
    E.YFREE_MIN_ELEMS = thr
    net = E.Net(name, nc, 'cuda', torch.bfloat16); net.reset_parameters(seed=11)
    ones = torch.ones(B, 1280, device='cuda')
    kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=ones)
    out = torch.zeros(16, device='cuda'); dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
    cfg = N.LossCfg(); cfg.c_l1, cfg.c_add, cfg.c_ce = 1.0, 0.1, 0.2
    cfg.smoothl1_beta, cfg.wing_w, cfg.wing_eps, cfg.lam_reg, cfg.lam_cls = 0.2, 5.18, 1.0, 1.0, 1.0
    gtd, cd = gt_kp.cuda().view(B, 18).contiguous(), cats.cuda()
    N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp.view(B, 18)), N.ptr(gtd), N.ptr(lg), N.ptr(cd), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
    net.backward(dkp, dlg); torch.cuda.synchronize()
    return {k: v.detach().float().cpu().clone() for k, v in net.g.items()}
def cmp(a, b, tag):
    rows = []
    for k in a:
        n = max(b[k].norm().item(), 1e-3 * b[k].numel() ** .5)
        rows.append(((a[k] - b[k]).norm().item() / n, k, b[k].norm().item()))
    rows.sort(reverse=True)
    print(tag, [(round(r, 4), k, round(nn, 5)) for r, k, nn in rows[:6]])
y1, y2, r1, r2 = run(1), run(1), run(0), run(0)
cmp(y1, y2, 'yfree vs yfree:'); cmp(r1, r2, 'regular vs regular:'); cmp(y1, r1, 'yfree vs regular:')
