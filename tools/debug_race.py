"""Debug aid: run the same train step before/after an idle gap and list the buffers that differ."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch
from oracle.weights import make_inputs, make_state_dict
from torchdet3d.models.engine import Net
name, B, HW, nc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sd = make_state_dict(name, nc)
imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
net = Net(name, nc, 'cuda', torch.float32); net.load_state_dict(sd)
mask = (torch.rand(B, net.arch.feat_c, generator=torch.Generator().manual_seed(3)) >= 0.5).float().cuda() * 2
g = torch.Generator().manual_seed(1)
dkp = torch.randn(B, 18, generator=g).cuda() * 0.01; dlg = torch.randn(B, nc, generator=g).cuda() * 0.01
im, ca = imgs.cuda(), cats.cuda()
def step():
    net.forward(im, ca, train=True, dropout_mask=mask)
    net.backward(dkp, dlg)
    torch.cuda.synchronize()
    snap = {str(k): v.clone() for k, v in net._bufs.items()}
    snap['gflat'] = net.gflat.clone(); snap['stat'] = net._statbuf.clone(); snap['aff'] = net._aff.clone()
    return snap
ref = step()
for trial in range(4):
    time.sleep(float(os.environ.get('SLEEP', '12')))
    cur = step()
    bad = []
    for k in ref:
        a, b = ref[k].double(), cur[k].double()
        sc = max(a.abs().max().item(), 1e-6)
        e = (a - b).abs().max().item() / sc
        if e > 1e-3: bad.append((k, e, int(((a - b).abs() > 1e-3 * sc).sum()), a.numel()))
    print('trial', trial, 'differing buffers:', len(bad))
    for x in bad[:40]: print('   ', x)
    k = "('dz:last', (%d, 1280), torch.float32)" % (B * (HW // 32) ** 2)
    a, b = ref[k], cur[k]
    idx = ((a - b).abs() > 1e-3 * a.abs().max()).nonzero()
    bn = net.bns['conv.1']
    for m, c in idx.tolist()[:5]:
        y = ref["('y:last', (%d, 1280), torch.float32)" % (B * (HW // 32) ** 2)][m, c].item()
        print('  elem', m, c, 'dz ref/cur', a[m, c].item(), b[m, c].item(), 'y', y, 'u(now)', (bn.scale[c] * y + bn.shift[c]).item(),
              'scale', bn.scale[c].item(), 'shift', bn.shift[c].item())
