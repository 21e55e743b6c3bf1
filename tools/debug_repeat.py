"""Debug aid: run the same train step repeatedly and report run-to-run gradient differences."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch
from oracle.weights import make_inputs, make_state_dict
from torchdet3d.models.engine import Net
name, B, HW, nc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dtype = torch.bfloat16 if len(sys.argv) > 5 and sys.argv[5] == 'bf16' else torch.float32
sd = make_state_dict(name, nc)
imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
net = Net(name, nc, 'cuda', dtype); net.load_state_dict(sd)
mask = (torch.rand(B, net.arch.feat_c, generator=torch.Generator().manual_seed(3)) >= 0.5).float().cuda() * 2
g = torch.Generator().manual_seed(1)
dkp = torch.randn(B, 18, generator=g).cuda() * 0.01; dlg = torch.randn(B, nc, generator=g).cuda() * 0.01
ref = None
for it in range(6):
    if it % 2: net.forward(imgs.cuda(), cats.cuda(), train=False)
    kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=mask if nc > 1 else None)
    net.backward(dkp, dlg if nc > 1 else None)
    torch.cuda.synchronize()
    gf = net.gflat.clone()
    if ref is None: ref = gf
    else:
        worst = []
        for k in net.g:
            o, n = net.offsets[k]
            a, b = gf[o:o+n], ref[o:o+n]
            sc = max(b.abs().max().item(), 1e-6)
            worst.append(((a-b).abs().max().item()/sc, k))
        worst.sort(reverse=True)
        print(it, worst[:4])
