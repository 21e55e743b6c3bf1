"""Two training runs of a few steps from the same seed: are the weights bit-identical?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '3d-object-detection.pytorch_amd'))
from torchdet3d.models import engine as E
from torchdet3d import _native as N
from tests.test_gpu_engine import _loss_cfg
name = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2'
B, HW, nc, steps = int(os.environ.get("RB", 32)), int(os.environ.get("RHW", 128)), 9, int(sys.argv[2]) if len(sys.argv) > 2 else 4
gen = torch.Generator().manual_seed(0)
imgs, gt_kp = torch.randn(B, 3, HW, HW, generator=gen).cuda(), torch.rand(B, 9, 2, generator=gen).cuda()
cats = torch.randint(0, nc, (B,), generator=gen).cuda()
cfg = _loss_cfg(['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2]))
def run():
    from torchdet3d.models.resnet import ResNetEngine
    net = (ResNetEngine if name.startswith('resnet') else E.Net)(name, nc, 'cuda', torch.bfloat16)
    net.reset_parameters(seed=11)
    m = torch.zeros_like(net.flat); v = torch.zeros_like(net.flat)
    snaps = []
    for it in range(steps):
        ones = torch.ones(B, net.arch.last_c if name == "mobilenetv2" else (2048 if name.startswith("resnet") else 1280), device="cuda")
        kp, lg = net.forward(imgs, cats, train=True, dropout_mask=ones)
        out = torch.zeros(16, device='cuda')
        dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
        N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp.view(B, 18)), N.ptr(gt_kp.view(B, 18).contiguous()), N.ptr(lg), N.ptr(cats), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
        net.backward(dkp, dlg)
        torch.cuda.synchronize()
        snaps.append((kp.clone(), out.clone(), net.gflat.clone()))
        net.flat.add_(net.gflat, alpha=-1e-2)     # plain SGD: keeps the optimizer out of the question
    return snaps, net.flat.clone()
a, fa = run(); b, fb = run()
for it, (x, y) in enumerate(zip(a, b)):
    print(it, 'kp', torch.equal(x[0], y[0]), 'loss', torch.equal(x[1], y[1]), 'grad', torch.equal(x[2], y[2]), 'max grad diff', (x[2] - y[2]).abs().max().item())
print('weights equal', torch.equal(fa, fb))
