"""Which named buffers differ between two train-mode forwards (and backwards) of the same net on the same batch."""
import os, sys, torch
sys.path[:0] = [os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), p) for p in ('', 'tests', '3d-object-detection.pytorch_amd')]
from torchdet3d.models import engine as E
from torchdet3d import _native as N
from tests.test_gpu_engine import _loss_cfg
name, B, HW, nc = sys.argv[1], 32, 128, 9
gen = torch.Generator().manual_seed(0)
imgs, gt_kp = torch.randn(B, 3, HW, HW, generator=gen).cuda(), torch.rand(B, 9, 2, generator=gen).cuda()
cats = torch.randint(0, nc, (B,), generator=gen).cuda()
cfg = _loss_cfg(['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2]))
net = E.Net(name, nc, 'cuda', torch.bfloat16)
net.reset_parameters(seed=11)
sd = {k: v.clone() for k, v in net.state_dict().items()}
ones = torch.ones(B, 1280, device='cuda')
snaps = []
for rep in range(2):
    net.load_state_dict(sd)
    kp, lg = net.forward(imgs, cats, train=True, dropout_mask=ones)
    torch.cuda.synchronize()
    fw = {k: v.clone() for k, v in net._bufs.items()}
    out = torch.zeros(16, device='cuda')
    dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
    N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp.view(B, 18)), N.ptr(gt_kp.view(B, 18).contiguous()), N.ptr(lg), N.ptr(cats), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
    net.backward(dkp, dlg)
    torch.cuda.synchronize()
    bw = {k: v.clone() for k, v in net._bufs.items() if k not in fw or True}
    snaps.append((fw, bw, {k: v.clone() for k, v in net.g.items()}))
def diff(a, b, what):
    bad = [str(k[0]) for k in a if k in b and a[k].shape == b[k].shape and not torch.equal(a[k], b[k]) and not str(k[0]).startswith('workspace')]
    print(what, len(bad), 'differ; first:', bad[:12])
diff(snaps[0][0], snaps[1][0], 'after forward :')
diff(snaps[0][1], snaps[1][1], 'after backward:')
print('grads differing:', [k for k in snaps[0][2] if not torch.equal(snaps[0][2][k], snaps[1][2][k])][:12])
