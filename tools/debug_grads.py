"""Debug aid: per-parameter gradient error of the HIP engine vs the fp64 oracle, in backward order."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch
from test_gpu_engine import _oracle_step, _loss_cfg
from oracle.weights import make_inputs, make_state_dict
from torchdet3d import _native as N
from torchdet3d.models.engine import Net

name, B, HW, nc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dtype = torch.bfloat16 if len(sys.argv) > 5 and sys.argv[5] == 'bf16' else torch.float32
lnames, coeffs = (['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2])) if nc > 1 else (['mse', 'diag_loss', 'add_loss'], ([1., .5, .1], []))
sd = make_state_dict(name, nc)
if os.environ.get('INIT') == 'ref':
    _n = Net(name, nc, 'cuda', torch.float32); _n.reset_parameters(seed=11); sd = {k: v.cpu() for k, v in _n.state_dict().items()}
imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
F = 1280 if 'small' not in name else 1024
mode = os.environ.get('DBG', '')
mask = torch.ones(B, F) if nc > 1 else None
if 'mask' in mode and nc > 1:
    mask = (torch.rand(B, F, generator=torch.Generator().manual_seed(3)) >= 0.5).float() * 2
r32 = _oracle_step(name, sd, imgs, gt_kp, cats, nc, lnames, coeffs, mask)
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
g64 = _oracle_step(name, sd64, imgs.double(), gt_kp.double(), cats, nc, lnames, coeffs, mask.double() if mask is not None else None)[3]
net = Net(name, nc, 'cuda', dtype)
net.load_state_dict(sd)
if 'eval' in mode:
    net.forward(imgs.cuda(), cats.cuda(), train=False)
    torch.cuda.synchronize()
    import time; time.sleep(float(os.environ.get('SLEEP', '0')))
kp, lg = net.forward(imgs.cuda(), cats.cuda(), train=True, dropout_mask=mask.cuda() if mask is not None else None)
print('kp err', (kp.cpu() - r32[0]).abs().max().item())
out = torch.zeros(16, device='cuda'); dkp = torch.empty(B, 18, device='cuda'); dlg = torch.empty(B, nc, device='cuda') if nc > 1 else None
gtd, cd = gt_kp.cuda().view(B, 18).contiguous(), cats.cuda()
N.call('t3d_loss_fwd_bwd', _loss_cfg(lnames, coeffs), N.ptr(kp.view(B, 18)), N.ptr(gtd), N.ptr(lg), N.ptr(cd), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
print('loss', out[0].item(), r32[2].item())
net.backward(dkp, dlg)
torch.cuda.synchronize()
for k in reversed(list(g64.keys())):
    ref = g64[k]; sc = max(ref.abs().max().item(), 1e-3)
    e = (net.g[k].cpu().double() - ref).abs().max().item() / sc
    er = (r32[3][k].double() - ref).abs().max().item() / sc
    flag = ' <<<' if e > max(2e-3, 3 * er) else ''
    a_, b_ = net.g[k].cpu().double().flatten(), ref.flatten()
    cos = ((a_ @ b_) / (a_.norm() * b_.norm() + 1e-300)).item()
    if flag or os.environ.get('ALL'): print(f'{k:34s} ours {e:9.2e}  f32-oracle {er:9.2e}  scale {sc:9.2e} cos {cos:.4f}{flag}')
