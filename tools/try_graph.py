"""Can one training step (forward + losses + backward, two streams) be captured in a HIP graph?  Replay vs eager."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd')]
import torch
from torchdet3d import _native as N
from torchdet3d.models.engine import Net
dev = torch.device('cuda', 0)
B, S = 256, 224
net = Net('mobilenetv2', 9, dev, torch.bfloat16); net.reset_parameters(seed=5)
flat = torch.nn.Parameter(net.flat); flat.grad = net.gflat
opt = torch.optim.AdamW([flat], lr=1e-3, weight_decay=1e-4, fused=True)
g = torch.Generator(device=dev).manual_seed(5)
imgs = torch.randn(B, 3, S, S, device=dev, generator=g); gts = torch.rand(B, 18, device=dev, generator=g)
cats = torch.randint(0, 9, (B,), device=dev, generator=g)
cfg = N.LossCfg(); cfg.c_l1, cfg.c_add, cfg.c_ce = 1.0, 0.1, 0.2
cfg.smoothl1_beta, cfg.wing_w, cfg.wing_eps, cfg.lam_reg, cfg.lam_cls = 0.2, 5.18, 1.0, 1.0, 1.0
out = torch.zeros(16, device=dev); dkp, dlg = torch.empty(B, 18, device=dev), torch.empty(B, 9, device=dev)
def body():
    kp, lg = net.forward(imgs, cats, train=True)
    N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp), N.ptr(gts), N.ptr(lg), N.ptr(cats), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, 9, N.stream())
    net.backward(dkp, dlg)
for _ in range(5):
    body(); opt.step()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    body()
torch.cuda.synchronize()
print('captured')
for _ in range(3):
    graph.replay(); opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    graph.replay(); opt.step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('graph: %.3f ms/step (host issue %.3f ms), loss %.5f' % ((t2 - t0) / 30 * 1e3, (t1 - t0) / 30 * 1e3, out[0].item()))
t0 = time.perf_counter()
for _ in range(30):
    body(); opt.step()
torch.cuda.synchronize(); t2 = time.perf_counter()
print('eager: %.3f ms/step, loss %.5f' % ((t2 - t0) / 30 * 1e3, out[0].item()))
