"""Where does the host spend its time while enqueuing a training step?  (cProfile over 10 steps of the bench workload)"""
import cProfile, os, pstats, sys
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
imgs = torch.randn(B, 3, S, S, device=dev); gts = torch.rand(B, 18, device=dev); cats = torch.randint(0, 9, (B,), device=dev)
cfg = N.LossCfg(); cfg.c_l1, cfg.c_add, cfg.c_ce = 1.0, 0.1, 0.2
cfg.smoothl1_beta, cfg.wing_w, cfg.wing_eps, cfg.lam_reg, cfg.lam_cls = 0.2, 5.18, 1.0, 1.0, 1.0
out = torch.zeros(16, device=dev); dkp, dlg = torch.empty(B, 18, device=dev), torch.empty(B, 9, device=dev)
def step():
    kp, lg = net.forward(imgs, cats, train=True)
    N.call('t3d_loss_fwd_bwd', cfg, N.ptr(kp), N.ptr(gts), N.ptr(lg), N.ptr(cats), N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, 9, N.stream())
    net.backward(dkp, dlg)
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(22)
