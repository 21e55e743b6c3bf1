import os, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/3d-object-detection.pytorch_amd']
sys.path[:0] = [os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), p) for p in ('', 'tests', '3d-object-detection.pytorch_amd')]
from test_host_logic import _cfg
from oracle.weights import make_inputs, make_state_dict
from torchdet3d.builders import build_loss, build_model, build_optimizer
from torchdet3d.losses import LossManager
name, B, HW = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
imgs, gt_kp, cats = make_inputs(B, HW, HW, 9)
imgs, gt_kp, cats = imgs.cuda(), gt_kp.cuda(), cats.cuda()
cfg = _cfg(name); cfg.model.storage_dtype = 'bf16'
sd = make_state_dict(name, 9) if name != 'resnet50' else None
def run():
    torch.manual_seed(3)
    m = build_model(cfg)
    if sd is not None: m.load_state_dict(sd)
    m.to('cuda'); m.train()
    opt = build_optimizer(cfg, m)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    gen = torch.Generator(device='cuda').manual_seed(5)
    mask = (torch.rand(B, m.net.arch.classifier or m.net.arch.last_c, device='cuda', generator=gen) > 0.2).float() * 1.25
    kp, tg = m(imgs, cats, dropout_mask=mask)
    loss = lm.parse_losses(kp, gt_kp, tg, cats, 0)
    opt.zero_grad(); loss.backward(); torch.cuda.synchronize()
    return {k: v.clone() for k, v in m.net.g.items()}, kp.detach().clone(), loss.detach().clone()
a, b = run(), run()
print('kp', torch.equal(a[1], b[1]), 'loss', torch.equal(a[2], b[2]))
print('differing:', [k for k in a[0] if not torch.equal(a[0][k], b[0][k])])
