import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch
from test_host_logic import _cfg
from oracle.weights import make_inputs
from torchdet3d.builders import build_loss, build_model, build_optimizer
from torchdet3d.losses import LossManager
from torchdet3d.trainer import Trainer
from torchdet3d.models import engine as E
B, HW = 256, 224
imgs, gt_kp, cats = make_inputs(B, HW, HW, 9)
imgs, gt_kp, cats = imgs.cuda(), gt_kp.cuda(), cats.cuda()
cfg = _cfg('mobilenetv2'); cfg.model.storage_dtype = 'bf16'
m = build_model(cfg); m.to('cuda'); m.train()
opt = build_optimizer(cfg, m)
lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
tr = Trainer(m, None, opt, None, lm, None, 1, '', device='cuda', save_chkpt=False)
tr.overlap_tail = '--plain' not in sys.argv
T = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
wrap(m.net, 'run_late'); wrap(m.net, 'forward'); wrap(m.net, 'backward'); wrap(opt, 'step')
for it in range(30):
    t0 = time.perf_counter(); tr.train_step(imgs, gt_kp, cats, it); T.setdefault('train_step', []).append(time.perf_counter() - t0)
torch.cuda.synchronize()
for k, v in T.items():
    v = v[10:]
    print(k, 'n', len(v), 'mean ms %.3f' % (1e3 * sum(v) / max(1, len(v))), 'max %.3f' % (1e3 * max(v) if v else 0))
