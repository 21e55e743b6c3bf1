import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch
from test_host_logic import _cfg
from oracle.weights import make_inputs, make_state_dict
from torchdet3d.builders import build_loss, build_model, build_optimizer
from torchdet3d.losses import LossManager
from torchdet3d.trainer import Trainer
B, HW = 64, 224
imgs, gt_kp, cats = make_inputs(B, HW, HW, 9)
imgs, gt_kp, cats = imgs.cuda(), gt_kp.cuda(), cats.cuda()
cfg = _cfg('mobilenetv2'); cfg.model.storage_dtype = 'bf16'
sd = make_state_dict('mobilenetv2', 9)
def run(overlap, sync_each, do_eval):
    torch.manual_seed(3)
    m = build_model(cfg); m.load_state_dict(sd); m.to('cuda'); m.train()
    opt = build_optimizer(cfg, m)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    tr = Trainer(m, None, opt, None, lm, None, 1, '', device='cuda', save_chkpt=False)
    tr.overlap_tail = overlap
    ws = []
    for it in range(5):
        r = tr.train_step(imgs, gt_kp, cats, it)
        if sync_each:
            tr.join_tail(); torch.cuda.synchronize()
            ws.append(m.net.flat.clone())
        if do_eval and it == 1:
            m.eval()
            with torch.no_grad(): m(imgs[:8], cats[:8])
            m.train()
    tr.join_tail(); torch.cuda.synchronize()
    ws.append(m.net.flat.clone())
    return ws, m.net
for sync_each in (True, False):
    for do_eval in (False, True):
        a, na = run(False, sync_each, do_eval)
        b, nb = run(True, sync_each, do_eval)
        lo = nb._late_lo
        print('sync_each', sync_each, 'eval', do_eval, 'late_lo', lo, 'of', nb.flat.numel(),
              [(float((x - y).abs().max()), float((x[:lo] - y[:lo]).abs().max()), float((x[lo:] - y[lo:]).abs().max())) for x, y in zip(a, b)])
