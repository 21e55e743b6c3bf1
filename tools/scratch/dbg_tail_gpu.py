import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, '3d-object-detection.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch
from test_host_logic import _cfg
from oracle.weights import make_inputs
from torchdet3d.builders import build_loss, build_model, build_optimizer
from torchdet3d.losses import LossManager
from torchdet3d.trainer import Trainer
B, HW = 256, 224
imgs, gt_kp, cats = make_inputs(B, HW, HW, 9)
imgs, gt_kp, cats = imgs.cuda(), gt_kp.cuda(), cats.cuda()
cfg = _cfg('mobilenetv2'); cfg.model.storage_dtype = 'bf16'
m = build_model(cfg); m.to('cuda'); m.train()
opt = build_optimizer(cfg, m)
lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
tr = Trainer(m, None, opt, None, lm, None, 1, '', device='cuda', save_chkpt=False)
tr.overlap_tail = '--plain' not in sys.argv
net = m.net
EV = []   # (tag, step, event)
cur = [0]
def ev(tag, stream=None):
    e = torch.cuda.Event(enable_timing=True); e.record(stream) if stream is not None else e.record(); EV.append((tag, cur[0], e))
ow = net._wait_late
def wl():
    pending = net._late_event is not None
    if pending: ev('wait_pre')
    ow()
    if pending: ev('wait_post')
net._wait_late = wl
orl = net.run_late
def rl(tail=None):
    ev('tail_issue_main')
    orl(tail)
    ev('tail_end_side', net._side)
net.run_late = rl
of, ob = net.forward, net.backward
def fw(*a, **k):
    ev('fwd_start'); r = of(*a, **k); ev('fwd_end'); return r
def bw(*a, **k):
    ev('bwd_start'); r = ob(*a, **k); ev('bwd_end'); return r
net.forward, net.backward = fw, bw
base = torch.cuda.Event(enable_timing=True)
for it in range(40):
    cur[0] = it
    if it == 10: torch.cuda.synchronize(); base.record()
    tr.train_step(imgs, gt_kp, cats, it)
    ev('step_end')
torch.cuda.synchronize()
import collections
by = collections.defaultdict(dict)
for tag, st, e in EV:
    if st >= 12: by[st][tag] = base.elapsed_time(e)
rel = collections.defaultdict(list)
for st in sorted(by):
    d = by[st]; t0 = d['fwd_start']
    for k, v in d.items(): rel[k].append(v - t0)
    if st + 1 in by: rel['next_fwd_start'].append(by[st + 1]['fwd_start'] - t0)
for k in sorted(rel, key=lambda k: sum(rel[k]) / len(rel[k])):
    print('%-18s %8.3f ms after fwd_start (mean of %d)' % (k, sum(rel[k]) / len(rel[k]), len(rel[k])))
