import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[ROOT, ROOT+'/3d-object-detection.pytorch_amd', ROOT+'/tests']
import torch, numpy as np
from oracle.weights import make_inputs, make_state_dict
from torchdet3d.models import resnet as RM
from torchdet3d import _native as N
from test_host_logic import _cfg
from torchdet3d.builders import build_loss, build_model, build_optimizer
from torchdet3d.losses import LossManager
def api():
    cfg = _cfg('resnet50'); cfg.model.storage_dtype = 'bf16'
    m = build_model(cfg).to('cuda')
    opt = build_optimizer(cfg, m)
    lm = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
    imgs, gt_kp, cats = make_inputs(16, 128, 128, 9)
    m.train()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    for it in range(steps):
        kp, tg = m(imgs.cuda(), cats.cuda())
        loss = lm.parse_losses(kp, gt_kp.cuda(), tg, cats.cuda(), it)
        opt.zero_grad(); loss.backward(); opt.step()
    if '--eval' in sys.argv:
        m.eval()
        with torch.no_grad():
            m(imgs.cuda(), cats.cuda())
api()
import gc; gc.collect(); torch.cuda.synchronize()
print('fold pending', N.lib().t3d_fold_pending())
B,HW,nc=8,96,9
imgs,gt,cats=make_inputs(B,HW,HW,nc)
sd=make_state_dict('resnet50',nc)
mask=torch.full((B,2048),2.0).cuda()
nets={}
for on in (False, True):
    RM.IMPLICIT3=on
    net=RM.ResNetEngine('resnet50',nc,'cuda',torch.bfloat16)
    net.load_state_dict(sd)
    kp,lg=net.forward(imgs.cuda(),cats.cuda(),train=True,dropout_mask=mask)
    if '--bwd' in sys.argv:
        from test_gpu_engine import _loss_cfg
        cfgl = _loss_cfg(['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2]))
        out = torch.zeros(16, device='cuda')
        dkp, dlg = torch.empty(B, 18, device='cuda'), torch.empty(B, nc, device='cuda')
        N.call('t3d_loss_fwd_bwd', cfgl, N.ptr(kp), N.ptr(gt.cuda().view(B, 18).contiguous()), N.ptr(lg), N.ptr(cats.cuda()),
               N.ptr(out), N.ptr(dkp), N.ptr(dlg), B, nc, N.stream())
        kpc = kp.clone()
        net.backward(dkp, dlg)
        torch.cuda.synchronize()
        print('on', on, 'loss', out[0].item())
        net.saved = {'kp': kpc}
    torch.cuda.synchronize()
    nets[on]=net
    if '--delnet' in sys.argv:
        nets[on] = None
        del net
        import gc; gc.collect()
if '--delnet' in sys.argv: sys.exit(0)
a,b=nets[False],nets[True]
da={k[0]:t for k,t in a._bufs.items() if isinstance(k,tuple)}
db={k[0]:t for k,t in b._bufs.items() if isinstance(k,tuple)}
n=0
for k in da:
    if k.startswith(('y1:','y2:','y3:','z:','wc:','wcf:')) and k in db and da[k].shape==db[k].shape:
        d=(da[k].float()-db[k].float()).abs().max().item()
        if d>0 and n<6: print(k, d, da[k].float().abs().max().item()); n+=1
print('kp diff', (a.saved['kp']-b.saved['kp']).abs().max().item())
