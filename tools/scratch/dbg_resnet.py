import os, sys
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'3d-object-detection.pytorch_amd'), os.path.join(ROOT,'tests')]
import torch, numpy as np
from oracle import losses as OL, resnet as R
from oracle.weights import make_inputs, make_state_dict
from test_gpu_engine import _loss_cfg
from torchdet3d import _native as N
from torchdet3d.models.resnet import ResNetEngine
name='resnet14'; B,HW,nc=16,96,9
imgs, gt_kp, cats = make_inputs(B,HW,HW,nc)
sd = make_state_dict(name, nc)
net = ResNetEngine(name, nc, 'cuda', torch.float32); net.load_state_dict(sd)
mask = torch.full((B,512),2.0)
params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k,v in sd.items()}
# oracle with hooks on intermediates of the top block
import torch.nn.functional as F
inter={}
def feats(sd,x):
    y = F.relu(R._bn(sd,'bn1',F.conv2d(x,sd['conv1.weight'],None,2,3),True)); y=F.max_pool2d(y,3,2,1)
    for li,(w,n,s) in enumerate(R.TINY_LAYERS):
        for i in range(n):
            p=f'layer{li+1}.{i}'; st=s if i==0 else 1
            y1=F.conv2d(y,sd[p+'.conv1.weight']); y1.retain_grad(); inter[p+'.y1']=y1
            u1=R._bn(sd,p+'.bn1',y1,True); u1.retain_grad(); inter[p+'.u1']=u1
            o=F.relu(u1)
            o=F.relu(R._bn(sd,p+'.bn2',F.conv2d(o,sd[p+'.conv2.weight'],None,st,1),True))
            o=R._bn(sd,p+'.bn3',F.conv2d(o,sd[p+'.conv3.weight']),True)
            if i==0: y=R._bn(sd,p+'.downsample.1',F.conv2d(y,sd[p+'.downsample.0.weight'],None,st),True)
            y=F.relu(o+y)
    return y
f=F.adaptive_avg_pool2d(feats(params,imgs),1).view(B,-1)
kp_o=torch.sigmoid(torch.stack([F.linear(f[b],params[f'regressors.{int(c)}.0.weight'],params[f'regressors.{int(c)}.0.bias']) for b,c in enumerate(cats)])).view(B,9,2)
lg_o=F.linear(f*mask,params['cls_fc.1.weight'],params['cls_fc.1.bias'])
lm=OL.LossManager(OL.build(['l1','add_loss','cross_entropy']),([1.,.1],[.2]))
lm.parse_losses(kp_o,gt_kp,lg_o,cats,0).backward()
kp,lg=net.forward(imgs.cuda(),cats.cuda(),train=True,dropout_mask=mask.cuda())
out=torch.zeros(16,device='cuda'); dkp,dlg=torch.empty(B,18,device='cuda'),torch.empty(B,nc,device='cuda')
gtd,cd=gt_kp.cuda().view(B,18).contiguous(),cats.cuda()
N.call('t3d_loss_fwd_bwd',_loss_cfg(['l1','add_loss','cross_entropy'],([1.,.1],[.2])),N.ptr(kp),N.ptr(gtd),N.ptr(lg),N.ptr(cd),N.ptr(out),N.ptr(dkp),N.ptr(dlg),B,nc,N.stream())
net.backward(dkp,dlg); torch.cuda.synchronize()
p='layer2.1'
d1=[t for k,t in net._bufs.items() if isinstance(k,tuple) and k[0]=='d1:'+p][0].cpu()
ref_d1=inter[p+'.u1'].grad.permute(0,2,3,1).reshape(-1,128)      # gradient at BN1 output
print('d1 max abs err', (d1-ref_d1).abs().max().item(), 'ref max', ref_d1.abs().max().item())
print('sum d1 rel err', ((d1.sum(0)-ref_d1.sum(0)).norm()/ref_d1.sum(0).norm()).item())
print('dbeta engine vs sum(d1 engine)', ((net.g[p+'.bn1.bias'].cpu()-d1.sum(0)).norm()/d1.sum(0).norm()).item())
print('dbeta oracle vs sum(ref d1)', ((params[p+'.bn1.bias'].grad-ref_d1.sum(0)).norm()/ref_d1.sum(0).norm()).item())
print('norms: ||dbeta||', params[p+'.bn1.bias'].grad.norm().item(), ' ||d1||', ref_d1.norm().item(), 'sqrt(n)*', (ref_d1.abs().mean()*ref_d1.shape[0]).item())
