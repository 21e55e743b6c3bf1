"""Oracle model: functional torch-CPU fp32 restatement of the reference forward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Operates on a plain
``dict[str, Tensor]`` with the reference's state-dict key names, so the very
same dict can be loaded into the real reference (`gen_golden.py`) and into the
product model.

Follows:
  * torchdet3d/models/mobilenetv3.py:74-89   h_sigmoid / h_swish
  * torchdet3d/models/mobilenetv3.py:92-107  SELayer
  * torchdet3d/models/mobilenetv3.py:126-166 InvertedResidual (both layouts, SE
    before the activation in the expand layout, after it in the other)
  * torchdet3d/models/mobilenetv3.py:169-203 MobileNetV3 / extract_features
  * torchdet3d/builders/model_builder.py:73-151 ModelWrapper (heads, forward,
    forward_to_onnx)
BatchNorm uses the PyTorch defaults the reference relies on (eps 1e-5,
momentum 0.1, biased variance to normalise, unbiased for the running estimate).
"""
import torch
import torch.nn.functional as F

from .specs import arch

BN_EPS = 1e-5
BN_MOM = 0.1


def act_fn(x, kind):
    if kind == 'relu':
        return F.relu(x)
    if kind == 'relu6':
        return F.relu6(x)
    if kind == 'hswish':                      # mobilenetv3.py:83-89
        return x * (F.relu6(x + 3.) / 6.)
    if kind == 'none':
        return x
    raise AssertionError(kind)


def hsigmoid(x):                              # mobilenetv3.py:74-80
    return F.relu6(x + 3.) / 6.


def _bn(sd, prefix, x, train):
    rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    if train and (prefix + '.num_batches_tracked') in sd:
        sd[prefix + '.num_batches_tracked'] += 1
    return F.batch_norm(x, rm, rv, sd[prefix + '.weight'], sd[prefix + '.bias'],
                        training=train, momentum=BN_MOM, eps=BN_EPS)


def _se(sd, prefix, x):
    b, c = x.shape[:2]
    y = x.mean(dim=(2, 3))
    y = F.relu(F.linear(y, sd[prefix + '.fc.0.weight'], sd[prefix + '.fc.0.bias']))
    y = hsigmoid(F.linear(y, sd[prefix + '.fc.2.weight'], sd[prefix + '.fc.2.bias']))
    return x * y.view(b, c, 1, 1)


def block_forward(sd, p, blk, x, train, taps=None):
    """One InvertedResidual (mobilenetv3.py:126-166)."""
    k, s, act = blk['k'], blk['s'], blk['act']
    inp = x
    if blk['cin'] == blk['cexp']:             # mobilenetv3.py:133-144
        y = F.conv2d(x, sd[p + '.conv.0.weight'], None, s, (k - 1) // 2, 1, blk['cexp'])
        y = _bn(sd, p + '.conv.1', y, train)
        y = act_fn(y, act)
        if blk['se']:
            y = _se(sd, p + '.conv.3', y)
        y = F.conv2d(y, sd[p + '.conv.4.weight'])
        y = _bn(sd, p + '.conv.5', y, train)
    else:                                     # mobilenetv3.py:146-160
        y = F.conv2d(x, sd[p + '.conv.0.weight'])
        y = _bn(sd, p + '.conv.1', y, train)
        y = act_fn(y, act)
        y = F.conv2d(y, sd[p + '.conv.3.weight'], None, s, (k - 1) // 2, 1, blk['cexp'])
        y = _bn(sd, p + '.conv.4', y, train)
        if blk['se']:
            y = _se(sd, p + '.conv.5', y)
        y = act_fn(y, act)
        y = F.conv2d(y, sd[p + '.conv.7.weight'])
        y = _bn(sd, p + '.conv.8', y, train)
    return inp + y if blk['res'] else y       # mobilenetv3.py:162-166


def extract_features(sd, a, x, train, taps=None):
    """mobilenetv3.py:199-203 (features + conv)."""
    y = F.conv2d(x, sd['features.0.0.weight'], None, 2, 1)          # :110-115
    y = act_fn(_bn(sd, 'features.0.1', y, train), a['stem_act'])
    if taps is not None:
        taps['features.0'] = y
    for i, blk in enumerate(a['blocks']):
        y = block_forward(sd, f'features.{i + 1}', blk, y, train)
        if taps is not None:
            taps[f'features.{i + 1}'] = y
    y = F.conv2d(y, sd['conv.0.weight'])                            # :118-123,188
    y = act_fn(_bn(sd, 'conv.1', y, train), a['last_act'])
    if taps is not None:
        taps['conv'] = y
    return y


def pooled_features(sd, a, x, train, taps=None):
    f = extract_features(sd, a, x, train, taps)
    f = F.adaptive_avg_pool2d(f, 1).view(x.size(0), -1)             # model_builder.py:96-110
    if a['classifier']:                                             # model_builder.py:130-131
        f = F.linear(f, sd['classifier.0.weight'], sd['classifier.0.bias'])
        f = act_fn(_bn(sd, 'classifier.1', f, train), 'hswish')     # mobilenetv3.py:191-195
    if taps is not None:
        taps['pooled'] = f
    return f


def forward(sd, name, x, cats, train=False, num_classes=9, dropout_mask=None, taps=None):
    """ModelWrapper.forward (model_builder.py:126-146).

    ``dropout_mask``: [B, feat_c] tensor of {0, 2} applied in place of
    nn.Dropout(0.5) (model_builder.py:83) when train=True; None in train mode
    means "no dropout" (mask of ones) so that goldens are RNG-free.
    """
    a = arch(name)
    f = pooled_features(sd, a, x, train, taps)
    kp = torch.stack([F.linear(f[b], sd[f'regressors.{int(c)}.0.weight'],
                               sd[f'regressors.{int(c)}.0.bias'])
                      for b, c in enumerate(cats)])                 # :137
    kp = torch.sigmoid(kp).view(x.size(0), 9, 2)                    # :138-139
    if num_classes > 1:
        fd = f * dropout_mask if (train and dropout_mask is not None) else f
        targets = F.linear(fd, sd['cls_fc.1.weight'], sd['cls_fc.1.bias'])  # :142
    else:
        targets = cats.unsqueeze(1)                                 # :144
    return kp, targets


def forward_to_onnx(sd, name, x, num_classes=9):
    """ModelWrapper.forward_to_onnx (model_builder.py:112-124): all 9 heads, eval."""
    a = arch(name)
    f = pooled_features(sd, a, x, False)
    outs = [F.linear(f, sd[f'regressors.{k}.0.weight'], sd[f'regressors.{k}.0.bias'])
            .view(1, x.size(0), 9, 2) for k in range(9)]
    kp = torch.sigmoid(torch.cat(outs))
    tg = F.linear(f, sd['cls_fc.1.weight'], sd['cls_fc.1.bias']) if num_classes > 1 \
        else torch.zeros(x.size(0))
    return kp, tg


def state_dict_shapes(name, num_classes=9):
    """Ordered {key: shape} with the reference's state-dict names."""
    if name in ('resnet50', 'resnet14'):
        from .resnet import TINY_LAYERS, state_dict_shapes as rs
        return rs(num_classes, TINY_LAYERS if name == 'resnet14' else None)
    a = arch(name)
    out = {}

    def bn(p, c):
        out[p + '.weight'] = (c,)
        out[p + '.bias'] = (c,)
        out[p + '.running_mean'] = (c,)
        out[p + '.running_var'] = (c,)
        out[p + '.num_batches_tracked'] = ()

    def se(p, c, h):
        out[p + '.fc.0.weight'] = (h, c)
        out[p + '.fc.0.bias'] = (h,)
        out[p + '.fc.2.weight'] = (c, h)
        out[p + '.fc.2.bias'] = (c,)

    out['features.0.0.weight'] = (a['stem_c'], 3, 3, 3)
    bn('features.0.1', a['stem_c'])
    for i, b in enumerate(a['blocks']):
        p = f'features.{i + 1}.conv'
        k = b['k']
        if b['cin'] == b['cexp']:
            out[p + '.0.weight'] = (b['cexp'], 1, k, k)
            bn(p + '.1', b['cexp'])
            if b['se']:
                se(p + '.3', b['cexp'], b['se'])
            out[p + '.4.weight'] = (b['cout'], b['cexp'], 1, 1)
            bn(p + '.5', b['cout'])
        else:
            out[p + '.0.weight'] = (b['cexp'], b['cin'], 1, 1)
            bn(p + '.1', b['cexp'])
            out[p + '.3.weight'] = (b['cexp'], 1, k, k)
            bn(p + '.4', b['cexp'])
            if b['se']:
                se(p + '.5', b['cexp'], b['se'])
            out[p + '.7.weight'] = (b['cout'], b['cexp'], 1, 1)
            bn(p + '.8', b['cout'])
    out['conv.0.weight'] = (a['last_c'], a['blocks'][-1]['cout'], 1, 1)
    bn('conv.1', a['last_c'])
    if a['classifier']:
        out['classifier.0.weight'] = (a['classifier'], a['last_c'])
        out['classifier.0.bias'] = (a['classifier'],)
        bn('classifier.1', a['classifier'])
    for k in range(9):                                              # always 9 heads, :78-81
        out[f'regressors.{k}.0.weight'] = (18, a['feat_c'])
        out[f'regressors.{k}.0.bias'] = (18,)
    out['cls_fc.1.weight'] = (num_classes, a['feat_c'])
    out['cls_fc.1.bias'] = (num_classes,)
    return out
