"""Oracle of the ResNet-50 regression model (TEST INFRASTRUCTURE, see oracle/__init__.py).

BASELINE config 4 names "ResNet-50 backbone via torchdet3d.builders"; the reference has no ResNet
(`torchdet3d/builders/model_builder.py:14-17` lists its models), so the backbone is the standard
torchvision ResNet-50 (v1.5: stride on the 3x3 conv) plugged into the reference's `ModelWrapper` the way its timm /
efficientnet branches are (`model_builder.py:73-151`: global average pool -> 9 per-class heads + class head,
`output_channels = 2048`, no extra classifier).  **Parity unpinned** w.r.t. the reference.  torch-CPU fp32, functional, on a
dict with torchvision's state-dict key names + the wrapper's head keys.
"""
import torch
import torch.nn.functional as F

LAYERS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]
TINY_LAYERS = [(64, 2, 1), (128, 2, 2)]      # test-only: every block kind of the full model in 4 bottlenecks (well conditioned)
BN_EPS, BN_MOM = 1e-5, 0.1


def state_dict_shapes(num_classes=9, layers=None):
    layers = layers or LAYERS
    feat = 4 * layers[-1][0]
    out = {}

    def bn(p, c):
        out[p + '.weight'] = (c,)
        out[p + '.bias'] = (c,)
        out[p + '.running_mean'] = (c,)
        out[p + '.running_var'] = (c,)
        out[p + '.num_batches_tracked'] = ()
    out['conv1.weight'] = (64, 3, 7, 7)
    bn('bn1', 64)
    cin = 64
    for li, (w, n, s) in enumerate(layers):
        for i in range(n):
            p = f'layer{li + 1}.{i}'
            out[p + '.conv1.weight'] = (w, cin, 1, 1)
            bn(p + '.bn1', w)
            out[p + '.conv2.weight'] = (w, w, 3, 3)
            bn(p + '.bn2', w)
            out[p + '.conv3.weight'] = (4 * w, w, 1, 1)
            bn(p + '.bn3', 4 * w)
            if i == 0:
                out[p + '.downsample.0.weight'] = (4 * w, cin, 1, 1)
                bn(p + '.downsample.1', 4 * w)
            cin = 4 * w
    for k in range(9):
        out[f'regressors.{k}.0.weight'] = (18, feat)
        out[f'regressors.{k}.0.bias'] = (18,)
    out['cls_fc.1.weight'] = (num_classes, feat)
    out['cls_fc.1.bias'] = (num_classes,)
    return out


def _bn(sd, p, x, train):
    if train and (p + '.num_batches_tracked') in sd:
        sd[p + '.num_batches_tracked'] += 1
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        training=train, momentum=BN_MOM, eps=BN_EPS)


def features(sd, x, train, layers=None):
    layers = layers or LAYERS
    y = F.relu(_bn(sd, 'bn1', F.conv2d(x, sd['conv1.weight'], None, 2, 3), train))
    y = F.max_pool2d(y, 3, 2, 1)
    for li, (w, n, s) in enumerate(layers):
        for i in range(n):
            p = f'layer{li + 1}.{i}'
            st = s if i == 0 else 1
            o = F.relu(_bn(sd, p + '.bn1', F.conv2d(y, sd[p + '.conv1.weight']), train))
            o = F.relu(_bn(sd, p + '.bn2', F.conv2d(o, sd[p + '.conv2.weight'], None, st, 1), train))
            o = _bn(sd, p + '.bn3', F.conv2d(o, sd[p + '.conv3.weight']), train)
            if i == 0:
                y = _bn(sd, p + '.downsample.1', F.conv2d(y, sd[p + '.downsample.0.weight'], None, st), train)
            y = F.relu(o + y)
    return y


def forward(sd, x, cats, train=False, num_classes=9, dropout_mask=None, layers=None):
    """ModelWrapper.forward (model_builder.py:126-146) over the ResNet-50 features."""
    f = F.adaptive_avg_pool2d(features(sd, x, train, layers), 1).view(x.size(0), -1)
    kp = torch.stack([F.linear(f[b], sd[f'regressors.{int(c)}.0.weight'], sd[f'regressors.{int(c)}.0.bias'])
                      for b, c in enumerate(cats)])
    kp = torch.sigmoid(kp).view(x.size(0), 9, 2)
    if num_classes > 1:
        fd = f * dropout_mask if (train and dropout_mask is not None) else f
        return kp, F.linear(fd, sd['cls_fc.1.weight'], sd['cls_fc.1.bias'])
    return kp, cats.unsqueeze(1)
