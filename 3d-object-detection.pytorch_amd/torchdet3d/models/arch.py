"""Backbone tables of the regression models.

mobilenetv3_{large,small}: the (k, t, c, SE, HS, s) rows, channel rounding and constructor loop
of the reference (torchdet3d/models/mobilenetv3.py:20-52 `model_params`, :54-71
`_make_divisible`, :174-188).  mobilenetv2: the standard MobileNetV2-1.0 (t, c, n, s) table
(the north-star throughput model; the reference only names it as the detector backbone,
configs/detection/mnv2_ssd_300_2_heads.py:8).
"""

MOBILENETV3 = {
    'mobilenetv3_large': dict(rows=[
        (3, 1, 16, 0, 0, 1), (3, 4, 24, 0, 0, 2), (3, 3, 24, 0, 0, 1), (5, 3, 40, 1, 0, 2), (5, 3, 40, 1, 0, 1),
        (5, 3, 40, 1, 0, 1), (3, 6, 80, 0, 1, 2), (3, 2.5, 80, 0, 1, 1), (3, 2.3, 80, 0, 1, 1),
        (3, 2.3, 80, 0, 1, 1), (3, 6, 112, 1, 1, 1), (3, 6, 112, 1, 1, 1), (5, 6, 160, 1, 1, 2),
        (5, 6, 160, 1, 1, 1), (5, 6, 160, 1, 1, 1)], feat=1280),
    'mobilenetv3_small': dict(rows=[
        (3, 1, 16, 1, 0, 2), (3, 4.5, 24, 0, 0, 2), (3, 3.67, 24, 0, 0, 1), (5, 4, 40, 1, 1, 2), (5, 6, 40, 1, 1, 1),
        (5, 6, 40, 1, 1, 1), (5, 3, 48, 1, 1, 1), (5, 3, 48, 1, 1, 1), (5, 6, 96, 1, 1, 2), (5, 6, 96, 1, 1, 1),
        (5, 6, 96, 1, 1, 1)], feat=1024),
}
MOBILENETV2 = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2),
               (6, 320, 1, 1)]
AVAILABLE_MODELS = ('mobilenetv2', 'mobilenetv3_large', 'mobilenetv3_small', 'resnet50')
# test-only name (not buildable through build_model): the reference's own `MobileNetV3(cfgs, mode='large')` class
# (mobilenetv3.py:169-197) instantiated with MobileNetV2's (t, c, n, s) table as its rows (k=3, SE=0, HS=0) -- every
# depthwise / pointwise layer from 112x112x96 on has exactly the headline model's shape, so the golden fixture generated
# from the REAL reference (oracle/gen_golden.py, tests/golden/mnv2rows_b32_224.npz) pins the benchmarked kernels' production
# shapes to the reference instead of to the oracle's restatement of MobileNetV2
MOBILENETV3['mobilenetv3_mnv2rows'] = dict(
    rows=[(3, t, c, 0, 0, s if i == 0 else 1) for t, c, n, s in MOBILENETV2 for i in range(n)], feat=1280)
TEST_ONLY_MODELS = ('resnet14', 'mobilenetv3_mnv2rows')
RESNET50_LAYERS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]      # (width, blocks, stride of the first block)


def make_divisible(v, divisor=8, min_value=None):
    """mobilenetv3.py:54-71."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class Block:
    """One InvertedResidual (mobilenetv3.py:126-166)."""

    def __init__(self, cin, cexp, cout, k, s, se, act):
        self.cin, self.cexp, self.cout, self.k, self.s, self.se, self.act = cin, cexp, cout, k, s, se, act
        self.expand = cin != cexp            # :133 vs :146 layouts
        self.res = (s == 1 and cin == cout)  # :131


class Arch:
    def __init__(self, name):
        assert name in AVAILABLE_MODELS or name in TEST_ONLY_MODELS, f'unknown model {name}'
        self.name = name
        self.blocks = []
        self.kind = 'resnet' if name in ('resnet50', 'resnet14') else 'mobilenet'
        if name == 'resnet14':
            # test-only: the four bottleneck kinds of ResNet-50 (projection / identity shortcut, stride 1 / 2) in a network
            # shallow enough for gradient-level comparisons (tests/test_gpu_resnet.py)
            self.layers = [(64, 2, 1), (128, 2, 2)]
            self.stem_c, self.stem_act = 64, 'relu'
            self.last_c, self.last_act, self.classifier, self.feat_c = 512, 'relu', 0, 512
        elif name == 'resnet50':
            # standard torchvision ResNet-50 (v1.5): BASELINE config 4; no reference source
            self.layers = RESNET50_LAYERS
            self.stem_c, self.stem_act = 64, 'relu'
            self.last_c, self.last_act, self.classifier, self.feat_c = 2048, 'relu', 0, 2048
        elif name == 'mobilenetv2':
            self.stem_c, self.stem_act = 32, 'relu6'
            cin = 32
            for t, c, n, s in MOBILENETV2:
                for i in range(n):
                    self.blocks.append(Block(cin, cin * t, c, 3, s if i == 0 else 1, 0, 'relu6'))
                    cin = c
            self.last_c, self.last_act, self.classifier, self.feat_c = 1280, 'relu6', 0, 1280
        else:
            spec = MOBILENETV3[name]
            self.stem_c, self.stem_act = make_divisible(16), 'hswish'
            cin = self.stem_c
            cexp = cin
            for k, t, c, se, hs, s in spec['rows']:
                cout = make_divisible(c)
                cexp = make_divisible(cin * t)
                self.blocks.append(Block(cin, cexp, cout, k, s, make_divisible(cexp // 4) if se else 0,
                                         'hswish' if hs else 'relu'))
                cin = cout
            self.last_c, self.last_act = cexp, 'hswish'
            self.classifier = self.feat_c = spec['feat']

    def param_shapes(self, num_classes):
        """Ordered {state-dict key: (shape, kind)}; kind in param | buffer.  Key names and order are the
        reference's (`state_dict()` of ModelWrapper(MobileNetV3))."""
        out = {}
        if self.kind == 'resnet':
            return self._resnet_shapes(num_classes)

        def bn(p, c):
            out[p + '.weight'] = ((c,), 'param')
            out[p + '.bias'] = ((c,), 'param')
            out[p + '.running_mean'] = ((c,), 'buffer')
            out[p + '.running_var'] = ((c,), 'buffer')
            out[p + '.num_batches_tracked'] = ((), 'buffer')

        def se(p, c, h):
            out[p + '.fc.0.weight'] = ((h, c), 'param')
            out[p + '.fc.0.bias'] = ((h,), 'param')
            out[p + '.fc.2.weight'] = ((c, h), 'param')
            out[p + '.fc.2.bias'] = ((c,), 'param')

        out['features.0.0.weight'] = ((self.stem_c, 3, 3, 3), 'param')
        bn('features.0.1', self.stem_c)
        for i, b in enumerate(self.blocks):
            p = f'features.{i + 1}.conv'
            if not b.expand:
                out[p + '.0.weight'] = ((b.cexp, 1, b.k, b.k), 'param')
                bn(p + '.1', b.cexp)
                if b.se:
                    se(p + '.3', b.cexp, b.se)
                out[p + '.4.weight'] = ((b.cout, b.cexp, 1, 1), 'param')
                bn(p + '.5', b.cout)
            else:
                out[p + '.0.weight'] = ((b.cexp, b.cin, 1, 1), 'param')
                bn(p + '.1', b.cexp)
                out[p + '.3.weight'] = ((b.cexp, 1, b.k, b.k), 'param')
                bn(p + '.4', b.cexp)
                if b.se:
                    se(p + '.5', b.cexp, b.se)
                out[p + '.7.weight'] = ((b.cout, b.cexp, 1, 1), 'param')
                bn(p + '.8', b.cout)
        last_name = self.last_name
        out[last_name + '.0.weight'] = ((self.last_c, self.blocks[-1].cout, 1, 1), 'param')
        bn(last_name + '.1', self.last_c)
        if self.classifier:
            out['classifier.0.weight'] = ((self.classifier, self.last_c), 'param')
            out['classifier.0.bias'] = ((self.classifier,), 'param')
            bn('classifier.1', self.classifier)
        for k in range(9):                                   # always 9 heads (model_builder.py:78-81)
            out[f'regressors.{k}.0.weight'] = ((18, self.feat_c), 'param')
            out[f'regressors.{k}.0.bias'] = ((18,), 'param')
        out['cls_fc.1.weight'] = ((num_classes, self.feat_c), 'param')
        out['cls_fc.1.bias'] = ((num_classes,), 'param')
        return out

    def _resnet_shapes(self, num_classes):
        """torchvision's ResNet-50 state-dict keys + the wrapper's heads (model_builder.py:79-85)."""
        out = {}

        def bn(p, c):
            out[p + '.weight'] = ((c,), 'param')
            out[p + '.bias'] = ((c,), 'param')
            out[p + '.running_mean'] = ((c,), 'buffer')
            out[p + '.running_var'] = ((c,), 'buffer')
            out[p + '.num_batches_tracked'] = ((), 'buffer')
        out['conv1.weight'] = ((64, 3, 7, 7), 'param')
        bn('bn1', 64)
        cin = 64
        for li, (w, n, s) in enumerate(self.layers):
            for i in range(n):
                p = f'layer{li + 1}.{i}'
                out[p + '.conv1.weight'] = ((w, cin, 1, 1), 'param')
                bn(p + '.bn1', w)
                out[p + '.conv2.weight'] = ((w, w, 3, 3), 'param')
                bn(p + '.bn2', w)
                out[p + '.conv3.weight'] = ((4 * w, w, 1, 1), 'param')
                bn(p + '.bn3', 4 * w)
                if i == 0:
                    out[p + '.downsample.0.weight'] = ((4 * w, cin, 1, 1), 'param')
                    bn(p + '.downsample.1', 4 * w)
                cin = 4 * w
        for k in range(9):
            out[f'regressors.{k}.0.weight'] = ((18, self.feat_c), 'param')
            out[f'regressors.{k}.0.bias'] = ((18,), 'param')
        out['cls_fc.1.weight'] = ((num_classes, self.feat_c), 'param')
        out['cls_fc.1.bias'] = ((num_classes,), 'param')
        return out

    @property
    def last_name(self):
        return 'conv'   # mobilenetv3.py:188 (`self.conv`); mobilenetv2 follows the same module layout
