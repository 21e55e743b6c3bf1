"""Architecture tables for the oracle (test infrastructure, see oracle/__init__.py).

mobilenetv3_{large,small}: rows follow torchdet3d/models/mobilenetv3.py:20-52
(k, t, c, SE, HS, s); channel rounding follows `_make_divisible`
(mobilenetv3.py:54-71) and the constructor loop (mobilenetv3.py:174-188).
mobilenetv2: standard MobileNetV2-1.0 (t,c,n,s) table -- no reference source
(SURVEY.md Appendix B), parity unpinned.
"""

MNV3_ROWS = {
    'mobilenetv3_large': [
        (3, 1, 16, 0, 0, 1), (3, 4, 24, 0, 0, 2), (3, 3, 24, 0, 0, 1),
        (5, 3, 40, 1, 0, 2), (5, 3, 40, 1, 0, 1), (5, 3, 40, 1, 0, 1),
        (3, 6, 80, 0, 1, 2), (3, 2.5, 80, 0, 1, 1), (3, 2.3, 80, 0, 1, 1),
        (3, 2.3, 80, 0, 1, 1), (3, 6, 112, 1, 1, 1), (3, 6, 112, 1, 1, 1),
        (5, 6, 160, 1, 1, 2), (5, 6, 160, 1, 1, 1), (5, 6, 160, 1, 1, 1)],
    'mobilenetv3_small': [
        (3, 1, 16, 1, 0, 2), (3, 4.5, 24, 0, 0, 2), (3, 3.67, 24, 0, 0, 1),
        (5, 4, 40, 1, 1, 2), (5, 6, 40, 1, 1, 1), (5, 6, 40, 1, 1, 1),
        (5, 3, 48, 1, 1, 1), (5, 3, 48, 1, 1, 1), (5, 6, 96, 1, 1, 2),
        (5, 6, 96, 1, 1, 1), (5, 6, 96, 1, 1, 1)],
}

MNV2_TCNS = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
             (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
# the reference's MobileNetV3 class fed MobileNetV2's table as rows (k=3, no SE, ReLU): a model the REAL reference can
# build (mobilenetv3.py:169-197 takes any row table) whose layers from 112x112x96 on have the headline model's shapes
MNV3_ROWS['mobilenetv3_mnv2rows'] = [(3, t, c, 0, 0, s if i == 0 else 1) for t, c, n, s in MNV2_TCNS for i in range(n)]


def make_divisible(v, divisor=8, min_value=None):
    # mobilenetv3.py:54-71
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def arch(name):
    """Returns a dict describing the network:
    stem_c, stem_act, blocks=[dict(cin, cexp, cout, k, s, se (hidden or 0), act, res)],
    last_c, last_act, classifier (out features or 0), feat_c (head input width)."""
    if name in MNV3_ROWS:
        cin = make_divisible(16)
        blocks = []
        for k, t, c, se, hs, s in MNV3_ROWS[name]:
            cout = make_divisible(c)
            cexp = make_divisible(cin * t)
            blocks.append(dict(cin=cin, cexp=cexp, cout=cout, k=k, s=s,
                               se=make_divisible(cexp // 4) if se else 0,
                               act='hswish' if hs else 'relu',
                               res=(s == 1 and cin == cout)))
            cin = cout
        feat = 1024 if name.endswith('small') else 1280
        return dict(name=name, stem_c=make_divisible(16), stem_act='hswish', blocks=blocks,
                    last_c=cexp, last_act='hswish', classifier=feat, feat_c=feat)
    if name == 'mobilenetv2':
        cin = 32
        blocks = []
        for t, c, n, s in MNV2_TCNS:
            for i in range(n):
                st = s if i == 0 else 1
                blocks.append(dict(cin=cin, cexp=cin * t, cout=c, k=3, s=st, se=0,
                                   act='relu6', res=(st == 1 and cin == c)))
                cin = c
        return dict(name=name, stem_c=32, stem_act='relu6', blocks=blocks,
                    last_c=1280, last_act='relu6', classifier=0, feat_c=1280)
    raise AssertionError(f'unknown model {name}')
