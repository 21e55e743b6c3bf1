"""Golden-vector generator: runs the REAL reference (imported from /root/reference
with its absent third-party imports stubbed) on deterministic inputs/weights and
writes small fixtures to tests/golden/.  Runs only in the build container (the
reference never travels to the GPU box); the fixtures + this script are
committed.  Usage:  python -m oracle.gen_golden
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _AttrDict(dict):
    """Tiny stand-in for addict.Dict (missing keys -> empty dict, attribute access)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = _AttrDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        if k not in self:
            return _AttrDict()
        return self[k]

    def __setattr__(self, k, v):
        self[k] = v


def import_reference():
    class _Any:
        def __init__(self, *a, **k):
            pass

    _stub('cv2', INTER_LINEAR=1)
    _stub('glog')
    _stub('gdown')
    _stub('icecream', ic=print)
    _stub('prettytable', PrettyTable=_Any)
    alb = _stub('albumentations', **{n: _Any for n in (
        'Compose', 'KeypointParams', 'Crop', 'Resize', 'HorizontalFlip', 'HueSaturationValue',
        'RGBShift', 'RandomBrightnessContrast', 'ColorJitter', 'Blur', 'OneOf')})
    aug = _stub('albumentations.augmentations')
    tr = _stub('albumentations.augmentations.transforms', Normalize=_Any)
    core = _stub('albumentations.core')
    ti = _stub('albumentations.core.transforms_interface', BasicTransform=_Any, ImageOnlyTransform=_Any,
               DualTransform=_Any, to_tuple=lambda *a, **k: a)
    alb.augmentations, aug.transforms, alb.core, core.transforms_interface = aug, tr, core, ti
    _stub('timm')
    _stub('timm.models')
    _stub('timm.models.mobilenetv3', mobilenetv3_large_100=_Any)
    _stub('efficientnet_lite_pytorch', EfficientNet=_Any)
    _stub('efficientnet_lite_pytorch.utils', get_model_params=lambda *a, **k: (None, None))
    for i in range(3):
        cls = type(f'EfficientnetLite{i}ModelFile', (), {'get_model_file_path': staticmethod(lambda: '')})
        _stub(f'efficientnet_lite{i}_pytorch_model', **{f'EfficientnetLite{i}ModelFile': cls})
    _stub('objectron')
    _stub('objectron.dataset', iou=types.ModuleType('iou'), box=types.ModuleType('box'),
          graphics=types.ModuleType('graphics'))
    _stub('addict', Dict=_AttrDict)
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        _stub('torch.utils.tensorboard', SummaryWriter=_Any)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import torchdet3d  # noqa: F401
    finally:
        os.chdir(cwd)
    return sys.modules['torchdet3d']


def cfg_for(name, num_classes, loss_names, coeffs):
    return _AttrDict(dict(
        model=dict(name=name, pretrained=False, num_classes=num_classes),
        loss=dict(names=loss_names, coeffs=coeffs, smoothl1_beta=0.2, w=5.18, eps=1.,
                  alwa=dict(use=False, lam_cls=1., lam_reg=1., C=100, compute_std=True))))


def checks(t):
    t = t.detach().double().flatten()
    idx = np.linspace(0, t.numel() - 1, 8).astype(np.int64)
    return np.array([t.mean().item(), t.abs().mean().item()] + [t[i].item() for i in idx])


def model_golden(td3, name, B, HW, num_classes, loss_names, coeffs, tag, big=False):
    from oracle.weights import make_state_dict, make_inputs
    from torchdet3d.builders import build_model, build_loss
    from torchdet3d.losses import LossManager
    cfg = cfg_for(name, num_classes, loss_names, coeffs)
    if name == 'mobilenetv3_mnv2rows':
        # not a name of the reference's build_model: its MobileNetV3 class (models/mobilenetv3.py:169-197) takes any row
        # table, so it is instantiated -- through the reference's own model_wrapper (builders/model_builder.py:73-152),
        # exactly as build_model does for 'mobilenetv3_large' (:43-46) -- with MobileNetV2's (t, c, n, s) table as rows
        # (k=3, SE=0, HS=0): from 112x112x96 on every depthwise / pointwise layer has the headline model's shape
        from torchdet3d.builders.model_builder import model_wrapper
        from torchdet3d.models.mobilenetv3 import MobileNetV3
        from oracle.specs import MNV3_ROWS
        net = model_wrapper(model_class=MobileNetV3, output_channels=1280, num_classes=num_classes,
                            cfgs=[list(r) for r in MNV3_ROWS[name]], mode='large')
    else:
        net = build_model(cfg)
    sd = make_state_dict(name, num_classes)
    assert list(net.state_dict().keys()) == list(sd.keys()), 'state-dict key mismatch'
    net.load_state_dict(sd)
    imgs, gt_kp, cats = make_inputs(B, HW, HW, num_classes)
    out = {}
    # ---- eval forward
    net.eval()
    with torch.no_grad():
        kp, tg = net(imgs, cats)
    out['eval_kp'] = kp.numpy()
    out['eval_targets'] = tg.numpy()
    if num_classes > 1:
        out['eval_argmax'] = tg.argmax(1).numpy()
    # ---- export-mode forward (all 9 heads)
    with torch.no_grad():
        okp, otg = net.forward_to_onnx(imgs)
    out['onnx_kp'] = okp.numpy()
    out['onnx_targets'] = otg.numpy()
    # ---- train forward + backward (dropout mask captured by hook)
    net.train()
    grab = {}
    h = net.cls_fc[0].register_forward_hook(
        lambda m, i, o: grab.__setitem__('mask', (o != 0).float() * 2 * (i[0] != 0).float() + (i[0] == 0).float() * 2))
    taps = {}
    hooks = [h]
    for i, m in enumerate(net.features):
        hooks.append(m.register_forward_hook(lambda m_, i_, o, k=f'features.{i}': taps.__setitem__(k, o)))
    hooks.append(net.conv.register_forward_hook(lambda m_, i_, o: taps.__setitem__('conv', o)))
    torch.manual_seed(1234)
    kp, tg = net(imgs, cats)
    crit = build_loss(cfg)
    lm = LossManager(crit, cfg.loss.coeffs, cfg.loss.alwa)
    kp.retain_grad()
    if num_classes > 1:
        tg.retain_grad()
    loss = lm.parse_losses(kp, gt_kp, tg, cats, 0)
    loss.backward()
    for hh in hooks:
        hh.remove()
    out['train_kp'] = kp.detach().numpy()
    out['train_targets'] = tg.detach().numpy()
    if 'mask' in grab:
        # an input element that is exactly 0 hides the mask bit; such elements are set to "kept" (2)
        out['dropout_mask'] = grab['mask'].numpy().astype(np.uint8)
    out['loss'] = loss.detach().numpy().reshape(-1)
    out['dkp'] = kp.grad.numpy()
    if num_classes > 1:
        out['dtargets'] = tg.grad.numpy()
    for k, v in taps.items():
        out['tap:' + k] = checks(v)
    if big:
        # production-resolution fixture (B >= 32 @ 224^2): also the reference's own metrics of its eval outputs, so the
        # bf16 throughput mode can be gated on ADD / SADD / accuracy (evaluation/metrics.py:10-37) against the reference
        from torchdet3d.evaluation.metrics import compute_average_distance, compute_accuracy
        ekp, etg = torch.from_numpy(out['eval_kp']), torch.from_numpy(out['eval_targets'])
        out['eval_add_sadd'] = np.array(compute_average_distance(ekp, gt_kp))
        out['eval_acc'] = np.array(compute_accuracy(etg, cats))
    full = {'features.0.0.weight', 'conv.0.weight', 'cls_fc.1.weight', 'cls_fc.1.bias',
            'regressors.0.0.weight', 'regressors.4.0.weight', 'regressors.4.0.bias'}
    for k, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out['gsum:' + k] = np.array([g.double().sum().item(), g.double().abs().sum().item()])
        if k in full or k.startswith('features.1.') or k.startswith('features.5.') or k.startswith('classifier') \
                or (name == 'mobilenetv3_mnv2rows' and k.startswith(('features.2.', 'features.3.', 'features.14.'))):
            if g.numel() <= 40000:
                out['grad:' + k] = g.numpy()
            else:                       # keep fixtures small: leading rows only
                out['gradrows:' + k] = g[:6].numpy()
    if big:
        # conditioning of the gradient comparison: the SAME reference model in fp64 (`net.double()`, the captured dropout mask
        # imposed by a fixed-mask module in place of nn.Dropout) -- its gradients are the exact ones to ~1e-12, so
        # `g64l2:<key>` = relative L2 distance of the reference's own fp32 gradient from the truth.  For the MobileNetV2-shaped
        # ReLU network that is 0.4 .. 0.8 % (two fp32 implementations cannot agree better than that); for mobilenetv3_large
        # 0.01 .. 0.05 %.  The HIP path is held to the fp64 gradient with a bound expressed in these numbers (test_gpu_golden.py)
        class _FixedMask(torch.nn.Module):
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, x):
                return x * self.m
        import copy
        net64 = copy.deepcopy(net)
        net64.load_state_dict(sd)                  # (the fp32 run moved the BatchNorm running statistics)
        net64.double()
        net64.train()
        if 'mask' in grab:
            net64.cls_fc[0] = _FixedMask(grab['mask'].double())
        for p in net64.parameters():
            p.grad = None
        kp64, tg64 = net64(imgs.double(), cats)
        lm64 = LossManager(build_loss(cfg), cfg.loss.coeffs, cfg.loss.alwa)
        lm64.parse_losses(kp64, gt_kp.double(), tg64, cats, 0).backward()
        g32 = dict(net.named_parameters())
        gmax64 = max(p.grad.abs().max().item() for p in net64.parameters() if p.grad is not None)
        out['g64max'] = np.array(gmax64)
        for k, p in net64.named_parameters():
            g64 = p.grad if p.grad is not None else torch.zeros_like(p)
            g_ = g32[k].grad if g32[k].grad is not None else torch.zeros_like(g32[k])
            nrm = max(g64.norm().item(), 1e-3 * gmax64 * g64.numel() ** 0.5)
            out['g64l2:' + k] = np.array((g_.double() - g64).norm().item() / nrm)
            if ('grad:' + k) in out:
                out['grad64:' + k] = g64.numpy()
            elif ('gradrows:' + k) in out:
                out['gradrows64:' + k] = g64[:6].numpy()
    new_sd = net.state_dict()
    for k in ('features.0.1', 'features.5.conv.4', 'conv.1', 'classifier.1'):
        if k + '.running_mean' in new_sd:
            out['rm:' + k] = new_sd[k + '.running_mean'].numpy()
            out['rv:' + k] = new_sd[k + '.running_var'].numpy()
            out['nbt:' + k] = new_sd[k + '.num_batches_tracked'].numpy()
    np.savez_compressed(os.path.join(OUT, tag + '.npz'), **out)
    print(tag, 'loss', out['loss'], 'keys', len(out))


def losses_golden(td3):
    from torchdet3d.losses import DiagLoss, ADD_loss, WingLoss
    g = torch.Generator().manual_seed(1234)
    out = {}
    for B in (256, 7):
        p = torch.sigmoid(torch.randn(B, 9, 2, generator=g))
        t = torch.sigmoid(torch.randn(B, 9, 2, generator=g))
        out[f'p{B}'], out[f't{B}'] = p.numpy(), t.numpy()
        crits = dict(l1=torch.nn.L1Loss(), mse=torch.nn.MSELoss(),
                     smoothl1=torch.nn.SmoothL1Loss(beta=0.2), add_loss=ADD_loss(), diag_loss=DiagLoss(),
                     wing_default=WingLoss(), wing_cfg=WingLoss(w=5.18, eps=1.),
                     wing_quirk=WingLoss(w=0.3, eps=0.05))
        for n, c in crits.items():
            pp = p.clone().requires_grad_(True)
            v = c(pp, t)
            v.backward()
            out[f'{n}:{B}:val'] = v.detach().numpy()
            out[f'{n}:{B}:grad'] = pp.grad.numpy()
        logits = torch.randn(B, 9, generator=g)
        cats = torch.randint(0, 9, (B,), generator=g)
        ll = logits.clone().requires_grad_(True)
        v = torch.nn.CrossEntropyLoss()(ll, cats)
        v.backward()
        out[f'logits{B}'], out[f'cats{B}'] = logits.numpy(), cats.numpy()
        out[f'ce:{B}:val'], out[f'ce:{B}:grad'] = v.detach().numpy(), ll.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'losses.npz'), **out)
    print('losses', {k: float(v) for k, v in out.items() if k.endswith('256:val')})


def metrics_golden(td3):
    from torchdet3d.evaluation.metrics import compute_average_distance, compute_accuracy
    g = torch.Generator().manual_seed(77)
    p, t = torch.rand(64, 9, 2, generator=g), torch.rand(64, 9, 2, generator=g)
    # permute a few gt rows so the symmetric distance differs from the plain one
    t[::3] = p[::3][:, torch.tensor([1, 0, 3, 2, 5, 4, 7, 6, 8])] + 0.01
    logits = torch.randn(64, 9, generator=g)
    cats = torch.randint(0, 9, (64,), generator=g)
    out = dict(p=p.numpy(), t=t.numpy(), logits=logits.numpy(), cats=cats.numpy())
    out['add_mean'] = np.array(compute_average_distance(p, t))
    out['add_sum'] = np.array(compute_average_distance(p, t, reduce_mean=False))
    out['acc_mean'] = np.array(compute_accuracy(logits, cats))
    out['acc_sum'] = np.array(compute_accuracy(logits, cats, reduce_mean=False))
    np.savez_compressed(os.path.join(OUT, 'metrics.npz'), **out)
    print('metrics', out['add_mean'], out['acc_mean'])


def alwa_golden(td3):
    from torchdet3d.losses import LossManager
    out = {}
    rng = np.random.default_rng(5)
    seq_reg = rng.uniform(0.2, 0.6, 250).astype(np.float32)
    seq_cls = rng.uniform(0.5, 2.5, 250).astype(np.float32)
    out['seq_reg'], out['seq_cls'] = seq_reg, seq_cls
    for ver in (True, False):
        alwa = _AttrDict(dict(use=True, lam_cls=1., lam_reg=1., C=50, compute_std=ver))
        state = {}
        reg_cr = [lambda p, t: state['reg']]
        cls_cr = [lambda p, t: state['cls']]
        lm = LossManager((reg_cr, cls_cr), ([1.], [1.]), alwa)
        lams, tot = [], []
        for it in range(250):
            state['reg'] = torch.tensor(seq_reg[it])
            state['cls'] = torch.tensor(seq_cls[it])
            tot.append(lm.parse_losses(None, None, None, None, it).item())
            lams.append(lm.lam_cls)
        out[f'lam_cls:{int(ver)}'] = np.array(lams)
        out[f'total:{int(ver)}'] = np.array(tot)
    np.savez_compressed(os.path.join(OUT, 'alwa.npz'), **out)
    print('alwa final lam', out['lam_cls:1'][-1], out['lam_cls:0'][-1])


def geometry_golden():
    spec = importlib.util.spec_from_file_location('ref_geometry', REF + '/torchdet3d/utils/geometry.py')
    geo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(geo)
    kps = np.array([[0.47714591, 0.47491544], [0.73884577, 0.39749265], [0.18508956, 0.40002537],
                    [0.74114597, 0.48664019], [0.18273196, 0.48833901], [0.64639187, 0.46719882],
                    [0.32766378, 0.46827659], [0.64726073, 0.51853681], [0.32699507, 0.51933688]])
    out = dict(test_kps=kps)            # reference tests/test_geometry.py:13-21
    out['lift_portrait'] = geo.lift_2d([kps], portrait=True)[0]
    out['lift_landscape'] = geo.lift_2d([kps], portrait=False)[0]
    np.random.seed(10)                  # tests/test_geometry.py:32-33
    noisy = np.clip(kps + 0.01 * np.random.rand(*kps.shape), 0, 1)
    out['noisy_kps'] = noisy
    out['lift_noisy'] = geo.lift_2d([noisy], portrait=True)[0]
    rng = np.random.default_rng(3)
    rnd = rng.uniform(0.1, 0.9, (16, 9, 2))
    out['rand_kps'] = rnd
    out['lift_rand'] = np.stack(geo.lift_2d(list(rnd), portrait=True))
    ndc = geo.convert_camera_matrix_2_ndc(geo.get_default_camera_matrix())
    out['ndc_cam'] = ndc
    out['reproj'] = geo.project_3d_points(out['lift_portrait'], ndc)
    out['kps_ndc'] = geo.convert_2d_to_ndc(kps, portrait=True)
    np.savez_compressed(os.path.join(OUT, 'geometry.npz'), **out)
    print('geometry first row', out['lift_portrait'][0])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    td3 = import_reference()
    default = (['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2]))
    model_golden(td3, 'mobilenetv3_large', 4, 96, 9, *default, tag='mnv3_large_b4_96')
    model_golden(td3, 'mobilenetv3_small', 4, 96, 9, *default, tag='mnv3_small_b4_96')
    model_golden(td3, 'mobilenetv3_large', 8, 96, 1, ['mse', 'diag_loss', 'add_loss'],
                 ([1., .5, .1], []), tag='mnv3_large_c1_b8_96')
    model_golden(td3, 'mobilenetv3_large', 2, 224, 9, ['smoothl1', 'wing', 'cross_entropy'],
                 ([1., .3], [.5]), tag='mnv3_large_b2_224')
    model_golden(td3, 'mobilenetv3_large', 32, 224, 9, *default, tag='mnv3_large_b32_224', big=True)
    model_golden(td3, 'mobilenetv3_mnv2rows', 32, 224, 9, *default, tag='mnv2rows_b32_224', big=True)
    losses_golden(td3)
    metrics_golden(td3)
    alwa_golden(td3)
    geometry_golden()


if __name__ == '__main__':
    main()
