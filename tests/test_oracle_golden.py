"""Pins the CPU oracle against golden vectors produced by the REAL reference
(oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import losses as OL
from oracle import metrics as OM
from oracle import model as OMod
from oracle.weights import make_inputs, make_state_dict

CASES = [('mnv3_large_b4_96', 'mobilenetv3_large', 4, 96, 9, ['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2])),
         ('mnv3_small_b4_96', 'mobilenetv3_small', 4, 96, 9, ['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2])),
         ('mnv3_large_c1_b8_96', 'mobilenetv3_large', 8, 96, 1, ['mse', 'diag_loss', 'add_loss'], ([1., .5, .1], [])),
         ('mnv3_large_b2_224', 'mobilenetv3_large', 2, 224, 9, ['smoothl1', 'wing', 'cross_entropy'], ([1., .3], [.5])),
         # the reference's MobileNetV3 class over MobileNetV2's row table (the headline model's layer shapes), 32 crops @224
         ('mnv2rows_b32_224', 'mobilenetv3_mnv2rows', 32, 224, 9, ['l1', 'add_loss', 'cross_entropy'], ([1., .1], [.2]))]


@pytest.mark.parametrize('tag,name,B,HW,nc,lnames,coeffs', CASES)
def test_model_matches_reference(golden_dir, tag, name, B, HW, nc, lnames, coeffs):
    g = np.load(os.path.join(golden_dir, tag + '.npz'))
    sd = make_state_dict(name, nc)
    imgs, gt_kp, cats = make_inputs(B, HW, HW, nc)
    with torch.no_grad():
        kp, tg = OMod.forward(sd, name, imgs, cats, train=False, num_classes=nc)
    np.testing.assert_allclose(kp.numpy(), g['eval_kp'], atol=2e-6)
    np.testing.assert_allclose(tg.numpy(), g['eval_targets'], atol=2e-5)
    if nc > 1:
        assert (tg.argmax(1).numpy() == g['eval_argmax']).all()
    okp, otg = OMod.forward_to_onnx(sd, name, imgs, nc)
    np.testing.assert_allclose(okp.numpy(), g['onnx_kp'], atol=2e-6)
    # train step
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k)
              for k, v in sd.items()}
    mask = torch.from_numpy(g['dropout_mask'].astype(np.float32)) if 'dropout_mask' in g.files else None
    taps = {}
    kp, tg = OMod.forward(params, name, imgs, cats, train=True, num_classes=nc, dropout_mask=mask, taps=taps)
    lm = OL.LossManager(OL.build(lnames), coeffs)
    kp.retain_grad()
    loss = lm.parse_losses(kp, gt_kp, tg, cats, 0)
    loss.backward()
    np.testing.assert_allclose(kp.detach().numpy(), g['train_kp'], atol=2e-6)
    np.testing.assert_allclose(loss.detach().numpy().reshape(-1), g['loss'], rtol=2e-6)
    np.testing.assert_allclose(kp.grad.numpy(), g['dkp'], atol=1e-7)
    for k in [f for f in g.files if f.startswith('tap:')]:
        t = taps[k[4:]].detach().double().flatten()
        assert abs(t.mean().item() - g[k][0]) < 1e-5 and abs(t.abs().mean().item() - g[k][1]) < 1e-5
    for k in [f for f in g.files if f.startswith('gsum:')]:
        gr = params[k[5:]].grad
        gr = torch.zeros(1) if gr is None else gr
        ref = g[k]
        assert abs(gr.double().abs().sum().item() - ref[1]) <= 2e-4 * max(1.0, ref[1]), k
    for k in [f for f in g.files if f.startswith('grad:')]:
        gr = params[k[5:]].grad
        gr = torch.zeros_like(params[k[5:]]) if gr is None else gr
        np.testing.assert_allclose(gr.numpy(), g[k], atol=3e-5, rtol=1e-3, err_msg=k)
    for k in [f for f in g.files if f.startswith('rm:')]:
        np.testing.assert_allclose(params[k[3:] + '.running_mean'].detach().numpy(), g[k], atol=1e-6)
        np.testing.assert_allclose(params[k[3:] + '.running_var'].detach().numpy(), g['rv:' + k[3:]], atol=1e-6)
        assert int(params[k[3:] + '.num_batches_tracked']) == int(g['nbt:' + k[3:]])


def test_losses_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'losses.npz'))
    fns = dict(l1=OL.l1, mse=OL.mse, smoothl1=lambda p, t: OL.smoothl1(p, t, 0.2), add_loss=OL.add_loss,
               diag_loss=OL.diag_loss, wing_default=OL.wing, wing_cfg=lambda p, t: OL.wing(p, t, 5.18, 1.),
               wing_quirk=lambda p, t: OL.wing(p, t, 0.3, 0.05))
    for B in (256, 7):
        p0, t = torch.from_numpy(g[f'p{B}']), torch.from_numpy(g[f't{B}'])
        for n, fn in fns.items():
            p = p0.clone().requires_grad_(True)
            v = fn(p, t)
            v.backward()
            np.testing.assert_allclose(v.item(), g[f'{n}:{B}:val'], rtol=1e-6, err_msg=n)
            np.testing.assert_allclose(p.grad.numpy(), g[f'{n}:{B}:grad'], atol=1e-8, rtol=1e-5, err_msg=n)
        lg = torch.from_numpy(g[f'logits{B}']).requires_grad_(True)
        v = OL.cross_entropy(lg, torch.from_numpy(g[f'cats{B}']))
        v.backward()
        np.testing.assert_allclose(v.item(), g[f'ce:{B}:val'], rtol=1e-6)
        np.testing.assert_allclose(lg.grad.numpy(), g[f'ce:{B}:grad'], atol=1e-8)


def test_metrics_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'metrics.npz'))
    p, t = torch.from_numpy(g['p']), torch.from_numpy(g['t'])
    np.testing.assert_allclose(OM.average_distance(p, t), g['add_mean'], rtol=1e-6)
    np.testing.assert_allclose(OM.average_distance(p, t, reduce_mean=False), g['add_sum'], rtol=1e-6)
    lg, c = torch.from_numpy(g['logits']), torch.from_numpy(g['cats'])
    assert OM.accuracy(lg, c) == g['acc_mean'] and OM.accuracy(lg, c, False) == g['acc_sum']


def test_alwa_trace_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'alwa.npz'))
    for ver in (1, 0):
        st = {}
        lm = OL.LossManager(([lambda p, t: st['r']], [lambda p, t: st['c']]), ([1.], [1.]),
                            use_alwa=True, C=50, compute_std=bool(ver))
        for it in range(250):
            st['r'], st['c'] = torch.tensor(g['seq_reg'][it]), torch.tensor(g['seq_cls'][it])
            tot = lm.parse_losses(None, None, None, None, it).item()
            assert abs(tot - g[f'total:{ver}'][it]) < 1e-6
            assert abs(lm.lam_cls - g[f'lam_cls:{ver}'][it]) < 1e-7
