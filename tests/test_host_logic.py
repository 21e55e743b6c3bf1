"""CPU tests of the host side of the boundary (no GPU, no compute through the HIP library): state-dict
compatibility with the reference, builders, LossManager / ALWA bookkeeping against the reference's golden trace,
config reader, checkpoint round trip, the host-side 3-D IoU, and the loud failure when the GPU path is asked to
run on the CPU."""
import os

import numpy as np
import pytest
import torch

from torchdet3d.utils import AttrDict


def _cfg(name='mobilenetv3_large', nc=9, **kw):
    d = dict(model=dict(name=name, num_classes=nc, pretrained=False),
             optim=dict(name='adam', lr=1e-3, betas=(0.9, 0.999), wd=1e-4, rho=0.9, alpha=0.99, momentum=0.9,
                        nesterov=True),
             scheduler=dict(name='cosine', exp_gamma=0.9, steps=[2, 4], gamma=0.1), data=dict(max_epochs=5),
             loss=dict(names=['l1', 'add_loss', 'cross_entropy'], coeffs=([1., .1], [.2]), smoothl1_beta=.2, w=5.18,
                       eps=1., alwa=dict(use=False, lam_cls=1., lam_reg=1., C=100, compute_std=True)))
    d.update(kw)
    return AttrDict(d)


@pytest.mark.parametrize('name', ['mobilenetv3_large', 'mobilenetv3_small', 'mobilenetv2'])
def test_state_dict_keys_and_shapes_match_reference(name):
    from oracle.model import state_dict_shapes       # pinned: gen_golden.py asserts these equal the reference's keys
    from torchdet3d.builders import build_model
    m = build_model(_cfg(name))
    sd = m.state_dict()
    ref = state_dict_shapes(name, 9)
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    assert sd['features.0.1.num_batches_tracked'].dtype == torch.int64
    if name == 'mobilenetv3_large':
        assert len(sd) == 335 and sum(v.numel() for k, v in sd.items() if 'running' not in k and 'tracked' not in k) == 4423643


def test_state_dict_round_trip_and_partial_load(tmp_path):
    from oracle.weights import make_state_dict
    from torchdet3d.builders import build_model
    from torchdet3d.utils import load_pretrained_weights, resume_from, save_snap
    ref = make_state_dict('mobilenetv2', 9)
    m = build_model(_cfg('mobilenetv2'))
    m.load_state_dict(ref)
    got = m.state_dict()
    assert all(torch.equal(got[k], ref[k]) for k in ref)
    # the flat parameter aliases the named tensors
    assert m.flat.data_ptr() == m.net.flat.data_ptr() and len(list(m.parameters())) == 1
    # DataParallel-style 'module.' prefix + a mismatching tensor are handled as utils.py:127-183 does
    dp = {'module.' + k: v for k, v in ref.items()}
    dp['module.cls_fc.1.weight'] = torch.zeros(3, 1280)
    m2 = build_model(_cfg('mobilenetv2'))
    load_pretrained_weights(m2, pretrained_dict=dp)
    assert torch.equal(m2.state_dict()['conv.0.weight'], ref['conv.0.weight'])
    with pytest.raises(RuntimeError):
        load_pretrained_weights(m2, pretrained_dict={'nothing': torch.zeros(1)})
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1)
    save_snap(m, opt, sched, 3, str(tmp_path))
    m3 = build_model(_cfg('mobilenetv2'))
    assert resume_from(m3, str(tmp_path / 'snap_3.pth'), torch.optim.AdamW(m3.parameters(), lr=1e-3)) == 4
    assert torch.equal(m3.state_dict()['features.3.conv.3.weight'], ref['features.3.conv.3.weight'])


def test_builders_like_reference_test_builders():
    """reference tests/test_pipeline.py:32-48: every loss builds 1+1 criterions, every optimizer x scheduler builds."""
    from torchdet3d.builders import AVAILABLE_LOSS, build_loss, build_model, build_optimizer, build_scheduler
    for name in AVAILABLE_LOSS:
        if name == 'cross_entropy':
            continue
        cfg = _cfg()
        cfg.loss.names = [name, 'cross_entropy']
        reg, cls = build_loss(cfg)
        assert len(reg) == 1 and len(cls) == 1
    m = build_model(_cfg('mobilenetv2'))
    for o in ['sgd', 'rmsprop', 'adam', 'adadelta']:
        for s in ['cosine', 'exp', 'stepLR', 'multistepLR']:
            cfg = _cfg()
            cfg.optim.name, cfg.scheduler.name = o, s
            opt = build_optimizer(cfg, m)
            assert build_scheduler(cfg, opt) is not None
    with pytest.raises(AssertionError):
        build_model(_cfg('resnet1000'))


def test_alwa_trace_matches_reference_golden(golden_dir):
    from torchdet3d.losses import LossManager
    g = np.load(os.path.join(golden_dir, 'alwa.npz'))
    for ver in (1, 0):
        st = {}
        lm = LossManager(([lambda p, t: st['r']], [lambda p, t: st['c']]), ([1.], [1.]),
                         AttrDict(use=True, lam_cls=1., lam_reg=1., C=50, compute_std=bool(ver)))
        for it in range(250):
            st['r'], st['c'] = torch.tensor(g['seq_reg'][it]), torch.tensor(g['seq_cls'][it])
            tot = lm.parse_losses(None, None, None, None, it).item()
            assert abs(tot - g[f'total:{ver}'][it]) < 1e-6
            assert abs(lm.lam_cls - g[f'lam_cls:{ver}'][it]) < 1e-7


def test_loss_manager_asserts_like_reference():
    from torchdet3d.losses import ADD_loss, CrossEntropyLoss, L1Loss, LossManager
    off = AttrDict(use=False, lam_cls=1., lam_reg=1., C=100, compute_std=True)
    with pytest.raises(AssertionError):
        LossManager(([L1Loss()], []), ([1., 2.], []), off)
    with pytest.raises(AssertionError):
        LossManager(([], [CrossEntropyLoss()]), ([], [1.]), off)
    with pytest.raises(AssertionError):        # ALWA needs a class criterion and unit leading coefficients
        LossManager(([L1Loss(), ADD_loss()], []), ([1., .1], []), AttrDict(use=True, lam_cls=1., lam_reg=1., C=10, compute_std=True))
    lm = LossManager(([L1Loss(), ADD_loss()], [CrossEntropyLoss()]), ([1., .1], [.2]), off)
    c = lm.loss_cfg()
    assert (c.c_l1, round(c.c_add, 6), round(c.c_ce, 6)) == (1.0, 0.1, 0.2)


def test_read_py_config_and_attrdict(tmp_path):
    from torchdet3d.utils import read_py_config
    f = tmp_path / 'cfg.py'
    f.write_text("model = dict(name='mobilenetv2', num_classes=9)\nloss = dict(names=['l1'], coeffs=([1.], []))\n")
    cfg = read_py_config(str(f))
    assert cfg.model.name == 'mobilenetv2' and cfg.loss.coeffs == ([1.], [])
    assert not cfg.model.resume and not cfg.data_parallel.use_parallel      # missing keys are falsy (addict behaviour)
    bad = tmp_path / 'a.b.py'
    bad.write_text('x = 1\n')
    with pytest.raises(ValueError):
        read_py_config(str(bad))


def test_gpu_path_fails_loudly_on_cpu():
    from torchdet3d.builders import build_model
    from torchdet3d.losses import L1Loss
    m = build_model(_cfg('mobilenetv2'))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        L1Loss()(torch.zeros(2, 9, 2), torch.zeros(2, 9, 2))


def test_host_camera_helpers_match_reference_cases(golden_dir):
    """The host-side camera helpers of torchdet3d.utils (geometry.py:16-48) against the reference's golden values: NDC
    camera, NDC keypoints, projection of the golden lift (`lift_2d` itself is the device kernel:
    tests/test_gpu_geometry.py::test_public_lift_2d_is_the_device_lift)."""
    from torchdet3d.utils import convert_2d_to_ndc, convert_camera_matrix_2_ndc, get_default_camera_matrix, project_3d_points
    g = np.load(os.path.join(golden_dir, 'geometry.npz'))
    ndc = convert_camera_matrix_2_ndc(get_default_camera_matrix())
    np.testing.assert_allclose(ndc, g['ndc_cam'])
    np.testing.assert_allclose(convert_2d_to_ndc(g['test_kps'], portrait=True), g['kps_ndc'])
    np.testing.assert_allclose(project_3d_points(g['lift_portrait'], ndc), g['reproj'], atol=1e-9)
    land = convert_2d_to_ndc(g['test_kps'], portrait=False)
    np.testing.assert_allclose(land, np.stack([g['test_kps'][:, 0] * 2 - 1, 1 - g['test_kps'][:, 1] * 2], 1))
    wide = convert_camera_matrix_2_ndc(np.array([[800., 0, 320], [0, 820, 240], [0, 0, 1]]), img_shape=(640, 480))
    np.testing.assert_allclose(wide, [[2.5, 0, 0.], [0, 820 / 240, 0.], [0, 0, 1]], atol=1e-12)
