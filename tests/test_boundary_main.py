"""The drop-in boundary as `scripts/main.py` of the reference uses it.

CPU part: exactly main.py's import lines (:10-15) and test_pipeline.py's (:3-8) resolve against this package, the
`Logger` tee behaves like utils.py:289-333, the model exposes the reference's attributes.
GPU part (`-m gpu`): a replay of main.py:36-106 -- config file -> Logger -> seeds -> build_model / optimizer /
scheduler -> LossManager -> loaders -> Trainer / Evaluator kwargs exactly as written there -> epochs -> visual_test --
on the synthetic crop source, plus `cfg.regime.type == 'evaluation'`."""
import os
import sys
import types

import pytest
import torch

CONFIG = '''
data = dict(root="synthetic", resize=(96, 96), train_batch_size=8, val_batch_size=8, max_epochs=2, num_workers=0,
            synthetic_len=32, category_list='all',
            normalization=dict(mean=[0.5931, 0.4690, 0.4229], std=[0.2471, 0.2214, 0.2157]))
model = dict(name='mobilenetv3_large', pretrained=False, num_classes=9, resume='', load_weights='')
data_parallel = dict(use_parallel=False, parallel_params=dict(device_ids=[0], output_device=0))
optim = dict(name='adam', lr=0.001, momentum=0.9, wd=1e-4, betas=(0.9, 0.999), rho=0.9, alpha=0.99, nesterov=True)
scheduler = dict(name='multistepLR', gamma=0.6, exp_gamma=0.975, steps=[1])
loss = dict(names=['l1', 'add_loss', 'cross_entropy'], coeffs=([1., .1], [.2]), smoothl1_beta=0.2,
            alwa=dict(use=False, lam_cls=1., lam_reg=1., C=100, compute_std=True), w=5.18, eps=1.)
output_dir = '%s'
utils = dict(debug_mode=False, random_seeds=5, save_freq=10, print_freq=20, debug_steps=100, eval_freq=1)
regime = dict(type='%s', vis_only=False)
'''


def test_main_py_and_test_pipeline_import_lines():
    # scripts/main.py:10-15, verbatim
    from torchdet3d.builders import (build_loader, build_model, build_loss,
                                        build_optimizer, build_scheduler)
    from torchdet3d.evaluation import Evaluator
    from torchdet3d.losses import LossManager
    from torchdet3d.trainer import Trainer
    from torchdet3d.utils import read_py_config, Logger, set_random_seed, check_isfile, resume_from
    # tests/test_pipeline.py:3-8, verbatim
    from torchdet3d.evaluation import compute_metrics_per_cls
    from torchdet3d.losses import WingLoss, ADD_loss, DiagLoss
    from torchdet3d.builders import (build_loss, build_optimizer, build_scheduler,              # noqa: F811
                                        build_model, AVAILABLE_LOSS, AVAILABLE_OPTIMS, AVAILABLE_SCHEDS)
    from torchdet3d.utils import read_py_config                                               # noqa: F811
    assert AVAILABLE_OPTIMS == ['sgd', 'rmsprop', 'adam', 'adadelta']
    assert AVAILABLE_SCHEDS == ['cosine', 'exp', 'stepLR', 'multistepLR']
    assert all(callable(f) for f in (build_loader, build_model, build_loss, build_optimizer, build_scheduler,
                                     read_py_config, set_random_seed, check_isfile, resume_from, compute_metrics_per_cls))
    assert all(isinstance(c, type) for c in (Evaluator, LossManager, Trainer, Logger, WingLoss, ADD_loss, DiagLoss))
    # methods main.py calls
    for m in ('val', 'visual_test', 'run_eval_pipe'):
        assert callable(getattr(Evaluator, m))
    assert callable(Trainer.train)


def test_logger_tees_console_into_file_and_check_isfile_warns(tmp_path, capsys):
    from torchdet3d.utils import Logger, check_isfile, mkdir_if_missing
    path = tmp_path / 'deep' / 'er' / 'train.log'
    old = sys.stdout
    try:
        sys.stdout = Logger(str(path))                  # main.py:39; creates the directory (utils.py:308)
        print('hello boundary')
        sys.stdout.flush()
        sys.stdout.close()
    finally:
        sys.stdout = old
    assert path.read_text() == 'hello boundary\n'
    assert 'hello boundary' in capsys.readouterr().out
    with pytest.warns(UserWarning):
        assert check_isfile(str(tmp_path / 'nope')) is False        # warns, does not raise (utils.py:33-45)
    mkdir_if_missing(str(tmp_path / 'deep'))                          # existing directory: no error


def test_model_exposes_reference_attributes_and_pooling_modes():
    from test_host_logic import _cfg
    from torchdet3d.builders import build_model
    from torchdet3d.builders.model_builder import ModelWrapper
    m = build_model(_cfg('mobilenetv3_large'))
    assert len(m.regressors) == 9 and m.regressors[3][0].weight.shape == (18, 1280)       # model_builder.py:78-81
    assert m.cls_fc[1].weight.shape == (9, 1280) and isinstance(m.cls_fc[0], torch.nn.Dropout)
    assert callable(m.sigmoid) and callable(m.extract_features) and callable(m._glob_feature_vector)
    # the head views alias the flat master buffer the optimizer updates
    assert m.regressors[3][0].weight.data_ptr() == m.net.p['regressors.3.0.weight'].data_ptr()
    assert len(list(m.parameters())) == 1
    for mode in ('avg', 'max', 'avg+max'):
        assert ModelWrapper('mobilenetv2', 9, pooling_mode=mode).pooling_mode == mode
    with pytest.raises(ValueError):                                                       # model_builder.py:105-106
        ModelWrapper('mobilenetv2', 9, pooling_mode='median')
    with pytest.raises(ValueError):
        ModelWrapper._glob_feature_vector(torch.zeros(1, 8, 2, 2), 'median')


def test_initialisation_follows_reference_distributions():
    """mobilenetv3.py:205-218: conv N(0, sqrt(2/(k*k*Cout))), BatchNorm2d 1/0, Linear N(0, .01) / 0; the heads are added
    AFTER `_initialize_weights` ran, so they keep PyTorch's default Linear init U(+-1/sqrt(fan_in))
    (model_builder.py:79-85)."""
    from torchdet3d.models.engine import Net
    net = Net('mobilenetv3_large', 9, 'cpu')
    net.reset_parameters(seed=11)
    p = net.p
    for k, std in (('features.0.0.weight', (2 / (9 * 16)) ** .5), ('conv.0.weight', (2 / 960) ** .5),
                   ('features.4.conv.3.weight', (2 / (25 * 72)) ** .5), ('features.14.conv.0.weight', (2 / 960) ** .5)):
        w = p[k]
        assert abs(w.std().item() / std - 1) < 0.12 and abs(w.mean().item()) < 4 * std / w.numel() ** .5, k
    for k, v in p.items():
        if (k.endswith('.1.weight') or k.endswith('.4.weight') or k.endswith('.8.weight') or k.endswith('.5.weight')) and v.dim() == 1:
            assert (v == 1).all(), k                                 # BatchNorm gamma
        if v.dim() == 1 and k.endswith('.bias') and not k.startswith(('regressors', 'cls_fc')):
            assert (v == 0).all(), k                                 # BatchNorm beta, Linear / SE biases
    lin = p['classifier.0.weight']
    assert abs(lin.std().item() / 0.01 - 1) < 0.02
    se = p['features.4.conv.5.fc.0.weight']
    assert abs(se.std().item() / 0.01 - 1) < 0.1
    b = 1 / 1280 ** .5
    for k in ('regressors.0.0.weight', 'regressors.8.0.bias', 'cls_fc.1.weight', 'cls_fc.1.bias'):
        v = p[k]
        assert v.abs().max().item() <= b and (v.numel() < 100 or abs(v.std().item() / (b / 3 ** .5) - 1) < 0.05), k
    assert all((net.buffers[k] == (1 if k.endswith('var') else 0)).all() for k in net.buffers)


class _Writer:
    """SummaryWriter stand-in (torch.utils.tensorboard needs the tensorboard package, absent in this image)."""

    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, tag, value, global_step=None):
        self.scalars.append((tag, float(value), global_step))


def _replay_main(config_path, device='cuda', wo_saving_checkpoint=True):
    """scripts/main.py:36-106 with `args` replaced by the function arguments; every call is the reference's."""
    import os.path as osp
    import time
    from shutil import copyfile
    from torchdet3d.builders import (build_loader, build_model, build_loss,
                                        build_optimizer, build_scheduler)
    from torchdet3d.evaluation import Evaluator
    from torchdet3d.losses import LossManager
    from torchdet3d.trainer import Trainer
    from torchdet3d.utils import read_py_config, Logger, set_random_seed, check_isfile, resume_from
    SummaryWriter = _Writer
    args = types.SimpleNamespace(config=config_path, device=device, wo_saving_checkpoint=wo_saving_checkpoint)
    cfg = read_py_config(args.config)
    log_name = 'train.log' if cfg.regime.type == 'training' else 'test.log'
    log_name += time.strftime('-%Y-%m-%d-%H-%M-%S')
    old_stdout = sys.stdout
    sys.stdout = Logger(osp.join(cfg.output_dir, log_name))
    try:
        copyfile(args.config, osp.join(cfg.output_dir, 'dumped_config.py'))
        set_random_seed(cfg.utils.random_seeds)
        net = build_model(cfg)
        net.to(args.device)
        optimizer = build_optimizer(cfg, net)
        scheduler = build_scheduler(cfg, optimizer)
        if cfg.model.resume:
            if check_isfile(cfg.model.resume):
                start_epoch = resume_from(net, cfg.model.resume, optimizer=optimizer, scheduler=scheduler)
            else:
                raise RuntimeError("the checkpoint isn't found ot can't be loaded!")
        else:
            start_epoch = 0
        if (torch.cuda.is_available() and args.device == 'cuda' and cfg.data_parallel.use_parallel):
            net = torch.nn.DataParallel(net, **cfg.data_parallel.parallel_params)
        criterions = build_loss(cfg)
        loss_manager = LossManager(criterions, cfg.loss.coeffs, cfg.loss.alwa)
        train_loader, val_loader, test_loader = build_loader(cfg)
        writer = SummaryWriter(cfg.output_dir)
        train_step = (start_epoch - 1) * len(train_loader) if start_epoch > 1 else 0
        trainer = Trainer(model=net, train_loader=train_loader, optimizer=optimizer, scheduler=scheduler,
                          loss_manager=loss_manager, writer=writer, max_epoch=cfg.data.max_epochs,
                          log_path=cfg.output_dir, device=args.device, save_chkpt=args.wo_saving_checkpoint,
                          debug=cfg.utils.debug_mode, debug_steps=cfg.utils.debug_steps, save_freq=cfg.utils.save_freq,
                          print_freq=cfg.utils.print_freq, train_step=train_step)
        evaluator = Evaluator(model=net, val_loader=val_loader, test_loader=test_loader, cfg=cfg, writer=writer,
                              device=args.device, max_epoch=cfg.data.max_epochs, path_to_save_imgs=cfg.output_dir,
                              debug=cfg.utils.debug_mode, debug_steps=cfg.utils.debug_steps)
        if cfg.regime.type == "evaluation":
            evaluator.run_eval_pipe(cfg.regime.vis_only)
        else:
            assert cfg.regime.type == "training"
            if cfg.model.resume:
                evaluator.val()
            for epoch in range(start_epoch, cfg.data.max_epochs):
                is_last_epoch = epoch == cfg.data.max_epochs - 1
                trainer.train(epoch, is_last_epoch)
                if epoch % cfg.utils.eval_freq == 0 or is_last_epoch:
                    evaluator.val(epoch, is_last_epoch)
            evaluator.visual_test()
    finally:
        sys.stdout.close()
        sys.stdout = old_stdout
    return cfg, writer, net


@pytest.mark.gpu
def test_main_py_flow_replayed_on_synthetic_crops(tmp_path):
    out = tmp_path / 'log'
    out.mkdir()
    cfgp = tmp_path / 'cfg_train.py'
    cfgp.write_text(CONFIG % (str(out), 'training'))
    cfg, writer, net = _replay_main(str(cfgp))
    files = os.listdir(out)
    assert 'dumped_config.py' in files and 'snap_1.pth' in files                 # last epoch checkpoint (train.py:110)
    assert any(f.startswith('train.log-') for f in files)
    log = open(os.path.join(out, [f for f in files if f.startswith('train.log-')][0])).read()
    assert 'Computed val metrics' in log and 'image №' in log and 'saving checkpoint' in log
    tags = {t for t, _, _ in writer.scalars}
    assert {'Train/loss', 'Train/ADD', 'Train/SADD', 'Train/ACC', 'Val/ADD', 'Val/SADD', 'Val/ACC'} <= tags
    assert all(v == v for _, v, _ in writer.scalars)                             # no NaN
    assert sum(f.endswith('_predicted.npy') for f in files) == 10                # visual_test: num_samples = 10
    # resume + evaluation regime from the checkpoint just written (main.py:52-57, 96-97)
    cfge = tmp_path / 'cfg_eval.py'
    cfge.write_text((CONFIG % (str(out), 'evaluation')).replace("resume=''", f"resume='{out}/snap_1.pth'"))
    cfg2, _, net2 = _replay_main(str(cfge))
    a, b = net.state_dict(), net2.state_dict()
    assert all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)
