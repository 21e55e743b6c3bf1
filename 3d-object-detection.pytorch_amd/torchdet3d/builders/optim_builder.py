"""Name -> torch.optim (torchdet3d/builders/optim_builder.py:3-19; 'adam' builds AdamW, :10-12).  The model
exposes ONE flat parameter (all weights, see models/engine.py), so every optimizer here is a single fused
elementwise update over ~2.4-4.4 M floats instead of ~190 small tensors."""
import torch

AVAILABLE_OPTIMS = ['sgd', 'rmsprop', 'adam', 'adadelta']


def build_optimizer(cfg, net):
    assert cfg.optim.name in AVAILABLE_OPTIMS
    params = list(net.parameters())
    if cfg.optim.name == 'adadelta':
        return torch.optim.Adadelta(params, lr=cfg.optim.lr, rho=cfg.optim.rho, weight_decay=cfg.optim.wd)
    if cfg.optim.name == 'adam':
        return torch.optim.AdamW(params, lr=cfg.optim.lr, betas=tuple(cfg.optim.betas), weight_decay=cfg.optim.wd)
    if cfg.optim.name == 'rmsprop':
        return torch.optim.RMSprop(params, lr=cfg.optim.lr, weight_decay=cfg.optim.wd, alpha=cfg.optim.alpha)
    return torch.optim.SGD(params, lr=cfg.optim.lr, weight_decay=cfg.optim.wd, momentum=cfg.optim.momentum,
                           nesterov=cfg.optim.nesterov)
