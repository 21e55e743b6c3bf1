"""Name -> LR scheduler (torchdet3d/builders/scheduler_builder.py:3-25)."""
import torch

AVAILABLE_SCHEDS = ['cosine', 'exp', 'stepLR', 'multistepLR']


def build_scheduler(cfg, optimizer):
    if not cfg.scheduler.name:
        return None
    assert cfg.scheduler.name in AVAILABLE_SCHEDS
    if cfg.scheduler.name == 'cosine':
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=cfg.data.max_epochs, eta_min=5e-6)
    if cfg.scheduler.name == 'exp':
        return torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=cfg.scheduler.exp_gamma)
    if cfg.scheduler.name == 'stepLR':
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=cfg.scheduler.steps[0], gamma=cfg.scheduler.gamma)
    return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=cfg.scheduler.steps, gamma=cfg.scheduler.gamma)
