"""Loss names -> (regression criterions, class criterions), the two lists `LossManager` takes
(torchdet3d/builders/loss_builder.py:4-28 of the reference: same names, same grouping, criterions in the order the
config lists them)."""
from .. import losses as L

# name -> (group, constructor taking cfg.loss); group 0 = keypoint regression, 1 = classification
_TABLE = {
    'smoothl1': (0, lambda c: L.SmoothL1Loss(beta=c.smoothl1_beta)),
    'l1': (0, lambda c: L.L1Loss()),
    'cross_entropy': (1, lambda c: L.CrossEntropyLoss()),
    'diag_loss': (0, lambda c: L.DiagLoss()),
    'mse': (0, lambda c: L.MSELoss()),
    'add_loss': (0, lambda c: L.ADD_loss()),
    'wing': (0, lambda c: L.WingLoss(w=c.w, eps=c.eps)),
}
AVAILABLE_LOSS = list(_TABLE)


def build_loss(cfg):
    groups = ([], [])
    for name in cfg.loss.names:
        assert name in _TABLE, f'unknown loss {name!r}; available: {AVAILABLE_LOSS}'
        group, make = _TABLE[name]
        groups[group].append(make(cfg.loss))
    return groups
