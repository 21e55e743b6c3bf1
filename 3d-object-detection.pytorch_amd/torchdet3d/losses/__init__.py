from .regression_losses import (LossManager, DiagLoss, ADD_loss, WingLoss, L1Loss, MSELoss, SmoothL1Loss,
                                CrossEntropyLoss, compute_diag)
