"""MI355X-native 3D-box keypoint-regression hot path behind the reference's own Python API
(sovrasov/3d-object-detection.pytorch: torchdet3d.builders / losses / trainer / evaluation / utils)."""
from . import builders, evaluation, losses, trainer, utils  # noqa: F401
