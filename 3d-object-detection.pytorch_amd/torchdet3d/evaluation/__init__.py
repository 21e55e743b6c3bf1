from .metrics import (compute_average_distance, compute_accuracy, compute_metrics_per_cls, compute_2d_based_iou, iou3d_per_sample)
from .evaluate import Evaluator
