"""Deterministic, framework-independent weight fill (test infrastructure).

One numpy PCG64 stream per state-dict key (seeded by crc32 of the key), so a
17 MB weight file never has to be committed: the reference (in
gen_golden.py), the oracle and the product model all get the identical values
via load_state_dict.  Recipe from SURVEY.md section 8(c).
"""
import zlib

import numpy as np
import torch

from .model import state_dict_shapes


def fill(key, shape, seed=0):
    rng = np.random.default_rng(zlib.crc32(key.encode()) + 7919 * seed)
    if key.endswith('num_batches_tracked'):
        return np.zeros((), np.int64)
    if key.endswith('running_mean'):
        return rng.uniform(-0.1, 0.1, shape).astype(np.float32)
    if key.endswith('running_var'):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) == 1 and key.endswith('.weight'):          # BN gamma
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) == 1:                                        # BN beta / linear bias
        return rng.uniform(-0.2, 0.2, shape).astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    a = np.sqrt(3.0 / fan_in)      # second-moment preserving: keeps eval-mode activations O(1)
    return rng.uniform(-a, a, shape).astype(np.float32)


def make_state_dict(name, num_classes=9, seed=0):
    return {k: torch.from_numpy(np.array(fill(k, s, seed)))
            for k, s in state_dict_shapes(name, num_classes).items()}


def make_inputs(B, H, W, num_classes=9, seed=0):
    """Synthetic crops per the input contract."""
    rng = np.random.default_rng(1000 + seed)
    imgs = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    gt_kp = rng.uniform(0, 1, (B, 9, 2)).astype(np.float32)
    cats = (np.arange(B) * 4 + seed) % 9
    if num_classes > 1:
        cats = cats % num_classes
    return torch.from_numpy(imgs), torch.from_numpy(gt_kp), torch.from_numpy(cats.astype(np.int64))
