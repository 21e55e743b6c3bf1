"""Host-side glue of the hot path's callers, behaviour of torchdet3d/utils/utils.py restated without its absent
third-party imports (cv2, addict, objectron.graphics): seeds (:24-31), checkpoint save / load / resume
(:56-64, :86-112, :127-208), python-file config reader (:66-84), `put_on_device` (:242-245), `AverageMeter`
(:272-287)."""
import importlib.util
import os
import os.path as osp
import random
from collections import OrderedDict

import numpy as np
import torch

OBJECTRON_CLASSES = ('bike', 'book', 'bottle', 'cereal_box', 'camera', 'chair', 'cup', 'laptop', 'shoe')


class AttrDict(dict):
    """Stand-in for addict.Dict as the reference uses it: attribute access, missing keys -> empty (falsy) dict."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        return self[k] if k in self else AttrDict()

    def __setattr__(self, k, v):
        self[k] = v


def read_py_config(filename):
    """utils.py:66-84: import a python file as a module, return its public globals as an attribute dict."""
    filename = osp.abspath(osp.expanduser(filename))
    check_isfile(filename)
    assert filename.endswith('.py')
    module_name = osp.basename(filename)[:-3]
    if '.' in module_name:
        raise ValueError('Dots are not allowed in config file path.')
    spec = importlib.util.spec_from_file_location(module_name, filename)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return AttrDict({k: v for k, v in mod.__dict__.items() if not k.startswith('__')})


def check_isfile(fpath):
    if not osp.isfile(fpath):
        raise RuntimeError(f'No file found at "{fpath}"')
    return True


def set_random_seed(seed, deterministic=False):
    """utils.py:24-31."""
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)


def save_snap(model, optimizer, scheduler, epoch, log_path):
    """utils.py:56-64 (`snap_{epoch}.pth` with state_dict / optimizer / scheduler / epoch)."""
    snap = {'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict(),
            'scheduler': scheduler.state_dict() if scheduler is not None else None, 'epoch': epoch}
    os.makedirs(log_path, exist_ok=True)
    name = osp.join(log_path, f'snap_{epoch}.pth')
    print(f'==> saving checkpoint to {name}')
    torch.save(snap, name)


def load_checkpoint(fpath, map_location=None):
    """utils.py:86-112."""
    if fpath is None:
        raise ValueError('File path is None')
    check_isfile(fpath)
    return torch.load(fpath, map_location=map_location or 'cpu', weights_only=False)


def load_pretrained_weights(model, file_path='', pretrained_dict=None, extra_prefix=''):
    """utils.py:127-183: strip `module.`, optional extra prefix, keep only name+shape matches; raise if none."""
    ckpt = load_checkpoint(file_path) if file_path else pretrained_dict
    state = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    model_dict = model.state_dict()
    new, matched, discarded = OrderedDict(), [], []
    for k, v in state.items():
        if k.startswith('module.'):
            k = k[7:]
        k = extra_prefix + k
        if k in model_dict and tuple(model_dict[k].shape) == tuple(v.shape):
            new[k] = v
            matched.append(k)
        else:
            discarded.append(k)
    if not matched:
        raise RuntimeError(f'The pretrained weights "{file_path}" cannot be loaded, check the key names manually')
    model_dict.update(new)
    model.load_state_dict(model_dict)
    if discarded:
        print(f'** The following layers are discarded due to unmatched keys or layer size: {discarded[:8]}...')
    return model


def resume_from(model, chkpt_path, optimizer=None, scheduler=None):
    """utils.py:185-208: returns the epoch to start from."""
    print(f'Loading checkpoint from "{chkpt_path}"')
    ckpt = load_checkpoint(chkpt_path)
    load_pretrained_weights(model, pretrained_dict=ckpt['state_dict'] if 'state_dict' in ckpt else ckpt)
    if optimizer is not None and ckpt.get('optimizer') is not None:
        optimizer.load_state_dict(ckpt['optimizer'])
    if scheduler is not None and ckpt.get('scheduler') is not None:
        scheduler.load_state_dict(ckpt['scheduler'])
    return ckpt.get('epoch', -1) + 1


def put_on_device(items, device):
    """utils.py:242-245."""
    return [item.to(device) for item in items]


class AverageMeter:
    """utils.py:272-287."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
