"""Host-side glue of the hot path's callers, behaviour of torchdet3d/utils/utils.py restated without its absent
third-party imports (cv2, addict, objectron.graphics): seeds (:24-31), `check_isfile` / `mkdir_if_missing`
(:33-54), checkpoint save / load / resume (:56-64, :86-112, :127-208), python-file config reader (:66-84),
`put_on_device` (:242-245), `AverageMeter` (:272-287), `Logger` (:289-333)."""
import errno
import importlib.util
import os
import os.path as osp
import random
import sys
import warnings
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
    if not check_isfile(filename):
        raise RuntimeError("config not found")
    assert filename.endswith('.py')
    module_name = osp.basename(filename)[:-3]
    if '.' in module_name:
        raise ValueError('Dots are not allowed in config file path.')
    spec = importlib.util.spec_from_file_location(module_name, filename)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return AttrDict({k: v for k, v in mod.__dict__.items() if not k.startswith('__')})


def check_isfile(fpath):
    """utils.py:33-45: warns (does not raise) and returns whether `fpath` is a file."""
    isfile = osp.isfile(fpath)
    if not isfile:
        warnings.warn(f'No file found at "{fpath}"')
    return isfile


def mkdir_if_missing(dirname):
    """utils.py:47-54."""
    if not osp.exists(dirname):
        try:
            os.makedirs(dirname)
        except OSError as e:
            if e.errno != errno.EEXIST:
                raise


def set_random_seed(seed, deterministic=False):
    """utils.py:24-31."""
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)


def save_snap(model, optimizer, scheduler, epoch, log_path):
    """utils.py:56-64 (`snap_{epoch}.pth` with state_dict / optimizer / scheduler / epoch).  One process per GPU: rank 0
    writes (every rank holds the same weights and optimizer state after the gradient exchange), the others wait at the
    barrier so that nobody reads a half-written file."""
    from ..parallel import barrier, is_main
    if not is_main():
        barrier()
        return
    try:
        snap = {'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict(),
                'scheduler': scheduler.state_dict() if scheduler is not None else None, 'epoch': epoch}
        mkdir_if_missing(log_path)
        name = osp.join(log_path, f'snap_{epoch}.pth')
        print(f'==> saving checkpoint to {name}')
        torch.save(snap, name)
    finally:
        barrier()          # (also when the write fails: the other ranks are waiting in theirs)


def load_checkpoint(fpath, map_location=None):
    """utils.py:86-112."""
    if fpath is None:
        raise ValueError('File path is None')
    if not osp.exists(fpath):
        raise FileNotFoundError(f'File is not found at "{fpath}"')
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


class Logger:
    """utils.py:289-333: tee of the console into a text file (`sys.stdout = Logger(path)`, scripts/main.py:39);
    creates the directory of `fpath`, `flush` also fsyncs the file.  One process per GPU: only rank 0 opens the file (N
    processes truncating and writing one `train.log` is what an unchanged main.py would otherwise do); the other ranks keep
    their console."""

    def __init__(self, fpath=None):
        from ..parallel import launch_rank
        self.console = sys.stdout
        self.file = None
        if fpath is not None and launch_rank() == 0:
            mkdir_if_missing(osp.dirname(fpath))
            self.file = open(fpath, 'w')      # noqa: SIM115  (lives as long as the logger)

    def __del__(self):
        self.close()

    def __enter__(self):
        pass

    def __exit__(self, *args):
        self.close()

    def write(self, msg):
        self.console.write(msg)
        if self.file is not None and not self.file.closed:
            self.file.write(msg)

    def flush(self):
        self.console.flush()
        if self.file is not None:
            self.file.flush()
            os.fsync(self.file.fileno())

    def close(self):
        # the reference also closes the console stream here (:330); closing the process's stdout from a destructor
        # is an accident, not a contract, so only the file is closed
        if self.file is not None:
            self.file.close()
            self.file = None
