import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, '3d-object-detection.pytorch_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


# ---- T3D_GUARD=1 (with PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1): every device tensor the kernel
# tests create sits at the END of its own 2 MB-granular hipMalloc, so a kernel reading or writing past a tensor hits
# unmapped memory and faults instead of silently touching a neighbour (tools/gpu_guard.sh).  Off by default.
def _guard_patches():
    import torch
    G = 2 << 20
    o_empty, o_cuda, o_to = torch.empty, torch.Tensor.cuda, torch.Tensor.to

    def is_cuda_dev(d):
        return d is not None and torch.device(d).type == 'cuda'

    def tail(shape, dtype):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        n = 1
        for v in shape:
            n *= int(v)
        esz = torch.empty(0, dtype=dtype).element_size()
        nbytes = (n * esz + 15) // 16 * 16
        tot = max(G, (nbytes + G - 1) // G * G)
        buf = o_empty(tot, dtype=torch.uint8, device='cuda')
        return buf[tot - nbytes:tot - nbytes + n * esz].view(dtype).view(shape)

    def g_empty(*shape, dtype=None, device=None, **kw):
        if is_cuda_dev(device) and not kw:
            return tail(shape, dtype or torch.get_default_dtype())
        return o_empty(*shape, dtype=dtype, device=device, **kw)

    def mk(fill):
        def f(*shape, dtype=None, device=None, **kw):
            if is_cuda_dev(device) and not kw:
                t = tail(shape, dtype or torch.get_default_dtype())
                t.fill_(fill)
                return t
            return {0: o_zeros, 1: o_ones}[fill](*shape, dtype=dtype, device=device, **kw)
        return f
    o_zeros, o_ones = torch.zeros, torch.ones

    def g_cuda(self, *a, **k):
        if self.is_cuda or self.numel() == 0:
            return o_cuda(self, *a, **k)
        t = tail(self.shape, self.dtype)
        t.copy_(self.contiguous())
        return t

    def g_to(self, *a, **k):
        dev = k.get('device', next((v for v in a if isinstance(v, (str, torch.device))), None))
        if self.is_cuda or not is_cuda_dev(dev) or self.numel() == 0:
            return o_to(self, *a, **k)
        dt = k.get('dtype', next((v for v in a if isinstance(v, torch.dtype)), self.dtype))
        src = o_to(self, dt).contiguous()
        t = tail(src.shape, dt)
        t.copy_(src)
        return t
    return {'empty': g_empty, 'zeros': mk(0), 'ones': mk(1), 'cuda': g_cuda, 'to': g_to}


@pytest.fixture(autouse=True)
def _t3d_guard(monkeypatch):
    if os.environ.get('T3D_GUARD'):
        import torch
        p = _guard_patches()
        monkeypatch.setattr(torch, 'empty', p['empty'])
        monkeypatch.setattr(torch, 'zeros', p['zeros'])
        monkeypatch.setattr(torch, 'ones', p['ones'])
        monkeypatch.setattr(torch.Tensor, 'cuda', p['cuda'])
        monkeypatch.setattr(torch.Tensor, 'to', p['to'])
    yield
