"""CPU checks of the C-ABI boundary: the library loads and exports exactly the symbols
include/t3d.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 't3d.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(t3d_\w+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from torchdet3d import _native as N
    assert os.path.exists(N.LIB_PATH), 'run `python __graft_entry__.py` (build()) first'
    lib = ctypes.CDLL(N.LIB_PATH)
    names = _declared()
    assert names, 'no declarations parsed'
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/t3d.h but not exported'
    assert sorted(N.SIGNATURES) == names, 'ctypes SIGNATURES and include/t3d.h disagree'


def test_version_call():
    from torchdet3d import _native as N
    assert N.lib().t3d_version() >= 1
