"""CPU-only: libvslnet_hip.so builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every entry point
declared in include/vslnet_hip.h.  No compute call is made here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'vslnet_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(vsl_[a-z_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from vslnet_amd import build
    lib = ctypes.CDLL(build.build())
    names = _declared()
    assert len(names) >= 13, names
    for n in names:
        assert hasattr(lib, n), 'missing export: ' + n
    from vslnet_amd.engine import ABI_SYMBOLS
    assert sorted(ABI_SYMBOLS) == names


def test_last_error_is_callable_without_gpu():
    from vslnet_amd import build
    lib = ctypes.CDLL(build.build())
    lib.vsl_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.vsl_last_error(), bytes)


def test_engine_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from vslnet_amd.engine import Engine, VslError
    from vslnet_amd.synthetic import make_configs
    with pytest.raises(VslError, match='no CPU fallback'):
        Engine(make_configs())
