"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/emo_hip.h declares; argument validation errors surface as Python exceptions (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'emo_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(emo_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from emo_disentanger_amd import _lib
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_lib.lib, n), 'libemo_hip.so does not export %s' % n
        assert n in _lib._SIG, 'ctypes signature missing for %s' % n
    assert set(_lib._SIG) == set(names)
    assert _lib.lib.emo_version() >= 100


def test_invalid_arguments_raise_without_a_gpu():
    from emo_disentanger_amd import _lib
    rc = _lib.lib.emo_gemm(None, 0, 8, None, 0, 8, None, 8, 4, 4, 4, 1, 1, 0, None, None)
    assert rc == -1 and b'null pointer' in _lib.lib.emo_last_error()
    with pytest.raises(_lib.EmoError):
        _lib.check(rc)
    import torch
    with pytest.raises(_lib.EmoError):
        _lib.ptr(torch.zeros(3))           # CPU tensor: the product path has no CPU fallback
    with pytest.raises(_lib.EmoError):
        _lib.dtype_code(torch.float16)


def test_epilogue_struct_mirror_has_the_library_size():
    # the ctypes mirror of emo_epilogue_t (fields are appended as epilogue features are added) must match the struct the library was built with
    import ctypes
    from emo_disentanger_amd import _lib
    assert _lib.lib.emo_epilogue_size() == ctypes.sizeof(_lib.Epilogue)

