"""The C-ABI shared library loads without a GPU and exports every symbol include/mv2d_hip.h declares; argument
validation errors are reported through the return code + mv2d_last_error (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from mv2d_amd import _lib
    return _lib.load()


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'mv2d_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mv2d_\w+)\s*\(', src)))


def test_header_symbols_are_exported(lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/mv2d_hip.h but not exported'
    from mv2d_amd._lib import SIGNATURES
    assert set(SIGNATURES) == set(names), set(SIGNATURES) ^ set(names)


def test_error_reporting_without_gpu(lib):
    assert lib.mv2d_abi_version() == 6
    rc = lib.mv2d_gemm_f32(None, None, 0, None, None, 1, 1, 32, 32, 32, 1, 0, 1.0, 0.0, None, 0, 1, 0, 1, 0, 0, 0, 0, None)
    assert rc == -1
    assert b'mv2d_gemm_f32' in lib.mv2d_last_error()
    buf = ctypes.create_string_buffer(64)
    rc = lib.mv2d_row_ln(None, 0, 0, None, None, None, None, 0, None, None, None, None, None, None, 0, 1e-5, 0, None)
    assert rc == -1 and b'mv2d_row_ln' in lib.mv2d_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mv2d_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(_lib.Mv2dHipError):
        _lib.load(str(tmp_path / 'nope.so'))
