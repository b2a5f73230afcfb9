"""The C-ABI library loads and exports exactly the symbols include/fsr_hip.h declares (no compute, no GPU)."""
import ctypes
import os
import re

from backend import L, ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "fsr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fsr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(L.SIGNATURES), "include/fsr_hip.h and _lib.SIGNATURES list different entry points"


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.LIB_PATH), "libfsr_hip.so is not built (python fast-srgan_amd/build.py)"
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.fsr_version.restype = ctypes.c_int
    assert lib.fsr_version() == L.ABI_VERSION
    # argument validation works without a device: nothing is launched for a rejected call
    lib.fsr_last_error.restype = ctypes.c_char_p
    assert lib.fsr_conv3x3(None, None, None, None, None, None, None, ctypes.c_float(0), None, None, None, None, None) < 0
    assert b"null" in lib.fsr_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    L._install_for_testing(None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libfsr_hip.so")
    with pytest.raises(L.FsrError):
        L.lib()
