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


def _struct_fields(name):
    """(field, ctype) list of `typedef struct <name> {...}` in include/fsr_hip.h, in declaration order."""
    text = open(os.path.join(ROOT, "include", "fsr_hip.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for ty, names in re.findall(r"\b(int|float)\s+([a-z0-9_,\s]+);", body):
        out += [(n.strip(), ty) for n in names.split(",")]
    return out


def test_binding_structs_match_the_header():
    for cls, name in ((L.ConvDesc, "fsr_conv_desc"), (L.WgradDesc, "fsr_wgrad_desc")):
        want = _struct_fields(name)
        got = [(n, "float" if t is ctypes.c_float else "int") for n, t in cls._fields_]
        assert got == want, name


def test_integration_md_snippet_matches_the_header():
    """INTEGRATION.md shows the ctypes stub a maintainer of the reference would paste: its ConvDesc must have the
    header's fields in the header's order (round 3 shipped a snippet one field short of the struct)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\nimport ctypes, torch\n(.*?)```", text, flags=re.S).group(1)
    cls_src = re.search(r"(class ConvDesc\(ctypes\.Structure\):.*?)\nP = ", code, flags=re.S).group(1)
    ns = {"ctypes": ctypes}
    exec(cls_src, ns)
    got = [(n, "float" if t is ctypes.c_float else "int") for n, t in ns["ConvDesc"]._fields_]
    assert got == _struct_fields("fsr_conv_desc")
    assert ctypes.sizeof(ns["ConvDesc"]) == ctypes.sizeof(L.ConvDesc)
    # the constructor call of the snippet passes one value per field
    call = re.search(r"d = ConvDesc\((.*?)\)", code).group(1)
    assert len(call.split(",")) == len(got)
    # and the entry-point count / ABI version quoted in the text are the binding's
    m = re.search(r"all (\d+) entry points of ABI version (\d+)", text)
    assert (int(m.group(1)), int(m.group(2))) == (len(L.SIGNATURES), L.ABI_VERSION)
