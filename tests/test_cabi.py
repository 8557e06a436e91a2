"""CPU: the C-ABI library loads and exports every entry point include/mafed_hip.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mafed_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(mafed_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    from mafed_amd import _lib
    lib = _lib.load()
    decl = declared_functions()
    assert len(decl) >= 25
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in include/mafed_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args, binding has {len(_lib.SIGNATURES[name][1])}"
    for name in _lib.SIGNATURES:
        assert name in decl, f"{name} bound but not declared in the header"
    assert lib.mafed_version() >= 100


def test_missing_library_fails_loudly(monkeypatch):
    from mafed_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmafed_hip.so")
    with pytest.raises(_lib.MafedHipError):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from mafed_amd import _lib, ops
    with pytest.raises(_lib.MafedHipError):
        ops.gelu(torch.zeros(8))
