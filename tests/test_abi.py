"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mppgpu.h
declares (and nothing undeclared), and refuses to run without a GPU (no CPU fallback)."""
import ctypes as ct
import os
import re
import subprocess

import pytest

from mpp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mppgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mppgpu_\w+)\s*\(", text)))


def test_header_matches_python_signatures():
    assert _header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    L = _lib.lib()
    for name in _header_symbols():
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.SO_PATH], text=True)
    exported = sorted(set(re.findall(r"\bT (mppgpu_\w+)", out)))
    assert exported == _header_symbols()


def test_no_cpu_fallback_without_a_gpu():
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    L = _lib.lib()
    h = ct.c_void_p()
    rc = L.mppgpu_create(0, ct.byref(h))
    if rc == 0:  # a GPU exists after all
        L.mppgpu_destroy(h)
        pytest.skip("a GPU is present")
    assert not h.value
    assert b"no usable HIP device" in L.mppgpu_last_error(None)


def test_product_package_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under mpp_amd/ may import, link or execute it"""
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "mpp_amd")):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "vsfm_oracle" not in text and "oracle/" not in text, os.path.join(dirpath, f)
    deps = subprocess.check_output(["ldd", _lib.SO_PATH], text=True)
    assert "oracle" not in deps


def test_fortran_module_binds_every_entry_point():
    """fortran/mppgpu_mod.F90 declares a bind(C) interface for every function include/mppgpu.h declares (the binding an
    MPP maintainer links against), and nothing else"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "mppgpu.h")).read()
    mod = open(os.path.join(root, "fortran", "mppgpu_mod.F90")).read()
    declared = set(re.findall(r"\b(mppgpu_\w+)\s*\(", hdr))
    bound = set(re.findall(r"bind\(C, name='(mppgpu_\w+)'\)", mod))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
