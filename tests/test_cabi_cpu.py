"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/vdjx.h declares,
and the product path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vdjer_amd import _lib
    header = open(os.path.join(ROOT, "include", "vdjx.h")).read()
    declared = sorted(set(re.findall(r"\b(vdjx_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.vdjx_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from vdjer_amd import api
    with pytest.raises(api.VdjxError):
        api.Context(0)


def test_product_never_imports_oracle():
    """nothing under vdjer_amd/ (nor bench-independent product code) imports, includes, links or dlopens oracle/"""
    pkg = os.path.join(ROOT, "vdjer_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|#\s*include\s*[<\"].*oracle|liboracle|vdjx_oracle|oracle/_ref|vdjer_ref", re.M)
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".c")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn)).read()
                assert not bad.search(txt), os.path.join(dp, fn)


def test_struct_layouts_match_header():
    from vdjer_amd import _lib
    assert ctypes.sizeof(_lib.Pair) == 20
    assert ctypes.sizeof(_lib.CovParams) == 28
