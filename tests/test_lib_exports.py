"""CPU: the C-ABI shared library builds, loads, and exports every symbol include/unigeo_hip.h declares
(no compute calls - there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "unigeo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ug_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from unigeo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load_library()
    syms = header_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_lib.EXPORTS) <= set(syms), set(_lib.EXPORTS) - set(syms)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from unigeo_amd._lib import Engine
    with pytest.raises(RuntimeError):
        Engine(0, 1 << 20, 1 << 20)


def test_product_package_never_imports_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "unigeo_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
