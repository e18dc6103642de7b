"""CPU: the C-ABI shared library builds, loads, and exports every symbol include/unigeo_hip.h (the drop-in boundary) and
include/unigeo_hip_test.h (test / tuning entry points) declare
(no compute calls - there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(names=("unigeo_hip.h", "unigeo_hip_test.h")):
    out = set()
    for n in names:
        txt = open(os.path.join(ROOT, "include", n)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"\b(ug_[a-z0-9_]+)\s*\(", txt))
    return sorted(out)


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


def test_product_header_is_the_boundary_only():
    """The product header carries what INTEGRATION.md section 2 maps to the reference - no op-level / bench / tuning entry points."""
    prod = header_symbols(("unigeo_hip.h",))
    assert not [s for s in prod if s.startswith(("ug_op_", "ug_bench_", "ug_tune_"))]
    for need in ("ug_create", "ug_load_tensor", "ug_dc_set_inputs", "ug_dc_run", "ug_dc_get_outputs", "ug_sn_run", "ug_last_error", "ug_destroy"):
        assert need in prod


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
