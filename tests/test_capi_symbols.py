"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/zkdl_amd.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkdl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from zkdl_amd import capi
    lib = capi.load_library()
    syms = declared_symbols()
    assert len(syms) > 80
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == syms          # the Python binding tracks the header


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from zkdl_amd.capi import Zk
    with pytest.raises(RuntimeError):
        Zk(0)


def test_product_never_imports_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "zkdl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle_py" not in src and "zkdl_oracle" not in src and "liboracle" not in src, os.path.join(d, f)
