"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/lhn.h declares."""
import os
import re

from litehandnet_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_loads():
    build.build_lib(verbose=False)
    L = _lib.lib()
    assert L.lhn_version() == 3


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "lhn.h")).read()
    declared = set(re.findall(r"\b(lhn_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_no_cpu_fallback():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from litehandnet_amd import heatmap
    with pytest.raises(_lib.LhnError):
        heatmap._get_max_preds(torch.zeros(1, 1, 8, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "litehandnet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn
            assert "import oracle" not in src and "from oracle" not in src, fn
