"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/pcp_hip.h declares, and the product path refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import pcp_amd.engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    if not os.path.exists(E.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return E.load_library()


def test_header_symbols_are_exported():
    L = _built()
    header = open(os.path.join(ROOT, "include", "pcp_hip.h")).read()
    declared = set(re.findall(r"\b(pcp_[a-z_0-9]+)\s*\(", header))
    declared -= {"pcp_ctx"}
    assert declared == set(E.ABI_SYMBOLS), declared ^ set(E.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), f"libpcp_hip.so does not export {name}"
    assert L.pcp_abi_version() == 7


def test_prop_struct_layout_matches_header():
    import numpy as np
    from pcp_amd.model import PROP_DTYPE
    assert PROP_DTYPE.itemsize == 32
    assert PROP_DTYPE.fields["var"][1] == 8 and PROP_DTYPE.fields["off"][1] == 20 and PROP_DTYPE.fields["group"][1] == 4
    assert ctypes.sizeof(E.PcpStats) == 64 and ctypes.sizeof(E.DeviceBatch) == 88 and ctypes.sizeof(E.PcpPlan) == 52


def test_fails_loudly_without_gpu():
    import torch
    _built()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(E.EngineUnavailable):
        E.Context(0)


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pcp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "pcp_oracle" not in src, os.path.join(dirpath, f)
