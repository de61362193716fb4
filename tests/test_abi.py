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
    assert L.pcp_abi_version() == 8


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


def _c_layout():
    """sizeof / offsetof of every struct of include/pcp_hip.h, from a C probe compiled against the header itself."""
    import subprocess
    import tempfile
    header = open(os.path.join(ROOT, "include", "pcp_hip.h")).read()
    structs = {}
    for body, name in re.findall(r"typedef struct(?:\s+\w+)?\s*\{(.*?)\}\s*(pcp_\w+)\s*;", header, flags=re.S):
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                m = re.search(r"(\w+)\s*(?:\[\s*\w+\s*\])?\s*$", part.strip())
                fields.append(m.group(1))
        structs[name] = fields
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "pcp_hip.h"', "int main(void) {"]
    for name, fields in structs.items():
        src.append(f'  printf("{name} %zu", sizeof({name}));')
        for f in fields:
            src.append(f'  printf(" {f}:%zu", offsetof({name}, {f}));')
        src.append('  printf("\\n");')
    src.append("  return 0; }")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "probe.c")
        open(c, "w").write("\n".join(src))
        exe = os.path.join(d, "probe")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe], text=True)
    layout = {}
    for line in out.splitlines():
        parts = line.split()
        layout[parts[0]] = (int(parts[1]), [(p.split(":")[0], int(p.split(":")[1])) for p in parts[2:]])
    return layout


def test_struct_layouts_agree_between_header_ctypes_and_the_rust_binding():
    """pcp_dfs_state grew a trailing `dirty` pointer in ABI v7 and the Rust mirror did not.  The header is the authority: a C
    probe compiled against it gives every struct's size and field offsets; the ctypes mirrors must match them field by field, and the Rust
    `#[repr(C)]` structs of integration/pcp-hip-sys must list the same fields in the same order."""
    layout = _c_layout()
    assert layout["pcp_dfs_state"][0] == 72 and layout["pcp_device_batch"][0] == 88
    mirrors = {"pcp_dfs_state": E.DfsState, "pcp_device_batch": E.DeviceBatch, "pcp_forest_state": E.ForestState, "pcp_stats": E.PcpStats,
               "pcp_plan": E.PcpPlan}
    for name, cls in mirrors.items():
        size, fields = layout[name]
        assert ctypes.sizeof(cls) == size, name
        assert [(f, getattr(cls, f).offset) for f, _ in cls._fields_] == fields, name
    rust = open(os.path.join(ROOT, "integration", "pcp-hip-sys", "src", "lib.rs")).read()
    for name, (size, fields) in layout.items():
        m = re.search(r"pub struct %s\s*\{(.*?)\n\}" % name, rust, flags=re.S)
        if m is None:
            continue
        body = re.sub(r"//[^\n]*", "", m.group(1))
        rust_fields = [f.rstrip("_") for f in re.findall(r"pub\s+(\w+)\s*:", body)]  # (`type` is a Rust keyword: `type_`)
        assert rust_fields == [f for f, _ in fields], (name, rust_fields, [f for f, _ in fields])
    assert re.search(r"pub struct pcp_dfs_state", rust)
