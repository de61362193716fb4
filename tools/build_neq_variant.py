#!/usr/bin/env python
"""Experiment build: pcp_neq.hip compiled with extra flags, every other object from the product build.
usage: python tools/build_neq_variant.py <name> [-DFLAG=..]...   -> pcp_amd/libpcp_hip_<name>.so (select with PCP_HIP_LIB)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
name = sys.argv[1]
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
bdir = os.path.join(ROOT, "build", "libpcp_hip.so")
pdir = os.path.join(ROOT, "build", "variant_" + name); os.makedirs(pdir, exist_ok=True)
o = os.path.join(pdir, "neq.o")
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-unused-function", *sys.argv[2:], "-c",
                os.path.join(ROOT, "pcp_amd/csrc/pcp_neq.hip"), "-o", o], check=True, cwd=ROOT)
objs = [o if n == "neq.o" else os.path.join(bdir, n) for n in g.HIP_OBJECTS]
out = os.path.join(ROOT, "pcp_amd", f"libpcp_hip_{name}.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True, cwd=ROOT)
print("built", out)
