#!/usr/bin/env python
"""Profiling build of the library: pcp_neq.hip compiled with -DPCP_NEQ_PROFILE=1 (phase timers, per-wavefront event trace), every other
object taken from the product build.  Writes pcp_amd/libpcp_hip_prof.so; select it with PCP_HIP_LIB=pcp_amd/libpcp_hip_prof.so."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
bdir = os.path.join(ROOT, "build", "libpcp_hip.so")
pdir = os.path.join(ROOT, "build", "profile"); os.makedirs(pdir, exist_ok=True)
o = os.path.join(pdir, "neq.o")
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-DPCP_NEQ_PROFILE=1", *sys.argv[1:], "-c",
                os.path.join(ROOT, "pcp_amd/csrc/pcp_neq.hip"), "-o", o], check=True, cwd=ROOT)
objs = [o if name == "neq.o" else os.path.join(bdir, name) for name in g.HIP_OBJECTS]
out = os.path.join(ROOT, "pcp_amd", "libpcp_hip_prof.so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True, cwd=ROOT)
print("built", out)
