#!/usr/bin/env python
"""Profiling / experiment build of the library: python tools/build_variant.py <PCP_ABLATE value> <out.so> [extra hipcc flags...]
(same parallel build as __graft_entry__.build_hip_lib, with -DPCP_ABLATE=<value>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build_hip_lib(sys.argv[2], [f"-DPCP_ABLATE={int(sys.argv[1])}", *sys.argv[3:]])
