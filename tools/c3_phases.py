#!/usr/bin/env python
"""Sweep / rounds time of the config-3 batch per node, from a -DPCP_ABLATE=64 profiling build (phase timers in the counters).
usage: PCP_HIP_LIB=.../lib64.so python tools/c3_phases.py [nodes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import workloads as W
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
V, P = 50_000, 500_000
props, lb, ub, sol = W.planted_binary_csp(0xC3, V, P)
L, U = W.unit_narrowing_prefix(0xC3 + 1, lb, ub, sol, nodes)
ctx = E.Context(0); ctx.set_model(V, props); ctx.set_hull(0, 999)
dev = torch.device("cuda:0")
t_lb, t_ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
st = torch.zeros(nodes, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    l2, u2 = t_lb.clone(), t_ub.clone()
    ctx.stats_reset(stream)
    ctx.propagate_device(nodes, l2, u2, l2, u2, None, None, st, stream)
    s = ctx.stats_read(stream)
print(ctx.last_plan())
print(f"kernel {ctx.last_kernel_ms():.2f} ms for {nodes} nodes; per node: sweep avg {s['steps3'] / nodes / 100:.0f} us (max {s['failed_nodes'] / 100:.0f}), rounds+tail avg {s['narrowings'] / nodes / 100:.0f} us (max {s['waves'] / 100:.0f})")
