#!/usr/bin/env python
"""Per-tile time of staging / sweep / wake-up rounds, from -DPCP_ABLATE=64 and =320 profiling builds (which report the phase
timers in the counters).  usage: PCP_HIP_LIB=.../lib64.so [PCP_ACTIVE=explicit] python tools/phase_times.py [nodes]
Regimes: the bench frontier (near the root), and the deep end of the stack after 500 / 3000-node depth-first dives."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd import workloads as W

n = 1000; nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
implicit = os.environ.get("PCP_ACTIVE", "implicit") == "implicit"
mode320 = os.environ.get("PCP_MODE320", "") == "1" or "320" in os.environ.get("PCP_HIP_LIB", "")
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
if os.environ.get('PCP_WORD_LEVEL'): ctx.set_option('word_level', int(os.environ['PCP_WORD_LEVEL'])); print('word_level', os.environ['PCP_WORD_LEVEL'])
dev = torch.device("cuda:0")
for D in [int(x) for x in os.environ.get('PCP_DIVES', '0,500,3000').split(',')]:
    ctx.set_option("nodes_per_block", 0)
    if D == 0:
        L, U, A = W.nqueens_frontier(ctx, n, nodes, 0, 8, implicit=implicit)
        lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
        act = None if A is None else torch.from_numpy(A.view(np.int64)).to(dev)
    else:
        lb, ub, act = W.nqueens_deep(ctx, n, D, nodes, implicit=implicit)
    N = lb.shape[0]
    ctx.set_option("nodes_per_block", 16)
    status = torch.zeros(N, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        l2, u2, a2 = lb.clone(), ub.clone(), None if act is None else act.clone()
        ctx.stats_reset(stream)
        ctx.propagate_device(N, l2, u2, l2, u2, a2, a2, status, stream)
        s = ctx.stats_read(stream)
    tiles = (N + 15) // 16
    if mode320:
        print("dive %d (%s): kernel %.3f ms, %d tiles; per tile: staging avg %.1f us, sweep avg %.1f us, rounds+tail avg %.1f us, whole block avg %.1f us" %
              (D, "implicit" if implicit else "explicit", ctx.last_kernel_ms(), tiles, s["failed_nodes"] / tiles / 100, s["steps3"] / tiles / 100, s["narrowings"] / tiles / 100, (s["nodes"] - N) / tiles / 100))
    else:
        print("dive %d (%s): kernel %.3f ms, %d tiles; per tile: sweep avg %.1f us max %.1f us, rounds+tail avg %.1f us max %.1f us" %
              (D, "implicit" if implicit else "explicit", ctx.last_kernel_ms(), tiles, s["steps3"] / tiles / 100, s["failed_nodes"] / 100, s["narrowings"] / tiles / 100, s["waves"] / 100))
