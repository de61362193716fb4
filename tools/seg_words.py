#!/usr/bin/env python
"""Word-group sweep: where the time goes (-DPCP_ABLATE=128 build).  usage: PCP_HIP_LIB=.../lib128.so python tools/seg_words.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd import workloads as W
n, N = 1000, 4096
implicit = os.environ.get("PCP_ACTIVE", "implicit") == "implicit"
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
L, U, A = W.nqueens_frontier(ctx, n, N, 0, 8, implicit=implicit)
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
act = None if A is None else torch.from_numpy(A.view(np.int64)).to(dev)
status = torch.zeros(N, dtype=torch.uint8, device=dev)
ctx.set_option("nodes_per_block", 16)
for _ in range(2):
    l2, u2, a2 = lb.clone(), ub.clone(), None if act is None else act.clone()
    ctx.stats_reset(stream)
    ctx.propagate_device(N, l2, u2, l2, u2, a2, a2, status, stream)
    s = ctx.stats_read(stream)
waves = N // 16 * 16
f = s["failed_nodes"]
print("kernel %.3f ms; per wavefront: loads + level -1 %.0f ticks, record level %.0f ticks; words reaching level 0: %d, level 1: %d, level 2: %d (per launch)"
      % (ctx.last_kernel_ms(), s["steps3"] / waves, s["narrowings"] / waves, f & 0xFFFFFF, (f >> 24) & 0xFFFFF, f >> 44))
print("  phase B per wavefront: waiting for batch loads %.0f ticks, processing %.0f ticks" % (s["waves"] / waves, s["nodes"] / waves))
