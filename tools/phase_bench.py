#!/usr/bin/env python
"""Block-level phase times of the bench launch from a -DPCP_ABLATE=320 build (64 + 256): staging, sweep, rounds, whole block
(100 MHz wall-clock ticks summed over blocks).  usage: PCP_HIP_LIB=.../lib320.so python tools/phase_bench.py [nodes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search import bfs_frontier

n = 1000
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = E.Context(0)
ctx.set_model(n, M.nqueens_props(n))
L, U, A, _ = bfs_frontier(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), N)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
act = torch.from_numpy(A.view(np.int64)).to(dev)
status = torch.zeros(N, dtype=torch.uint8, device=dev)
for _ in range(3):
    l2, u2, a2 = lb.clone(), ub.clone(), act.clone()
    ctx.stats_reset(stream)
    ctx.propagate_device(N, l2, u2, l2, u2, a2, a2, status, stream)
    s = ctx.stats_read(stream)
B = 32 if N >= 8192 else 16
tiles = (N + B - 1) // B
us = lambda v: v / tiles / 100.0
print("kernel %.1f us; per block (avg): staging %.1f us, sweep %.1f us, rounds %.1f us, whole block %.1f us" %
      (ctx.last_kernel_ms() * 1e3, us(s["failed_nodes"]), us(s["steps3"]), us(s["narrowings"]), us(s["nodes"] - N)))
