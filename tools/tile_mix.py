#!/usr/bin/env python
"""What the node -> tile assignment costs: the deep-dive batches (and the bench frontier) with the rows permuted so that a tile of
16 nodes is made of 16/g groups of g consecutive rows taken from places N*g/16 rows apart (g = 16: the rows as they are).
usage: tile_mix.py [nodes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd import workloads as W
n = 1000; nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
dev = torch.device("cuda:0")
for D in (0, 500, 3000):
    ctx.set_option("nodes_per_block", 0)
    if D == 0:
        L, U, _ = W.nqueens_frontier(ctx, n, 16384, 0, 8, implicit=True)
        lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    else:
        lb, ub, _ = W.nqueens_deep(ctx, n, D, nodes, implicit=True)
    N = lb.shape[0] // 16 * 16
    lb, ub = lb[:N], ub[:N]
    status = torch.zeros(N, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.set_option("nodes_per_block", 16)
    tiles = N // 16
    for g in (16, 8, 4, 2, 1):
        # tile t, group k (0..16/g-1), member j: source row = ((k * tiles + t) * g + j)
        t = torch.arange(tiles, device=dev)[:, None, None]; k = torch.arange(16 // g, device=dev)[None, :, None]; j = torch.arange(g, device=dev)[None, None, :]
        perm = ((k * tiles + t) * g + j).reshape(-1)
        ms = []
        for _ in range(3):
            l2, u2 = lb[perm].contiguous(), ub[perm].contiguous()
            ctx.stats_reset(stream)
            ctx.propagate_device(N, l2, u2, l2, u2, None, None, status, stream)
            s = ctx.stats_read(stream)
            ms.append(ctx.last_kernel_ms())
        print(f"dive {D} N {N} groups of {g}: kernel ms {min(ms):.3f} steps/s {s['steps'] / min(ms) * 1e3:.3e} evaluated {s['evaluated']:.3e} full {s['full_evals']:.3e}")
