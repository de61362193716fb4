#!/usr/bin/env python
"""Kernel time of the deep-dive batches (4096 open nodes after a 500 / 3000-node DFS dive on N-queens-1000) against the tile
size: nodes_per_block 16 = 256 tiles, one per CU; 8 = 512 tiles.  usage: deep_tiles.py [nodes]"""
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
for D in (500, 3000):
    ctx.set_option("nodes_per_block", 0)
    lb, ub, _ = W.nqueens_deep(ctx, n, D, nodes, implicit=True)
    N = lb.shape[0]
    status = torch.zeros(N, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for npb, blk in ((16, 0), (32, 0), (32, 1024), (8, 512)):
        ctx.set_option("nodes_per_block", npb)
        ctx.set_option("neq_block", blk)
        ms = []
        for _ in range(4):
            l2, u2 = lb.clone(), ub.clone()
            ctx.stats_reset(stream)
            ctx.propagate_device(N, l2, u2, l2, u2, None, None, status, stream)
            s = ctx.stats_read(stream)
            ms.append(ctx.last_kernel_ms())
        pl = ctx.last_plan()
        print(f"dive {D} nodes_per_block {npb} neq_block {blk} block {pl['block']} path {pl['path']}: kernel ms {['%.3f' % m for m in ms]} grid {pl['grid']} lds {pl['lds_bytes']} cap {pl['list_cap']} wl {pl['word_level']} steps/s {s['steps'] / min(ms) * 1e3:.3e}")
