#!/usr/bin/env python
"""The headline launch (16384 frontier nodes, pcp_neq.hip tiles) against tile size, block size and the LDS share a tile is sized for
(nodes_per_block x neq_block x neq_wgs).  usage: tile_sweep.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M, workloads as W
n = 1000
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
dev = torch.device("cuda", 0)
L, U, _ = W.nqueens_frontier(ctx, n, 16384, share=0, shares=8, implicit=True)
lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
st = torch.zeros(lb.shape[0], dtype=torch.uint8, device=dev)
for B in (16, 8, 4, 2):
    for blk in (256, 512):
        for wgs in (2, 3, 4):
            ctx.set_option("nodes_per_block", B); ctx.set_option("neq_block", blk); ctx.set_option("neq_wgs", wgs)
            ms = []
            try:
                for i in range(6):
                    l, u = lb.clone(), ub.clone()
                    torch.cuda.synchronize()
                    ctx.propagate_device(lb.shape[0], l, u, l, u, None, None, st)
                    if i: ms.append(ctx.last_kernel_ms())
                p = ctx.last_plan()
                print(f"B={B:2d} block={blk} wgs={wgs}: {np.median(ms)*1e3:7.1f} us  grid={p['grid']} lds={p['lds_bytes']}", flush=True)
            except Exception as e:
                print(f"B={B} block={blk} wgs={wgs}: {e}")
