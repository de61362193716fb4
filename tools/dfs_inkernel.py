#!/usr/bin/env python
"""pcp_dfs_device with its defaults (all-XNeqY model: the whole search loop in ONE workgroup, pcp_neq.hip DFS = true) on N-queens n:
us per node against the node limit, i.e. against the depth of the dive.  usage: dfs_inkernel.py [n] [limit ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
limits = [int(x) for x in sys.argv[2:]] or [256, 2048, 8192]
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
ctx.dfs_device(lb0, ub0, 32, capacity=4096, node_limit=32)
torch.cuda.synchronize()
for K in limits:
    for chunk in (64, 512):
        t0 = time.perf_counter()
        r = ctx.dfs_device(lb0, ub0, K, capacity=4096, stop_on_solution=True, node_limit=K, chunk=chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        p = ctx.last_plan()
        print(f"n={n} limit {K} chunk {chunk}: {r['nodes']} nodes in {dt*1e3:.2f} ms = {dt/max(r['nodes'],1)*1e6:.2f} us/node; failed {r['failed']} "
              f"solutions {r['solutions']} open {r['open']} path {p['path']} grid {p['grid']} block {p['block']}", flush=True)
