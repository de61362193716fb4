#!/usr/bin/env python
"""pcp_dfs_device on N-queens n: the reference's one-node-per-step DFS with no host in the loop.  usage: dfs_device.py [n] [nodes]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
ctx.dfs_device(lb0, ub0, 32, capacity=2048, node_limit=32)
torch.cuda.synchronize()
for path, team in ((2, 0), (1, 0), (2, 32), (2, 64), (2, 128), (2, 256)):
  ctx.set_option("force_path", path); ctx.set_option("team", team)
  for chunk in ((16, K, 1, 33) if (path, team) == (2, 0) else (64,)):
    ctx.dfs_device(lb0, ub0, 8, capacity=2048, node_limit=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = ctx.dfs_device(lb0, ub0, K, capacity=2048, stop_on_solution=True, node_limit=K, chunk=chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"n={n} path {path} team {team} plan {ctx.last_plan()}: {r['nodes']} nodes (chunks of {chunk} steps) in {dt*1e3:.2f} ms = {dt/r['nodes']*1e6:.1f} us/node; failed {r['failed']} open {r['open']}")
