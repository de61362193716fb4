#!/usr/bin/env python
"""pcp_dfs_forest_device on N-queens n over Interval<i32> domains: expansion to `trees` open nodes, then one in-kernel DFS per open node.
usage: neq_forest_probe.py [n] [budget] [trees ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search_forest import forest_search
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
tree_counts = [int(x) for x in sys.argv[3:]] or [256, 512, 768, 1536]
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
for trees in tree_counts:
  for blk in (0, 512):
    ctx.set_option("neq_dfs_block", blk)
    for steps in (1024,):
        forest_search(ctx, lb0, ub0, node_limit=4 * trees, n_trees=trees, steps_per_launch=4, capacity=64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = forest_search(ctx, lb0, ub0, node_limit=budget, n_trees=trees, steps_per_launch=steps, capacity=4096)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"n={n} block {blk} trees {r['trees']} steps/launch {steps}: {r['nodes']} nodes in {dt*1e3:.1f} ms = {r['nodes']/dt:.3e} nodes/s; launches {r['launches']} "
              f"failed {r['failed']} solutions {r['solutions']} error {r['error']} last kernel {ctx.last_kernel_ms():.2f} ms grid {ctx.last_plan()['grid']}", flush=True)
