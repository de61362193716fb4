#!/usr/bin/env python
"""pcp_dfs_forest_device_set on N-queens n over IntervalSet<i32> domains (FDSpace): a frontier of open nodes from the batched device
search, then one tree per open node, each in one CU's LDS with an undo trail.  Nodes per second against the number of trees and the
nodes per launch.   usage: set_forest_probe.py [n] [budget] [trees ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search_device import DeviceSearch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
tree_counts = [int(x) for x in sys.argv[3:]] or [256, 512, 1024]
sw = (n + 63) // 64
ctx = E.Context(0)
ctx.set_model(n, M.nqueens_props(n), set_words=sw)
ctx.set_hull(1, n)
lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)

for want in tree_counts:
    ds = DeviceSearch(ctx, batch=want, capacity=8 * want + 64, implicit=True)
    ds.reset(lb0, ub0, 1)
    while 0 < ds.size < want:
        if ds.advance(all_solutions=True, max_rounds=1, keep_solutions=0):
            break
    ds.compact()
    k = min(ds.size, want)
    roots = ds.bits[ds.size - k:ds.size].clone()
    seeded = ds.stats.num_nodes
    del ds
    torch.cuda.empty_cache()
    for steps in (256, 2048):
        info = {}
        ctx.stats_reset()
        ctx.dfs_forest_set(roots, node_limit=2 * k, steps_per_launch=2, want_solution=False)  # warm-up
        torch.cuda.synchronize()
        ctx.stats_reset()
        t0 = time.perf_counter()
        r = ctx.dfs_forest_set(roots, node_limit=budget, steps_per_launch=steps, trail_capacity=1 << 21, level_capacity=1 << 14, want_solution=False, info=info)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s = ctx.stats_read()
        print(f"n={n} trees {k} (frontier after {seeded} nodes) steps/launch {steps}: {r['nodes']} nodes in {dt*1e3:.1f} ms = {r['nodes']/dt:.3e} nodes/s "
              f"({dt/max(r['nodes'],1)*1e6*k:.2f} us per node and tree); launches {r['launches']} failed {r['failed']} solutions {r['solutions']} error {r['error']} "
              f"trail max {info.get('trail_max')} levels max {info.get('levels_max')} evaluated/node {s['evaluated']/max(r['nodes'],1):.0f} last kernel {ctx.last_kernel_ms():.2f} ms", flush=True)
