#!/usr/bin/env python
"""Phase ablation of the two-pass launch's first pass (neqwave_kernel) on the bench frontier: neq_debug 256 = no rounds, 512 = no status
scan (results are wrong with either).  usage: wave_probe.py"""
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
for opts in ({"neq_wave": 0}, {}, {"neq_debug": 768}, {"neq_wave_block": 64, "neq_wave_per_cu": 16}, {"neq_wave_block": 128, "neq_wave_per_cu": 8}, {"neq_wave_block": 256, "neq_wave_per_cu": 2}, {"neq_wave_block": 256, "neq_wave_per_cu": 8}, {"neq_wave_block": 64, "neq_wave_per_cu": 16, "neq_debug": 768}):
    for k, v in {"neq_wave": 1, "neq_debug": 0, "neq_wave_max": 4, "neq_wave_block": 256, "neq_wave_per_cu": 4, **opts}.items():
        ctx.set_option(k, v)
    ms = []
    for i in range(6):
        l, u = lb.clone(), ub.clone()
        torch.cuda.synchronize()
        ctx.propagate_device(lb.shape[0], l, u, l, u, None, None, st)
        if i: ms.append(ctx.last_kernel_ms())
    print(f"{json.dumps(opts):72s} {np.median(ms)*1e3:8.1f} us  two-pass={ctx.last_plan()['compact']}", flush=True)

ctx.set_option("neq_wave", 1); ctx.set_option("neq_wave_max", 4); ctx.set_option("neq_debug", 1024)
ctx.stats_reset()
l, u = lb.clone(), ub.clone()
ctx.propagate_device(lb.shape[0], l, u, l, u, None, None, st)
torch.cuda.synchronize()
s = ctx.stats_read()
nw = 4096
print(f"ticks per wavefront (4 nodes): prologue {s['steps3']/nw:.0f}  row loads {s['narrowings']/nw:.0f}  cells {s['full_evals']/nw:.0f}  decision {s['waves']/nw:.0f}  rounds {s['evaluated']/nw:.0f}  status {s['steps']/nw:.0f}  tail {s['failed_nodes']/nw:.0f}  lifetime {s['nodes']/nw:.0f}  (kernel {ctx.last_kernel_ms()*1e3:.1f} us)")
ctx.set_option("neq_debug", 0); ctx.set_option("neq_wave", 0)
