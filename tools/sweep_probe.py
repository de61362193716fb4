#!/usr/bin/env python
"""How much of the bench kernel is the sweep's hot part?  Times one launch on (a) the bench frontier, (b) the same nodes
again after their fixpoint (nothing left to narrow or entail: hot part only), (c) 4096 copies of the root node.
usage: python tools/sweep_probe.py [n] [nodes]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search import bfs_frontier

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = E.Context(0)
ctx.set_model(n, M.nqueens_props(n))
L, U, A, _ = bfs_frontier(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), N)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream


def run(tag, lb, ub, act, reps=5, in_place=False):
    if in_place:  # every repetition on its own fresh copy
        copies = [(lb.clone(), ub.clone(), act.clone()) for _ in range(reps)]
    lbo, ubo, acto = torch.empty_like(lb), torch.empty_like(ub), torch.empty_like(act)
    status = torch.zeros(lb.shape[0], dtype=torch.uint8, device=dev)
    ms = []
    for i in range(reps):
        ctx.stats_reset(stream)
        if in_place:
            lbo, ubo, acto = copies[i]
            ctx.propagate_device(lb.shape[0], lbo, ubo, lbo, ubo, acto, acto, status, stream)
        else:
            ctx.propagate_device(lb.shape[0], lb, ub, lbo, ubo, act, acto, status, stream)
        s = ctx.stats_read(stream)
        ms.append(ctx.last_kernel_ms())
    k = min(ms)
    print("%-28s kernel %.3f ms  steps %.3e  %.2f Tsteps/s  narrowings %d  waves/node %.3f" %
          (tag, k, s["steps"], s["steps"] / k / 1e9, s["narrowings"], s["waves"] / max(1, s["nodes"])))
    return lbo, ubo, acto


lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
act = torch.from_numpy(A.view(np.int64)).to(dev)
run("frontier, in place", lb, ub, act, in_place=True)
o = run("frontier", lb, ub, act)
o2 = run("frontier, at fixpoint", *o)
root_l = torch.ones((N, n), dtype=torch.int32, device=dev)
root_u = torch.full((N, n), n, dtype=torch.int32, device=dev)
root_a = torch.from_numpy(E.full_active(N, ctx.n_units).view(np.int64)).to(dev)
o3 = run("root copies", root_l, root_u, root_a)
run("root copies, at fixpoint", *o3)
