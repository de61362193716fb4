"""The headline launch at several batch WIDTHS of the same regime: k shares of the N-queens-1000 frontier (16 384 nodes each, the same depth)
in one launch.  Separates the launch's fixed costs (start-up, the lockstep first tile) from its steady state.  usage: python tools/wide_batch.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
import pcp_amd.engine as E
from pcp_amd import model as M, workloads as W
n = 1000
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
for a in sys.argv[1:]:
    k_, v_ = a.split("="); ctx.set_option(k_, int(v_))
dev = torch.device("cuda", 0)
parts = [W.nqueens_frontier(ctx, n, 16384, share=s, shares=8, implicit=True)[:2] for s in range(8)]
for k in (1, 2, 4, 8):
    L = torch.from_numpy(np.concatenate([p[0] for p in parts[:k]])).to(dev)
    U = torch.from_numpy(np.concatenate([p[1] for p in parts[:k]])).to(dev)
    N = L.shape[0]
    st = torch.zeros(N, dtype=torch.uint8, device=dev)
    copies = [(L.clone(), U.clone()) for _ in range(6)]
    ctx.set_option("time_kernels", 0)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.propagate_device(N, *copies[0], *copies[0], None, None, st, stream)
    torch.cuda.synchronize()
    ctx.stats_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for c in copies[1:]:
        ctx.propagate_device(N, c[0], c[1], c[0], c[1], None, None, st, stream)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    s = ctx.stats_read()
    print(f"{N:7d} nodes: {ms * 1e3:7.1f} us per launch, {N * 8000 / ms / 1e6:7.0f} GB/s = {N * 8000 / ms / 1e6 / 8000:.3f} of 8 TB/s; narrowings/launch {s['narrowings'] / 5:.0f}, grid {ctx.last_plan()['grid']}")
    del copies, L, U
