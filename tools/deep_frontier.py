#!/usr/bin/env python
"""Throughput of one launch on a DEEP frontier: the 4096 open nodes on top of the stack after R rounds of the
device-resident search (vs bench.py's breadth-first frontier near the root). usage: deep_frontier.py R [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search_device import DeviceSearch

n = 1000; D = int(sys.argv[1]); R = int(sys.argv[2]) if len(sys.argv) > 2 else 14; batch = 4096
ctx = E.Context(0)
ctx.set_model(n, M.nqueens_props(n))
ds = DeviceSearch(ctx, batch=batch, capacity=24 * batch)
lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
ds.reset(lb0, ub0)
t0 = time.perf_counter()
ds.advance(max_rounds=D, batch=1)        # a depth-first dive of D nodes (the reference's order)
t1 = time.perf_counter()
ds.advance(max_rounds=R, batch=batch)    # then R batched rounds at the deep end of the stack
st = ds.stats
print(f"dive: {D} nodes in {t1-t0:.2f}s; then {R} rounds: total {st.num_nodes} nodes, failed {st.num_failed_node}, open {ds.size}")
lb, ub, act = (t.clone() for t in ds.top(batch))
N = lb.shape[0]
lbo, ubo, acto = torch.empty_like(lb), torch.empty_like(ub), torch.empty_like(act)
status = torch.zeros(N, dtype=torch.uint8, device=lb.device)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ctx.propagate_device(N, lb, ub, lbo, ubo, act, acto, status, stream)
torch.cuda.synchronize(); ctx.stats_reset(stream)
K = 10; t0 = time.perf_counter()
for _ in range(K):
    ctx.propagate_device(N, lb, ub, lbo, ubo, act, acto, status, stream)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
s = ctx.stats_read(stream)
assigned = (lb == ub).sum(dim=1).float().mean().item()
print(f"deep frontier: N={N} avg assigned vars/node={assigned:.1f} steps/launch={s['steps']/K:.3e} narrowings/launch={s['narrowings']/K:.0f} "
      f"waves/node={s['waves']/K/N:.2f} status={np.bincount(status.cpu().numpy(), minlength=3).tolist()} ms={dt*1e3:.3f} steps/s={s['steps']/K/dt:.3e}")
