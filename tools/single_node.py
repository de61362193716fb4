#!/usr/bin/env python
"""Latency of ONE node per call (the reference's call pattern: Space::consistency once per search node) on
N-queens n: device-resident buffers, team path.  Prints us/node and filter-steps/s for several team sizes."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
props = M.nqueens_props(n)
ctx = E.Context(0)
ctx.set_model(n, props)
dev = torch.device("cuda", 0)
lb = torch.ones((1, n), dtype=torch.int32, device=dev)
ub = torch.full((1, n), n, dtype=torch.int32, device=dev)
act = torch.from_numpy(E.full_active(1, ctx.n_units).view(np.int64)).to(dev)
lbo, ubo, acto = torch.empty_like(lb), torch.empty_like(ub), torch.empty_like(act)
st = torch.zeros(1, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for team in [0, 1, 16, 64, 128, 256, 512, 1024]:
    ctx.set_option("force_path", 2 if team != 1 else 1)
    ctx.set_option("team", team if team > 1 else 0)
    for _ in range(3):
        ctx.propagate_device(1, lb, ub, lbo, ubo, act, acto, st, stream)
    torch.cuda.synchronize()
    ctx.stats_reset(stream)
    K = 50
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.propagate_device(1, lb, ub, lbo, ubo, act, acto, st, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    s = ctx.stats_read(stream)
    print(f"team={team:5d}  {dt*1e6:9.1f} us/node  kernel {ctx.last_kernel_ms()*1e3:8.1f} us  {s['steps']/K/dt/1e9:8.2f} Gsteps/s  status={int(st.cpu()[0])}")
