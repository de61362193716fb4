#!/usr/bin/env python
"""Phase times of setfix_kernel per node on the set-mode frontier of bench.py, from a -DPCP_ABLATE=2048 profiling build.
usage: PCP_HIP_LIB=.../lib2048.so python tools/set_phases.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd import workloads as W
n = 1000; sw = (n + 63) // 64
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n), set_words=sw); ctx.set_hull(1, n)
Bs, _, _ = W.nqueens_frontier_set(ctx, n, 1000)
N = Bs.shape[0]
dev = torch.device("cuda:0")
t_bits = torch.from_numpy(Bs.view(np.int64)).to(dev)
t_lb = torch.zeros((N, n), dtype=torch.int32, device=dev); t_ub = torch.zeros_like(t_lb)
st = torch.zeros(N, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    b2 = t_bits.clone()
    ctx.stats_reset(stream)
    ctx.propagate_device(N, t_lb, t_ub, t_lb, t_ub, None, None, st, stream, bits_in=b2, bits_out=b2)
    s = ctx.stats_read(stream)
f = lambda k: s[k] / N / 100
print(f"{N} nodes, kernel {ctx.last_kernel_ms():.3f} ms; per node: staging {f('steps3'):.1f} us, sweep {f('narrowings'):.1f} us, rounds {f('failed_nodes'):.1f} us, status {f('waves'):.1f} us, write-back {f('evaluated'):.1f} us")
