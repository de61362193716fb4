#!/usr/bin/env python
"""Where a wavefront of the sweep spends its time: -DPCP_ABLATE=128 build, s_memtime ticks per segment of process()
summed over all wavefronts (reported through the steps3 / narrowings / failed_nodes / waves counters).
usage: PCP_HIP_LIB=/path/lib128.so python tools/seg_times.py [n] [nodes]"""
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
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
DIVE = int(os.environ.get("DIVE", "0"))
if os.environ.get("PCP_WORD_LEVEL"): ctx.set_option("word_level", int(os.environ["PCP_WORD_LEVEL"]))
if DIVE:  # a deep frontier: DIVE nodes depth-first, then 14 batched rounds (tools/deep_frontier.py)
    from pcp_amd.search_device import DeviceSearch
    ds = DeviceSearch(ctx, batch=N, capacity=24 * N)
    ds.reset(np.ones(n, np.int32), np.full(n, n, np.int32))
    ds.advance(max_rounds=DIVE, batch=1)
    ds.advance(max_rounds=14, batch=N)
    lb, ub, act = (t.clone() for t in ds.top(N))
else:
    L, U, A, _ = bfs_frontier(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), N)
    lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    act = torch.from_numpy(A.view(np.int64)).to(dev)
lbo, ubo, acto = torch.empty_like(lb), torch.empty_like(ub), torch.empty_like(act)
status = torch.zeros(N, dtype=torch.uint8, device=dev)
for _ in range(2):  # in place on a fresh copy, like bench.py
    l2, u2, a2 = lb.clone(), ub.clone(), act.clone()
    ctx.stats_reset(stream)
    ctx.propagate_device(N, l2, u2, l2, u2, a2, a2, status, stream)
    s = ctx.stats_read(stream)
ms = ctx.last_kernel_ms()
words = (ctx.n_units + 63) // 64
B = 32 if N >= 8192 else 16
tiles = (N + B - 1) // B
waves = tiles * 16
chunks = words / 4 / 16  # per wavefront
segs = [s["steps3"], s["narrowings"], s["failed_nodes"], s["waves"]]
tot = sum(segs)
print("kernel %.3f ms = %.0f cycles at 2.4 GHz; %d wavefronts, %.0f chunks each" % (ms, ms * 2.4e6, waves, chunks))
for name, v in zip(("wait live+alive4", "hot (4 words)", "cold", "store"), segs):
    print("  %-18s %6.1f %% of timed ticks, %8.0f ticks per chunk per wavefront" % (name, 100.0 * v / tot, v / waves / chunks))
print("  of the hot part, LDS reads + level-1 arithmetic: %.0f ticks per chunk per wavefront" % ((s["nodes"] - N) / waves / chunks))
print("  timed ticks per wavefront: %.0f" % (tot / waves))
