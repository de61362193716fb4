#!/usr/bin/env python
"""Per-tile time of the sweep vs the wake-up rounds, from a -DPCP_ABLATE=64 profiling build (which reports the phase
timers in the steps3 / narrowings counters).  usage: PCP_HIP_LIB=/tmp/lib64.so python tools/phase_times.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search_device import DeviceSearch

n = 1000; batch = 4096
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n))
if os.environ.get('PCP_WORD_LEVEL'): ctx.set_option('word_level', int(os.environ['PCP_WORD_LEVEL'])); print('word_level', os.environ['PCP_WORD_LEVEL'])
ds = DeviceSearch(ctx, batch=batch, capacity=24 * batch)
for D in (0, 500, 3000):
    ds.reset(np.ones(n, np.int32), np.full(n, n, np.int32))
    if D:
        ds.advance(max_rounds=D, batch=1)
    ds.advance(max_rounds=14, batch=batch)
    lb, ub, act = (t.clone() for t in ds.top(batch))
    N = lb.shape[0]
    lbo, ubo, acto = torch.empty_like(lb), torch.empty_like(ub), torch.empty_like(act)
    status = torch.zeros(N, dtype=torch.uint8, device=lb.device)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.propagate_device(N, lb, ub, lbo, ubo, act, acto, status, stream)
    ctx.stats_reset(stream)
    ctx.propagate_device(N, lb, ub, lbo, ubo, act, acto, status, stream)
    s = ctx.stats_read(stream)
    tiles = N / 16
    print("dive %d: kernel %.3f ms; per tile: sweep avg %.1f us max %.1f us, rounds+tail avg %.1f us max %.1f us" %
          (D, ctx.last_kernel_ms(), s["steps3"] / tiles / 100, s["failed_nodes"] / 100, s["narrowings"] / tiles / 100, s["waves"] / 100))
