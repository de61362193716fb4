#!/usr/bin/env python
"""Histogram of the wake-up rounds of the deep-dive batches by number of changed (node, variable) pairs, from a
-DPCP_ABLATE=1024 profiling build.  usage: PCP_HIP_LIB=.../lib1024.so python tools/round_hist.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd import workloads as W
n = 1000; nodes = 4096
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
dev = torch.device("cuda:0")
for D in (500, 3000):
    ctx.set_option("nodes_per_block", 0)
    lb, ub, _ = W.nqueens_deep(ctx, n, D, nodes, implicit=True)
    N = lb.shape[0]
    status = torch.zeros(N, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.set_option("nodes_per_block", 16)
    for solo in (1, 0):
        ctx.set_option("solo_cascade", solo)
        l2, u2 = lb.clone(), ub.clone()
        ctx.stats_reset(stream)
        ctx.propagate_device(N, l2, u2, l2, u2, None, None, status, stream)
        s = ctx.stats_read(stream)
        tiles = (N + 15) // 16
        M40 = (1 << 40) - 1
        cls = [("1 pair", "steps3"), ("2-4", "narrowings"), ("5-16", "failed_nodes"), ("17..cap", "evaluated"), (">cap", "full_evals")]
        txt = ", ".join(f"{nm}: {(s[k] >> 40) / tiles:.1f} rounds {(s[k] & M40) / tiles / 100:.1f} us" for nm, k in cls)
        w = s["waves"]
        print(f"dive {D} solo {solo}: kernel {ctx.last_kernel_ms():.3f} ms; per tile avg: {txt}; slowest tile: {(w >> 40) / 100:.0f} us of rounds, "
              f"{((w >> 20) & 0xfffff) / 100:.0f} us in >cap, {(w & 0xfffff) / 100:.0f} us in 17..cap; slowest round anywhere: {(s['nodes'] >> 40) / 100:.0f} us, {(s['nodes'] >> 28) & 0xfff} pairs, {s['nodes'] & 0xfffffff} items")
