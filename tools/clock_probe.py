"""Does the headline launch run faster on a busy GPU?  Times the same launch (HIP events) right after idle, and at the end of a burst of
back-to-back launches; reads the shader clock the driver reports while the burst runs.  usage: python tools/clock_probe.py"""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
import pcp_amd.engine as E
from pcp_amd import model as M, workloads as W
n, N = 1000, 16384
ctx = E.Context(0)
ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
dev = torch.device("cuda", 0)
L, U, _ = W.nqueens_frontier(ctx, n, N, share=0, shares=8, implicit=True)
lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
st = torch.zeros(N, dtype=torch.uint8, device=dev)
pool = [(lb.clone(), ub.clone()) for _ in range(64)]
def one(i):
    l, u = pool[i % 64]
    ctx.propagate_device(N, l, u, l, u, None, None, st)
torch.cuda.synchronize(); time.sleep(0.5)
iso = []
for i in range(5):
    torch.cuda.synchronize(); time.sleep(0.05)
    one(i); iso.append(ctx.last_kernel_ms() * 1e3)
print("isolated (after 50 ms idle):", [round(x, 1) for x in iso])
for burst in (50, 500, 3000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(burst):
        one(i)
    last = ctx.last_kernel_ms() * 1e3
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"burst of {burst}: {dt / burst * 1e6:.1f} us per launch wall, last launch {last:.1f} us by HIP events")
try:
    print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e:
    print("rocm-smi:", e)
