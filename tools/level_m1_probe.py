#!/usr/bin/env python
"""CPU emulation of the sweep's level -1 test on the bench frontier: which words fail, and why."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search import bfs_frontier

n, N = 1000, 4096
props = M.nqueens_props(n)
ctx = E.Context(0); ctx.set_model(n, props)
L, U, A, _ = bfs_frontier(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), N)
x = props["var"][:, 0].astype(np.int64); y = props["var"][:, 1].astype(np.int64)
d = (props["off"][:, 1] - props["off"][:, 0]).astype(np.int64)
P = len(props); W = (P + 63) // 64
pad = W * 64 - P
xp = np.concatenate([x, np.full(pad, x[-1])]).reshape(W, 64); yp = np.concatenate([y, np.full(pad, y[-1])]).reshape(W, 64)
dp = np.concatenate([d, np.full(pad, d[-1])]).reshape(W, 64)
xlo, xhi, ylo, yhi = xp.min(1), xp.max(1), yp.min(1), yp.max(1)
dmin, dmax = dp.min(1), dp.max(1)
print("words", W, "x range max", (xhi - xlo).max(), "y range max", (yhi - ylo).max(), "first records", x[:6], y[:6], d[:6])
for t in (0, 1, 100, 255):
    l, u = L[16 * t:16 * t + 16], U[16 * t:16 * t + 16]
    mn_n, mn_u = (-l).min(0), u.min(0)
    def rmin(a, lo, hi):
        return np.array([a[lo[i]:hi[i] + 1].min() for i in range(W)])
    Xn, Xu, Yn, Yu = rmin(mn_n, xlo, xhi), rmin(mn_u, xlo, xhi), rmin(mn_n, ylo, yhi), rmin(mn_u, ylo, yhi)
    c1 = Xn + Yu + dmin - 1; c2 = Xu + Yn - dmax - 1
    fail = (c1 < 0) | (c2 < 0)
    print("tile", t, "fail words", int(fail.sum()), "of", W, "| c1<0:", int((c1 < 0).sum()), "c2<0:", int((c2 < 0).sum()),
          "| narrow vars (max width<900):", np.nonzero((u - l).max(0) < 900)[0][:8], "differing lb vars:", int((l.max(0) != l.min(0)).sum()))
    bad = np.nonzero(fail)[0][:5]
    for w in bad:
        print("   word", w, "x", xlo[w], xhi[w], "y", ylo[w], yhi[w], "d", dmin[w], dmax[w], "Xn,Xu,Yn,Yu", Xn[w], Xu[w], Yn[w], Yu[w])
