#!/usr/bin/env python
"""One 256-node pcp_dfs_device run on N-queens n, for `rocprofv3 --kernel-trace --stats -- python tools/dfs_trace.py`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
team = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
ctx.set_option("team", team)
lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
ctx.stats_reset()
r = ctx.dfs_device(lb0, ub0, K, capacity=2048, node_limit=K, chunk=64)
torch.cuda.synchronize()
print(r["nodes"], r["failed"], ctx.stats_read())
