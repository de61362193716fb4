#!/usr/bin/env python
"""Device-resident search on N-queens n (stack, propagation and branching on the GPU): nodes/s, filter-steps/s.
usage: device_search.py n batch node_limit [all]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search_device import DeviceSearch

n = int(sys.argv[1]); batch = int(sys.argv[2]); limit = int(sys.argv[3]); allsol = len(sys.argv) > 4
ctx = E.Context(0)
ctx.set_model(n, M.nqueens_props(n))
ds = DeviceSearch(ctx, batch=batch, capacity=max(24 * batch, limit + 2 * batch))  # near the root every node is Unknown: the stack grows by a batch per round
lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
ds.run(lb0, ub0, all_solutions=allsol, node_limit=min(limit, 4 * batch))  # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
st = ds.run(lb0, ub0, all_solutions=allsol, node_limit=limit, keep_solutions=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"n={n} batch={batch} nodes={st.num_nodes} rounds={st.rounds} solutions={st.num_solution} failed={st.num_failed_node} max_open={st.max_open} "
      f"time={dt:.3f}s nodes/s={st.num_nodes/dt:.0f} filter-steps={st.filter_steps:.3e} steps/s={st.filter_steps/dt:.3e}")
if st.solutions:
    s = st.solutions[0]
    ok = len(set(s)) == n and len({int(s[i]) + i for i in range(n)}) == n and len({int(s[i]) - i for i in range(n)}) == n
    print("first solution valid:", ok)
