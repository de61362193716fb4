"""Config 3 on bigfix_kernel: time, filter steps and wake-up rounds per node under the three round policies (big_round 0 auto / 1 dense / 2 sparse)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
import pcp_amd.engine as E
from pcp_amd import workloads as W
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = E.Context(0)
p3, lb3, ub3, sol3 = W.planted_binary_csp(0xC3, 50_000, 500_000)
L3, U3 = W.unit_narrowing_prefix(0xC3 + 1, lb3, ub3, sol3, N)
ctx.set_model(50_000, p3); ctx.set_hull(0, 999)
dev = torch.device("cuda", 0)
lb, ub = torch.from_numpy(L3).to(dev), torch.from_numpy(U3).to(dev)
st = torch.zeros(N, dtype=torch.uint8, device=dev)
for mode, dk in ((0, 1), (0, 2), (0, 3), (0, 4), (0, 6), (0, 8), (0, 12), (0, 16)):
    ctx.set_option("big_round", mode); ctx.set_option("big_dense_k", dk)
    ms = []
    for i in range(3):
        l, u = lb.clone(), ub.clone()
        ctx.stats_reset()
        ctx.propagate_device(N, l, u, l, u, None, None, st)
        ms.append(ctx.last_kernel_ms())
    s = ctx.stats_read(); d = ctx.debug_counters()
    print(json.dumps({"big_round": mode, "dense_k": dk, "ms": round(float(np.median(ms)), 2), "steps_per_node": s["evaluated"] / N, "narrowings_per_node": s["narrowings"] / N,
                      "dense_rounds_per_node": d["big_dense"] / N, "sparse_rounds_per_node": d["big_sparse"] / N, "path": ctx.last_plan()["path"]}))
