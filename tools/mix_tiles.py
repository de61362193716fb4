"""The mix leg's batch (16 384 open nodes sampled along the reference's DFS of N-queens-1000: unrelated deep nodes) against the tile shape: nodes per
tile, threads per workgroup, workgroups per CU.  A tile of unrelated nodes shares no list walk, and a launch of 1024 tiles of very unequal cost on 512
persistent workgroups ends with its slowest pair.   usage: python tools/mix_tiles.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
import pcp_amd.engine as E
from pcp_amd import model as M, workloads as W
n = 1000
ctx = E.Context(0); ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
dev = torch.device("cuda", 0)
lb, ub, depth = W.nqueens_dfs_samples(ctx, n, 16384, 12)
N = lb.shape[0]
st = torch.zeros(N, dtype=torch.uint8, device=dev)
ref = None
configs = json.loads(os.environ["MIX_CONFIGS"]) if "MIX_CONFIGS" in os.environ else [
    {}, {"nodes_per_block": 8}, {"nodes_per_block": 4}, {"nodes_per_block": 2}, {"nodes_per_block": 1},
    {"nodes_per_block": 4, "neq_block": 256, "neq_wgs": 4}, {"nodes_per_block": 2, "neq_block": 256, "neq_wgs": 4}, {"nodes_per_block": 1, "neq_block": 256, "neq_wgs": 4},
    {"nodes_per_block": 1, "neq_block": 128, "neq_wgs": 8}, {"neq_persist": 0}, {"nodes_per_block": 4, "neq_persist": 0}]
for cfg in configs:
    for k, v in {"nodes_per_block": 0, "neq_block": 0, "neq_wgs": 2, "neq_persist": 1, **cfg}.items():
        ctx.set_option(k, v)
    ms = []
    for i in range(3):
        l, u = lb.clone(), ub.clone()
        ctx.stats_reset()
        ctx.propagate_device(N, l, u, l, u, None, None, st)
        ms.append(ctx.last_kernel_ms())
    torch.cuda.synchronize()
    s = ctx.stats_read(); pl = ctx.last_plan()
    if ref is None:
        ref = (l.clone(), u.clone(), st.clone())
    ok = st != 0
    same = bool(torch.equal(st, ref[2]) and torch.equal(l[ok], ref[0][ok]) and torch.equal(u[ok], ref[1][ok]))
    print(json.dumps({"cfg": cfg, "ms": round(float(np.median(ms)), 2), "min": round(min(ms), 2), "evaluated": f"{s['evaluated']:.3e}", "B": pl["nodes_per_block"], "block": pl["block"], "grid": pl["grid"],
                      "lds": pl["lds_bytes"], "same": same}), flush=True)
