"""Replays ONE bench leg's timed launches and nothing else, so that a rocprofv3 PMC pass sees only those dispatches (the deep
batches are built by thousands of one-node launches, which made the PMC passes of round 2 abort):
   python tools/replay_leg.py save deep500 deep3000     # build the batches once, un-profiled, into /tmp on the GPU box
   python tools/replay_leg.py run deep500 [launches]    # load and launch: in place on fresh copies, HIP-event time per launch
   python tools/replay_leg.py run c3 | c4 | f4 | mix | explicit | frontier | cells | search | setforest   # these build their input with a handful of launches"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
import pcp_amd.engine as E
from pcp_amd import model as M, workloads as W

mode, names = sys.argv[1], sys.argv[2:]
n = 1000
dev = torch.device("cuda", 0)
ctx = E.Context(0)


def nq():
    ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)


HINT = None
CELLS = False


def launches(lb, ub, act, k):
    N = lb.shape[0]
    if CELLS:  # the nodes resident as packed cells (pcp_device_batch.cell_format PCP_CELLS_PACKED16)
        st = torch.zeros(N, dtype=torch.uint8, device=dev)
        cells0 = ctx.pack_rows(lb, ub)
        ms = []
        for i in range(k + 1):
            c = cells0.clone()
            torch.cuda.synchronize()
            ctx.propagate_device(N, c, None, c, None, None, None, st, dirty=HINT, cells=True)
            if i:
                ms.append(ctx.last_kernel_ms())
        return ms, ctx.last_plan()
    st = torch.zeros(N, dtype=torch.uint8, device=dev)
    ms = []
    for i in range(k + 1):
        l, u = lb.clone(), ub.clone()
        a = None if act is None else act.clone()
        torch.cuda.synchronize()
        ctx.propagate_device(N, l, u, l, u, a, a, st, dirty=HINT)
        if i:
            ms.append(ctx.last_kernel_ms())
    return ms, ctx.last_plan()


if mode == "save":
    nq()
    for nm in names:
        d = int(nm.replace("deep", ""))
        lb, ub, _ = W.nqueens_deep(ctx, n, d, 4096, implicit=True)
        torch.save((lb.cpu(), ub.cpu()), f"/tmp/{nm}.pt")
        print("saved", nm, tuple(lb.shape))
else:
    nm = names[0]
    k = int(names[1]) if len(names) > 1 else 5
    act = None
    if nm.startswith("deep"):
        nq()
        lb, ub = (t.to(dev) for t in torch.load(f"/tmp/{nm}.pt"))
    elif nm in ("frontier", "cells"):
        CELLS = nm == "cells"
        nq()
        L, U, _ = W.nqueens_frontier(ctx, n, 16384, share=0, shares=8, implicit=True)
        lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    elif nm == "c3":
        p3, lb3, ub3, sol3 = W.planted_binary_csp(0xC3, 50_000, 500_000)
        L3, U3 = W.unit_narrowing_prefix(0xC3 + 1, lb3, ub3, sol3, 4096)
        ctx.set_model(50_000, p3); ctx.set_hull(0, 999)
        lb, ub = torch.from_numpy(L3).to(dev), torch.from_numpy(U3).to(dev)
    elif nm == "c4":
        p4, L4, U4, A4 = W.golomb_frontier(ctx, 4096)
        lb, ub, act = torch.from_numpy(L4).to(dev), torch.from_numpy(U4).to(dev), torch.from_numpy(A4.view(np.int64)).to(dev)
    elif nm == "f4":
        vs4, cs4, Lf, Uf = W.cumulative_nodes(4096)
        M.push_model(ctx, cs4, len(vs4))
        lb, ub = torch.from_numpy(Lf).to(dev), torch.from_numpy(Uf).to(dev)
    elif nm == "mix":
        nq()
        lb, ub, _ = W.nqueens_dfs_samples(ctx, n, 16384, 12)
        nq()
    elif nm == "mixh":
        # the children of the mix nodes with their dirty-variable hints (bench.py's MIXH leg): propagate the mix batch, branch it with hints
        nq()
        pl, pu, _ = W.nqueens_dfs_samples(ctx, n, 16384, 12)
        nq()
        stp = torch.zeros(pl.shape[0], dtype=torch.uint8, device=dev)
        ctx.propagate_device(pl.shape[0], pl, pu, pl, pu, None, None, stp)
        cl = torch.empty((2 * pl.shape[0], n), dtype=torch.int32, device=dev); cu = torch.empty_like(cl)
        cd = torch.full((2 * pl.shape[0],), -1, dtype=torch.int32, device=dev)
        cnt = torch.zeros(5, dtype=torch.int32, device=dev)
        ctx.branch_device(pl.shape[0], pl, pu, None, stp, cl, cu, None, cnt, child_dirty=cd)
        kc = min(int(cnt[0].item()), 16384)
        lb, ub, HINT = cl[:kc].clone(), cu[:kc].clone(), cd[:kc].clone()
    elif nm == "explicit":
        nq()
        L, U, A = W.nqueens_frontier(ctx, n, 4096, share=0, shares=8, implicit=False)
        lb, ub, act = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev), torch.from_numpy(A.view(np.int64)).to(dev)
    elif nm == "search":
        from pcp_amd.search_device import DeviceSearch
        nq()
        ds = DeviceSearch(ctx, batch=4096, capacity=40 * 4096, implicit=True)
        ds.reset(np.ones(n, np.int32), np.full(n, n, np.int32))
        ds.advance(max_rounds=40)   # ~30 full rounds of 4096 nodes: propagate + branch per round
        print(json.dumps({"leg": nm, "nodes": ds.stats.num_nodes, "rounds": ds.stats.rounds}))
        sys.exit(0)
    elif nm == "neqforest":
        # the interval-mode forest (pcp_dfs_forest_device): 2048 trees, 256 nodes per tree and launch, 500 000 nodes
        import time
        from pcp_amd.search_forest import forest_search
        nq()
        t0 = time.perf_counter()
        big = os.environ.get("PCP_FOREST_8K")  # the forest at its operating point: 2 M nodes, 8192 trees of 128 threads
        r = forest_search(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), node_limit=2_000_000 if big else 500_000, n_trees=8192 if big else 2048, steps_per_launch=256)
        torch.cuda.synchronize()
        print(json.dumps({"leg": nm, "nodes": r["nodes"], "trees": r["trees"], "launches": r["launches"], "seconds_incl_expansion_and_allocation": time.perf_counter() - t0,
                          "last_kernel_ms": ctx.last_kernel_ms(), "plan": ctx.last_plan()}))
        sys.exit(0)
    elif nm == "setforest":
        # the set-mode forest (pcp_dfs_forest_device_set): 512 trees, launches of 256 nodes per tree, 200 000 nodes
        import time
        from pcp_amd.search_forest import forest_search_set
        sw = (n + 63) // 64
        ctx.set_model(n, M.nqueens_props(n), set_words=sw); ctx.set_hull(1, n)
        t0 = time.perf_counter()
        r = forest_search_set(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), 1, node_limit=200_000, n_trees=512, steps_per_launch=256, trail_capacity=1 << 19)
        torch.cuda.synchronize()
        print(json.dumps({"leg": nm, "nodes": r["nodes"], "trees": r["trees"], "launches": r["launches"], "seconds_incl_expansion_and_allocation": time.perf_counter() - t0,
                          "last_kernel_ms": ctx.last_kernel_ms(), "plan": ctx.last_plan()}))
        sys.exit(0)
    else:
        raise SystemExit(nm)
    for kv in filter(None, os.environ.get("PCP_OPTS", "").split(",")):  # e.g. PCP_OPTS=neq_debug=3,neq_stagger=8000 — for the timed launches only
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    ms, pl = launches(lb, ub, act, k)
    print(json.dumps({"leg": nm, "kernel_ms": ms, "median_ms": float(np.median(ms)), "plan": pl}))
