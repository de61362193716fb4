"""Times the assignment-driven kernel (pcp_neq.hip) on the bench batches under several launch shapes: HIP-event kernel time per
launch, in place on fresh copies.  usage: python tools/neq_probe.py [frontier|deep500|deep3000 ...]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
import pcp_amd.engine as E
from pcp_amd import model as M, workloads as W

n = 1000
ctx = E.Context(0)
ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["frontier", "deep500", "deep3000"]
batches = {}
if "frontier" in which:
    L, U, _ = W.nqueens_frontier(ctx, n, 16384, share=0, shares=8, implicit=True)
    batches["frontier"] = (torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev))
for d in (500, 3000):
    if f"deep{d}" in which:
        lb, ub, _ = W.nqueens_deep(ctx, n, d, 4096, implicit=True)
        batches[f"deep{d}"] = (lb, ub)

def timeit(lb, ub, opts, reps=5):
    for k, v in {"nodes_per_block": 0, "neq_block": 0, "neq_debug": 0, "neq_path": 1, "neq_wgs": 2, "neq_persist": 1, **opts}.items():
        ctx.set_option(k, v)
    N = lb.shape[0]
    st = torch.zeros(N, dtype=torch.uint8, device=dev)
    ms = []
    for i in range(reps + 1):
        l, u = lb.clone(), ub.clone()
        torch.cuda.synchronize()
        ctx.propagate_device(N, l, u, l, u, None, None, st)
        if i: ms.append(ctx.last_kernel_ms())
    pl = ctx.last_plan()
    if opts.get("neq_debug", 0) & 32:
        ctx.stats_reset()
        l, u = lb.clone(), ub.clone()
        ctx.propagate_device(N, l, u, l, u, None, None, st)
        torch.cuda.synchronize()
        s = ctx.stats_read()
        g_ = pl["grid"]
        print(f"    phase ticks per workgroup (avg): staging {s['steps3']/g_:.0f}  rounds {s['failed_nodes']/g_:.0f}  status {s['waves']/g_:.0f}  write-back+counters {s['full_evals']/g_:.0f}")
        r = ctx.debug_counters()["raw"]
        if r[15]:
            k = r[15]
            names = ["staging", "r0 list build", "r0 walk (wave 0)", "r0 end barrier", "later rounds", "status", "write-back+counters"]
            print("    finer (avg ticks per workgroup): " + "  ".join(f"{nm} {r[8 + i] / k:.0f}" for i, nm in enumerate(names)))
            first = (~r[5]) & 0xFFFFFFFFFFFFFFFF
            print(f"    timeline (100 MHz): first workgroup start 0, last start {(r[7] - first) / 100:.2f} us, last end {(r[6] - first) / 100:.2f} us; workgroups {k}")
    if opts.get("neq_debug", 0) & 8:
        ctx.stats_reset()
        l, u = lb.clone(), ub.clone()
        ctx.propagate_device(N, l, u, l, u, None, None, st)
        torch.cuda.synchronize()
        s = ctx.stats_read()
        nw = pl["grid"] * pl["block"] // 64
        print(f"    timers (round 0, per wavefront avg): walk {s['steps3']/nw:.0f} ticks, node loops {s['failed_nodes']/nw:.0f} ticks, pieces {(s['waves']-N)/nw:.1f}; evaluated {s['evaluated']:.3e}")
    return float(np.median(ms)), pl

configs = json.loads(os.environ["NEQ_CONFIGS"]) if "NEQ_CONFIGS" in os.environ else [{}, {"neq_persist": 0}, {"neq_debug": 1}, {"neq_debug": 2}, {"neq_debug": 3}, {"neq_debug": 4}, {"neq_wgs": 1}]
if "frontier" in batches:  # one generation of tiles only (512 tiles = two per CU): what the first wave of workgroups costs by itself
    l8, u8 = batches["frontier"]
    batches["frontier-half"] = (l8[:8192].contiguous(), u8[:8192].contiguous())
for name, (lb, ub) in batches.items():
    for c in configs:
        ms, pl = timeit(lb, ub, c)
        print(f"{name:9s} {json.dumps(c):48s} {ms*1e3:9.1f} us   B={pl['nodes_per_block']} block={pl['block']} grid={pl['grid']} lds={pl['lds_bytes']} path={pl['path']}", flush=True)
