"""Per-wavefront event stamps of the headline launch (option neq_trace_ptr, pcp_neq.hip PCP_TR): where does a frontier tile's time go,
and which wavefront does every barrier wait for?  usage: python tools/neq_trace.py [nodes]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
g.build()
import pcp_amd.engine as E
from pcp_amd import model as M, workloads as W

n = 1000
N = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 16384
OPTS = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
ctx = E.Context(0)
ctx.set_model(n, M.nqueens_props(n)); ctx.set_hull(1, n)
for k_, v_ in OPTS.items():
    ctx.set_option(k_, v_)
print("options", OPTS)
dev = torch.device("cuda", 0)
L, U, _ = W.nqueens_frontier(ctx, n, N, share=0, shares=8, implicit=True)
lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
st = torch.zeros(N, dtype=torch.uint8, device=dev)
names = ["entry", "zeroed", "staged (arrive)", "staged (released)", "list built (arrive)", "list built (released)", "first payload loads issued",
         "walk done (arrive)", "round end (released)", "rounds done", "status done (arrive)", "status (released)", "counters (arrive)", "counters (released)", "end"]
for rep in range(3):
    l, u = lb.clone(), ub.clone()
    tiles = (N + 15) // 16
    tr = torch.zeros((tiles, 16, 16), dtype=torch.int64, device=dev)
    ctx.set_option("neq_trace_ptr", tr.data_ptr())
    torch.cuda.synchronize()
    ctx.propagate_device(N, l, u, l, u, None, None, st)
    torch.cuda.synchronize()
    ms = ctx.last_kernel_ms()
    ctx.set_option("neq_trace_ptr", 0)
    pl = ctx.last_plan()
    nwv = pl["block"] // 64
    t = tr.cpu().numpy()[:, :nwv, :]
    rel = t[:, :, :15] - t[:, :1, :1]           # ticks since wavefront 0's entry
    if rep < 2:
        continue
    print(f"kernel {ms * 1e3:.1f} us, grid {pl['grid']} x {pl['block']} threads")
    rt = t[:, 0, 15]
    print(f"realtime (100 MHz) end stamps: first {(rt.min() - rt.min()) / 100:.1f} us ... last {(rt.max() - rt.min()) / 100:.1f} us")
    # the launch's own clock: ticks from the first entry to the last end against the HIP-event time (s_memtime is one counter for the chip)
    span = int(t[:, :, 14].max() - t[:, :, 0][t[:, :, 0] > 0].min())
    print(f"ticks first entry -> last end: {span}  = {span / (ms * 1e3):.0f} ticks/us of kernel time")
    ent = (t[:, 0, 0] - t[:, :, 0][t[:, :, 0] > 0].min())
    g_ = pl["grid"]
    for lo, hi, what in ((0, g_, "first tiles"), (g_, tiles, "second tiles")):
        e0, e1 = ent[lo:hi], (t[lo:hi, :, 14].max(axis=1) - t[:, :, 0][t[:, :, 0] > 0].min())
        print(f"  {what}: entry ticks min {e0.min()} median {int(np.median(e0))} max {e0.max()};  end ticks min {e1.min()} median {int(np.median(e1))} max {e1.max()}")
    half = g_ if g_ < tiles else tiles // 2
    # distributions: per tile, ticks from entry to end; the 100 MHz end stamp; by generation and by XCD (workgroup index mod 8)
    dur = t[:, :, 14].max(axis=1) - t[:, 0, 0]
    rte = (rt - rt.min()) / 100.0
    pct = lambda a: " ".join(f"{np.percentile(a, q):9.1f}" for q in (0, 10, 50, 90, 99, 100))
    for lo, hi, what in ((0, g_, "first tiles"), (g_, tiles, "second tiles")):
        if hi <= lo:
            continue
        print(f"  {what}: tile ticks   p0/10/50/90/99/100: {pct(dur[lo:hi])}")
        print(f"  {what}: end stamp us p0/10/50/90/99/100: {pct(rte[lo:hi])}")
        for x in range(8):
            sel = np.arange(lo, hi)[(np.arange(lo, hi) % g_) % 8 == x]
            print(f"     xcd {x}: ticks median {np.median(dur[sel]):8.0f} max {dur[sel].max():8.0f}   end us median {np.median(rte[sel]):6.1f} max {rte[sel].max():6.1f}")
    # the tiles that narrowed something (the lean round 0 ran a second pass: stamp 4 is later than stamp 11): where their extra time goes
    w0a = t[:, 0, :15] - t[:, 0, :1]
    rare = np.nonzero(w0a[:, 4] > w0a[:, 11])[0]
    if len(rare):
        d = lambda a, b: f"{np.median(w0a[rare, b] - w0a[rare, a]):8.0f}"
        print(f"  {len(rare)} tiles ran a second pass (median ticks): pass-0 vote -> pass-1 vote {d(11, 4)}   -> statuses again {d(4, 5)}   -> barrier {d(5, 6)}   -> write-back done (12) {d(6, 12)}"
              f"   -> its barrier (13) {d(12, 13)}   -> end {d(13, 14)};   all others 11 -> 12: {np.median(np.delete(w0a[:, 12] - w0a[:, 11], rare)):8.0f}")
    # per phase, per generation: distribution over tiles of (released k) - (released k-1) for the barrier-delimited phases
    for lo, hi, what in ((0, g_, "first tiles"), (g_, tiles, "second tiles")):
        if hi <= lo:
            continue
        w0 = t[lo:hi, 0, :15] - t[lo:hi, 0, :1]
        for a, b, nm in ((0, 1, "entry->zeroed"), (1, 3, "staging"), (3, 5, "list build"), (5, 8, "walk+round end"), (8, 11, "status"), (11, 14, "write-back, counters, hand-over")):
            d = w0[:, b] - w0[:, a]
            print(f"  {what}: {nm:32s} ticks p0/10/50/90/99/100: {pct(d)}")
    for lo, hi, what in ((0, half, "first half of the grid"), (half, tiles, "second half")):
        r = rel[lo:hi]
        print(f"-- {what}: ticks since wavefront 0 entered, mean over tiles: [wavefront 0] [earliest wavefront] [latest wavefront] (latest - earliest)")
        for k, nm in enumerate(names):
            col = r[:, :, k]
            print(f"  {k:2d} {nm:28s} w0 {col[:, 0].mean():8.0f}   min {col.min(axis=1).mean():8.0f}   max {col.max(axis=1).mean():8.0f}   spread {(col.max(axis=1) - col.min(axis=1)).mean():7.0f}   latest wave (mode) {np.bincount(col.argmax(axis=1)).argmax()}")
