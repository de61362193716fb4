#!/usr/bin/env python
"""bench.py — propagator filter-steps/sec to fixpoint on N-queens-1000 (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: `pcp_propagate_device` runs every open node of this rank's batch
to its propagation fixpoint (one kernel launch, inputs and outputs resident in HBM, in place like
`Store::consistency(&mut vstore)`).  The batch is this rank's share of the breadth-first frontier of the reference's own
search tree on N-queens n=1000 (FirstSmallestVar / MiddleVal / BinarySplit), `--nodes` open nodes per GPU: per-GPU work
is fixed as N grows (weak scaling); nodes are independent, so there is no collective in the data path.

What `value` is: filter steps EXECUTED per second — the (propagator, node) pairs the engine tested one by one on the node's own
domains (pcp_stats.evaluated), all ranks, divided by the max-over-ranks wall time of the K timed steps.  An all-XNeqY model
needs few of them: an XNeqY between two unassigned variables is a no-op (x_neq_y.rs:82-93), so the initial sweep of a node is
the adjacency lists of its assigned variables (pcp_neq.hip).  `config.steps_reference_equivalent_per_s` is the rate of pairs the
REFERENCE's scheduler would pop to reach the same fixpoints (every propagator of every node once, store.rs:144-149, plus every
wake-up) — a bookkeeping figure, not work done; `config.nodes_per_s` and `config.time_to_fixpoint_vs_cpu` compare time to the
same bit-identical fixpoints.  `roofline` is PHYSICAL: the bytes the ABI contract forces across HBM per launch (every node's
rows in once, the rows of changed nodes out once) / kernel time / 8 TB/s — always <= 1.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches it under torch.distributed.run
(one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.  `--mode search` runs BASELINE config 5 instead (the sharded
open-node worklist over RCCL, nodes/s).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak, /opt/skills/guides/MI355X_MICROARCH.md


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + f" ({os.cpu_count()} hw threads visible)"
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n, props, lb, ub, act, budget_s):
    """The oracle (a port: libpcp cannot be built here) timed on this host, single thread like the reference, on a
    bounded sample of the SAME batch: as many of its first nodes as fit in ~budget_s.  Returns the JSON object and the
    oracle's outputs for those nodes (bench.py compares the GPU's results with them: parity_checked_nodes)."""
    from oracle import oracle as orc
    om = orc.OracleModel(n, props)
    out, keep = {}, None
    for label, check in (("restatement-noassert", False), ("libpcp-restatement", True)):
        steps, nodes, t0, res = 0, 0, time.perf_counter(), []
        while nodes < lb.shape[0] and (time.perf_counter() - t0) < budget_s / 2:
            r = om.consistency(lb[nodes:nodes + 1], ub[nodes:nodes + 1], None if act is None else act[nodes:nodes + 1], check_dup=check)
            res.append(r[:4])
            steps += r[4]["steps"]
            nodes += 1
        dt = time.perf_counter() - t0
        out[label] = {"steps_per_s": steps / dt, "nodes": nodes, "seconds": dt, "nodes_per_s": nodes / dt}
        if not check:
            keep = res
    main = out["libpcp-restatement"]
    obj = {
        "value": main["steps_per_s"], "unit": "filter-steps/s", "cores": 1, "kind": "port",
        "sample": f"first {main['nodes']} nodes of the same batch, {main['seconds']:.1f} s, structure-faithful C++ restatement "
                  f"of libpcp incl. the duplicate-subscription assert; without that assert: {out['restatement-noassert']['steps_per_s']:.3e} steps/s "
                  f"over {out['restatement-noassert']['nodes']} nodes",
        "host_cpu": _cpu_name(),
        "nodes_per_s": main["nodes_per_s"], "nodes_per_s_noassert": out["restatement-noassert"]["nodes_per_s"],
    }
    return obj, keep


def box_ceilings(torch, t_a, t_b, stream):
    """Side measurements on THIS box, untimed (tools/micro/box_probe.hip, its own small library): the streaming-read ceiling over the very
    buffers a headline launch reads ("the measured ceiling of a plain copy/read kernel on the box") and the integer VALU
    issue rate.  Returns a dict, or {} when the probe library is not there."""
    import ctypes as C
    path = os.path.join(ROOT, "tools", "micro", "libbox_probe.so")
    if not os.path.exists(path):
        return {}
    lib = C.CDLL(path)
    lib.box_stream_read_ms.restype = C.c_float
    lib.box_stream_read_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.box_valu_issue.restype = C.c_double
    lib.box_valu_issue.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]
    nbytes = t_a.numel() * t_a.element_size()
    best = None
    for gpc, nt in ((2, 1), (2, 0), (8, 1)):
        ms = lib.box_stream_read_ms(t_a.data_ptr(), t_b.data_ptr(), nbytes, 10, gpc, nt, C.c_void_p(stream))
        if ms > 0 and (best is None or ms < best[0]):
            best = (ms, gpc, nt)
    out = {}
    if best:
        out.update(ceiling_gbs=2 * nbytes / best[0] / 1e6, ceiling_kernel_ms=best[0],
                   ceiling_note=f"read-only streaming kernel over the same two {nbytes / 1e6:.1f} MB buffers, 16 B per lane and load, 8 loads in flight per lane, "
                                f"{best[1]} x 256 threads per CU, {'nt' if best[2] else 'plain'} loads; best of three shapes, 10 passes each")
    cyc, mhz = C.c_double(), C.c_double()
    r = lib.box_valu_issue(0, 4, C.byref(cyc), C.byref(mhz), C.c_void_p(stream))
    cyc2 = C.c_double()
    r2 = lib.box_valu_issue(1, 4, C.byref(cyc2), C.byref(mhz), C.c_void_p(stream))
    if r > 0:
        out.update(valu_wave_inst_per_s_per_cu=r, valu_cycles_per_inst=cyc.value, valu_pk16_cycles_per_inst=cyc2.value if r2 > 0 else None, valu_clock_mhz=mhz.value)
    torch.cuda.synchronize()
    return out


def profiled_valu():
    """VALU wave-instructions per launch of the compute-bound legs from the committed rocprofv3 PMC passes (profiles/*valu_counts.json,
    tools/valu_json.py), newest file; {} when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*valu_counts.json")))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
        d["_file"] = os.path.relpath(files[-1], ROOT)
        return d
    except (OSError, ValueError):
        return {}


def profiled_traffic(tag_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*traffic.json, written
    by tools/profile_bench.sh + tools/traffic_json.py for exactly this workload), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if all(d.get(k) == v for k, v in tag_key.items()):
            best = d
    return best


class Leg:
    """One timed workload: K launches of pcp_propagate_device, each in place on its own fresh copy of the batch (staged in
    HBM beforehand), HIP-event time per launch, counters per launch, compulsory bytes per launch."""

    def __init__(self, ctx, torch, name, lb, ub, act, compulsory_bytes, note=""):
        self.ctx, self.torch, self.name, self.note = ctx, torch, name, note
        self.lb, self.ub, self.act = lb, ub, act
        self.n = lb.shape[0]
        self.bytes = compulsory_bytes
        self.status = torch.zeros(self.n, dtype=torch.uint8, device=lb.device)
        self.dirty = None  # optional pcp_device_batch.dirty_var ([n] int32)

    def out_bytes(self, per):
        """Compulsory HBM WRITE bytes per launch: the bounds rows (8 B per variable) of the nodes that changed, once — in place an
        unchanged node is not written.  The counters do not say how many nodes changed: at most min(nodes, narrowings)."""
        return 8 * self.lb.shape[1] * min(self.n, per["narrowings"])

    def run(self, launches=5, warmup=1):
        ctx, torch = self.ctx, self.torch
        stream = torch.cuda.current_stream().cuda_stream
        copies = [(self.lb.clone(), self.ub.clone(), None if self.act is None else self.act.clone()) for _ in range(launches + warmup)]
        for i in range(warmup):
            l, u, a = copies[i]
            ctx.propagate_device(self.n, l, u, l, u, a, a, self.status, stream, dirty=self.dirty)
        torch.cuda.synchronize()
        ctx.stats_reset(stream)
        ms = []
        for i in range(launches):
            l, u, a = copies[warmup + i]
            ctx.propagate_device(self.n, l, u, l, u, a, a, self.status, stream, dirty=self.dirty)
            ms.append(ctx.last_kernel_ms())
        st = ctx.stats_read(stream)
        self.last_out = copies[-1][:2]  # the rows the LAST timed launch left behind (parity checks compare these, not a re-run)
        per = {k: v / launches for k, v in st.items()}
        med = float(np.median(ms))
        steps = per["steps"] + per["steps3"]
        self.result = {
            "name": self.name, "nodes": self.n, "launches": launches,
            "kernel_ms": {"min": float(min(ms)), "median": med, "max": float(max(ms))},
            "steps_per_launch": steps, "evaluated_per_launch": per["evaluated"], "full_evals_per_launch": per["full_evals"],
            "narrowings_per_launch": per["narrowings"], "waves_per_node": per["waves"] / self.n,
            "steps_per_s": steps / (med * 1e-3), "evaluated_per_s": per["evaluated"] / (med * 1e-3),
            "nodes_per_s": self.n / (med * 1e-3),
            "compulsory_bytes_per_launch": self.bytes + self.out_bytes(per),
            "hbm_frac": (self.bytes + self.out_bytes(per)) / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "status_false_true_unknown": np.bincount(self.status.cpu().numpy(), minlength=3)[:3].tolist(),
            "plan": {k: v for k, v in ctx.last_plan().items() if k in ("nodes_per_block", "team", "packed", "word_level", "global_dom", "implicit_active", "grid")},
        }
        if self.note:
            self.result["note"] = self.note
        return self.result


def node_bytes(V, words, explicit):
    """Compulsory HBM READ bytes per node and launch: the (lb, ub) rows once (8 B per variable); with explicit `active` rows
    additionally the row once.  In place, the only writes the contract forces are the bounds that changed (8 B per narrowing,
    added by the caller from the counters) — a tile that narrows nothing writes nothing back."""
    return 8 * V + (8 * words if explicit else 0)


def run_search_mode(args, torch, dist, world, rank, dev):
    """BASELINE config 5: N-queens-n parallel subtree search, the open-node worklist sharded over the ranks and balanced
    GPU-to-GPU over RCCL (pcp_amd.distributed.parallel_search_device), on a fixed node budget."""
    import pcp_amd.engine as E
    from pcp_amd import model as M
    from pcp_amd import distributed as D
    from pcp_amd.search_device import DeviceSearch
    n = args.n
    ctx = E.Context(dev.index)
    set_mode = args.domains == "set"
    ctx.set_model(n, M.nqueens_props(n), set_words=(n + 63) // 64 if set_mode else 0)
    ctx.set_hull(1, n)
    for kv in args.opt:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    batch = args.search_batch
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    if set_mode or args.engine == "forest":
        return run_forest_mode(args, torch, dist, world, rank, dev, ctx, lb0, ub0, set_mode)
    ds = DeviceSearch(ctx, batch=batch, capacity=max(32 * batch, args.node_budget + 4 * batch), implicit=True, cells=args.cells)
    if world == 1 and not dist.is_initialized():
        # a single GPU still goes through the process group (RCCL with one rank): same driver, same exchange steps
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    grp = dist
    # warm-up: a short search (kernels loaded, buffers touched)
    D.parallel_search_device(ds, lb0, ub0, grp, all_solutions=True, node_limit=min(args.node_budget, 8 * batch), rounds_per_exchange=args.rounds_per_exchange, base=1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    info = {}
    nodes, sols, fails, steps, moved = D.parallel_search_device(ds, lb0, ub0, grp, all_solutions=True, node_limit=args.node_budget,
                                                                 rounds_per_exchange=args.rounds_per_exchange, info=info, base=1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t_dt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_dt, op=dist.ReduceOp.MAX)
    dt = float(t_dt.item())
    if rank == 0:
        _flush_c_stdio()
        emit_json({
            "metric": "propagator filter-steps/sec to fixpoint, N-queens-1000 (config 5: sharded open-node worklist)",
            "value": info.get("evaluated", 0) / dt, "unit": "filter-steps/s", "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": dt * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {
                "workload": f"N-queens n={n} parallel subtree search over " + ("IntervalSet<i32> domains (FDSpace, the reference's default)" if set_mode else "Interval<i32> domains")
                            + f", first {args.node_budget} nodes of the tree (all ranks together), device-resident stacks, "
                            f"batch {batch} nodes per round and GPU, worklist balanced every {args.rounds_per_exchange} rounds by all_gather + pairwise send/recv (RCCL)",
                "value_is": "filter steps EXECUTED per second, all ranks: (propagator, node) pairs tested one by one (pcp_stats.evaluated); "
                            "steps_reference_equivalent_per_s = the pairs the reference's scheduler would pop for the same nodes",
                "steps_reference_equivalent_per_s": steps / dt,
                "nodes": nodes, "nodes_per_s": nodes / dt, "solutions": sols, "failed_nodes": fails, "moved_records": moved,
                "exchange_seconds_rank0": info.get("exchange_s"), "exchange_share_rank0": (info.get("exchange_s") or 0) / dt, "exchanges": info.get("exchanges"),
                "record_bytes": (4 * n if args.cells else 8 * n) + (8 * n * ((n + 63) // 64) if set_mode else 0), "domains": args.domains,
                "node_format": "packed cells (PCP_CELLS_PACKED16)" if args.cells else "int32 rows",
                "parallelism": f"worklist sharded over {world} GPU(s)",
            },
        })


def c5_legs(args, torch, dist, world, rank, dev, n):
    """--gpus N > 1, appended to the headline run (every rank calls it): BASELINE config 5 on a fixed node budget, twice —
    the sharded open-node WORKLIST (distributed.parallel_search_device: all_gather of the stack sizes + pairwise send/recv of node
    records over RCCL every few rounds) and the interval FOREST with its cross-rank refill.  Returns flat keys (rank 0 prints them):
    c5_nps nodes/s of the worklist engine, c5_xchg_share = the slowest rank's share of wall time inside the exchange step,
    c5_moved node records that changed GPU, c5_moved_mb; c5f_nps / c5f_moved the forest's."""
    import pcp_amd.engine as E
    from pcp_amd import model as M
    from pcp_amd import distributed as D
    from pcp_amd.search_device import DeviceSearch
    from pcp_amd.search_forest import forest_search
    ctx = E.Context(dev.index)
    ctx.set_model(n, M.nqueens_props(n))
    ctx.set_hull(1, n)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    batch, budget = args.search_batch, args.c5_budget
    out = {}

    def timed(fn):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(); dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return r, float(t.item())

    # the worklist engine twice: open nodes as int32 rows (c5_i32_*), then as rows of packed cells (c5_*: pcp_device_batch.cell_format
    # PCP_CELLS_PACKED16, pcp_branch_device_cells — half the bytes per open node and per record moved between ranks).  Same budget, same batch,
    # same exchanges: the two must count the same tree.
    totals = {}
    for fmt, cells in (("i32", False), ("cells", True)):
        if cells and args.no_cells:
            continue
        # an Unknown node adds one open node net, so the stack can hold up to `budget` rows before the search ends — but never ask for more than
        # a quarter of the free HBM (17 GB of int32 rows at the defaults; a smaller or shared GPU gets a smaller stack, and a stack
        # that does fill up is reported as c5_error, not as a lost headline line)
        rec = (4 if cells else 8) * n + 4
        free_b = torch.cuda.mem_get_info(dev)[0]
        capacity = int(min(max(32 * batch, budget + 4 * batch), max(8 * batch, free_b // 4 // rec)))
        ds = DeviceSearch(ctx, batch=batch, capacity=capacity, implicit=True, cells=cells)
        D.parallel_search_device(ds, lb0, ub0, dist, all_solutions=True, node_limit=min(budget, 8 * batch * world), rounds_per_exchange=args.rounds_per_exchange, base=1)
        info = {}
        (nodes, sols, fails, steps, moved), dt = timed(lambda: D.parallel_search_device(ds, lb0, ub0, dist, all_solutions=True, node_limit=budget,
                                                                                      rounds_per_exchange=args.rounds_per_exchange, info=info, base=1))
        x = torch.tensor([info.get("exchange_s", 0.0)], dtype=torch.float64, device=dev)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        totals[fmt] = (int(nodes), int(sols), int(fails))
        if cells or args.no_cells:
            out.update(c5_nps=float(f"{nodes / dt:.4g}"), c5_xchg_share=round(float(x.item()) / dt, 4), c5_moved=int(moved),
                       c5_moved_mb=round(moved * info.get("record_bytes", 8 * n) / 1e6, 2), c5_nodes=int(nodes), c5_ms=round(dt * 1e3, 2), c5_exchanges=int(info.get("exchanges", 0)),
                       c5_fmt="cells" if cells else "i32", c5_rec_bytes=int(info.get("record_bytes", 0)))
        else:
            out.update(c5_i32_nps=float(f"{nodes / dt:.4g}"), c5_i32_ms=round(dt * 1e3, 2))
        del ds
        torch.cuda.empty_cache()
    if len(totals) == 2:
        # (reported, not raised: the headline record is complete by now and must not be lost to a side leg — ADVICE r5; 0 is a parity failure)
        out["c5_same_tree"] = int(totals["i32"] == totals["cells"])
        if not out["c5_same_tree"]:
            out["c5_error"] = f"PARITY FAILURE (c5 worklist): packed cells {totals['cells']} != int32 rows {totals['i32']} (nodes, solutions, failures)"
            print(out["c5_error"], file=sys.stderr, flush=True)
    trees = args.trees if args.trees else 4096
    spl = args.steps_per_launch if args.steps_per_launch else 1024
    # The forest's stacks grow on demand INSIDE the timed region (with a process group nothing is reserved: 4 -> 8 -> 16 GB): take that memory from
    # the driver once, outside it, and hand it to torch's caching allocator, so that the timed growth steps are splits of a cached block and not
    # hipMalloc calls (on one box of round 6's last pass the same leg took 0.9 s instead of 34 ms, twice in a row, for exactly the same launches)
    pre = torch.empty(int(min(torch.cuda.mem_get_info(dev)[0] // 2, 64 << 30)), dtype=torch.uint8, device=dev)
    del pre
    forest_search(ctx, lb0, ub0, node_limit=4 * trees * world, n_trees=trees, steps_per_launch=4, rank=rank, world=world, dist=dist)
    finfo = {}
    fr, dtf = timed(lambda: forest_search(ctx, lb0, ub0, node_limit=min(8 * budget, 2_097_152), n_trees=trees, steps_per_launch=spl, rank=rank, world=world, dist=dist, info=finfo))
    tot = torch.tensor([fr["nodes"], finfo.get("moved_rows", 0), fr["error"]], dtype=torch.int64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    fn, fm, fe = (int(v) for v in tot.tolist())
    out.update(c5f_nps=float(f"{fn / dtf:.4g}"), c5f_moved=fm, c5f_nodes=fn, c5f_ms=round(dtf * 1e3, 2), c5f_err=fe)
    return out


def run_forest_mode(args, torch, dist, world, rank, dev, ctx, lb0, ub0, set_mode=True):
    """--mode search (--engine forest, and always with --domains set): every rank expands the root to the same frontier (no communication), takes the open nodes
    r, r + world, ... and searches each as a tree in one CU's LDS with an undo trail (pcp_amd.search_forest); one all_reduce of
    the counters at the end.  Interval forest with N > 1: the node budget is global and a rank whose trees ran dry is refilled from the
    others between launches (search_forest.refill_across_ranks; N-queens-1000: no subtree ends within the budget, so nothing moves)."""
    from pcp_amd.search_forest import forest_search, forest_search_set
    n = args.n
    trees = args.trees if args.trees else (512 if set_mode else 4096)
    spl = args.steps_per_launch if args.steps_per_launch else (2048 if set_mode else 1024)
    info = {}

    # one rank: a tree's stack never needs more rows than its share of the budget + 64; the warm-up search allocates stacks of exactly that
    # size, so that the timed search finds them in torch's allocator cache instead of paying for an 18 GB hipMalloc inside the timed region
    # (47 ms or 290 ms for the same search, depending on the box, when it did).  Several ranks: the stacks start small and grow.
    cap1 = 0 if (set_mode or world > 1) else -(-args.node_budget // trees) + 64

    def search(limit, steps):
        if set_mode:
            return forest_search_set(ctx, lb0, ub0, 1, node_limit=limit, n_trees=trees, steps_per_launch=steps, rank=rank, world=world, info=info)
        return forest_search(ctx, lb0, ub0, node_limit=limit, n_trees=trees, steps_per_launch=steps, rank=rank, world=world,
                             dist=dist if world > 1 else None, info=info, capacity=cap1)

    search(4 * trees * world, 4)  # warm-up
    torch.cuda.synchronize()
    ctx.stats_reset()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    fr = search(args.node_budget, spl)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    fs = ctx.stats_read()
    tot = torch.tensor([fr["nodes"], fr["solutions"], fr["failed"], fs["evaluated"], fs["steps"] + fs["steps3"], fr["trees"], fr["error"]], dtype=torch.int64, device=dev)
    t_dt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(t_dt, op=dist.ReduceOp.MAX)
    dt = float(t_dt.item())
    nodes, sols, fails, evaluated, steps, ntrees, err = (int(x) for x in tot.tolist())
    if rank == 0:
        _flush_c_stdio()
        emit_json({
            "metric": "propagator filter-steps/sec to fixpoint, N-queens-1000 (config 5: parallel subtree search" + (" over FDSpace)" if set_mode else ")"),
            "value": evaluated / dt, "unit": "filter-steps/s", "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": dt * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64" if set_mode else "i32", "data": "synthetic",
            "config": {
                "workload": f"N-queens n={n} parallel subtree search over " + ("IntervalSet<i32> domains (FDSpace, the reference's default)" if set_mode else "Interval<i32> domains")
                            + f", about the first {args.node_budget} nodes of the tree (all ranks together; `nodes` is the exact count): the root is expanded breadth-first to {ntrees} open nodes, "
                            "each the root of a tree searched depth-first by one workgroup ("
                            + ("current node in LDS, undo trail in HBM: pcp_dfs_forest_device_set" if set_mode else "current node in LDS, stack rows in HBM: pcp_dfs_forest_device")
                            + "); no data-path collective",
                "value_is": "filter steps EXECUTED per second, all ranks (pcp_stats.evaluated); steps_reference_equivalent_per_s = every propagator of every node once "
                            "(init_scheduler) plus the wake-ups",
                "steps_reference_equivalent_per_s": steps / dt, "nodes": nodes, "nodes_per_s": nodes / dt, "solutions": sols, "failed_nodes": fails,
                "trees": ntrees, "steps_per_launch": spl, "launches_rank0": fr["launches"], "error": err, "engine": "forest",
                "trail_entries_max_rank0": info.get("trail_max"), "levels_max_rank0": info.get("levels_max"), "domains": "set" if set_mode else "interval",
                "stack_rows_per_tree_rank0": info.get("capacity"), "stack_grown_rank0": info.get("grown"), "moved_rows_rank0": info.get("moved_rows"),
                "exchange_seconds_rank0": info.get("exchange_s"),
                "parallelism": f"subtrees sharded over {world} GPU(s)",
            },
        })


_JSON_FD = None


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_local_ranks(args) -> int:
    """`python bench.py --gpus N` with WORLD_SIZE unset: start N local ranks of this same command, one process per device (RANK = LOCAL_RANK =
    0..N-1, WORLD_SIZE = N, MASTER_ADDR 127.0.0.1, a free MASTER_PORT), exactly the environment torch.distributed.run would set.  Rank 0
    inherits stdout (its ONE JSON line is this command's output); the other ranks' stdout goes to stderr.  Fails loudly when fewer than N
    devices are visible.  Returns the exit code: 0 iff every rank returned 0; a rank that fails takes the others down."""
    import subprocess
    n = int(args.gpus)
    if not args.spawn_dry_run:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py --gpus {n}: only {have} device(s) visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?) — not starting", file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, os.path.abspath(__file__), *sys.argv[1:]]
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        for p_ in list(alive):
            code = p_.poll()
            if code is None:
                continue
            alive.remove(p_)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {procs.index(p_)} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in alive:  # (exact PIDs this function started)
                    q.terminate()
        if alive:
            time.sleep(0.05)
    return rc


def spawn_dry_run_rank(args) -> None:
    """One self-spawned rank of --spawn-dry-run: the rendezvous and one all_reduce over gloo on CPU; rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if world != args.gpus or int(os.environ["LOCAL_RANK"]) != rank:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world} RANK={rank} LOCAL_RANK={os.environ['LOCAL_RANK']}")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1, 1], dtype=torch.int64)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        emit_json({"dry_run": True, "n_gpus": world, "ranks": int(t[1].item()), "rank_sum": int(t[0].item()), "master_port": int(os.environ["MASTER_PORT"])})
    dist.destroy_process_group()


def emit_json(obj) -> None:
    """Write the result line to the process's original stdout (see main)."""
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    _flush_c_stdio()
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def _flush_c_stdio():
    """librccl prints a version banner through C stdio when the process group comes up; flush it now so that the JSON line is
    the LAST line of rank 0's stdout."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1000, help="N-queens size (BASELINE config: 1000)")
    ap.add_argument("--nodes", type=int, default=16384, help="open nodes per GPU per step")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline work (rank 0, N=1 only; 0 = skip)")
    ap.add_argument("--active", choices=["implicit", "explicit"], default="implicit",
                    help="node format of the headline leg: domains only (liveness derived) or domains + `active` rows")
    ap.add_argument("--legs", default="auto", help="'auto' = all side legs at N=1, 'none', or a comma list of: cells,wide,mix,explicit,c2,forest,deep500,deep3000,set,setsearch,c3,c4,f4")
    ap.add_argument("--share", type=int, default=-1, help="which share of the frontier this process runs (default: its rank)")
    ap.add_argument("--nodes-per-block", type=int, default=0)
    ap.add_argument("--block-threads", type=int, default=1024)
    ap.add_argument("--mode", choices=["propagate", "search"], default="propagate")
    ap.add_argument("--node-budget", type=int, default=2_000_000, help="--mode search: nodes of the tree to explore (all ranks together)")
    ap.add_argument("--search-batch", type=int, default=16384, help="worklist engine: open nodes per round and GPU (4096: 2.0e7 nodes/s on one MI355X, 16384: 3.4e7, 65536: 5.1e7 — a round is two launches and one 20-byte read-back)")
    ap.add_argument("--rounds-per-exchange", type=int, default=4)
    ap.add_argument("--c5-timeout", type=float, default=240.0, help="seconds after which the config-5 legs are given up and the headline line is printed without them")
    ap.add_argument("--c5-single", action="store_true", help="(default since round 6; kept so that old command lines still parse) run the config-5 legs on one GPU too")
    ap.add_argument("--no-c5", action="store_true", help="skip the config-5 legs (worklist engine through RCCL, interval forest with cross-rank refill); by default they run "
                                                         "behind the headline at every --gpus N, on one GPU through a one-rank RCCL group")
    ap.add_argument("--c5-budget", type=int, default=2097152,
                    help="--gpus N > 1: nodes of the short config-5 leg appended to the headline run (worklist engine; the forest leg runs 8x as many); 0 = skip")
    ap.add_argument("--cells", action="store_true", help="--mode search --engine worklist: keep the open nodes as rows of packed cells (cell_format PCP_CELLS_PACKED16)")
    ap.add_argument("--no-cells", action="store_true", help="config-5 worklist leg (--gpus N > 1, --c5-single): int32 rows only (default: both formats, c5_* = packed cells)")
    ap.add_argument("--engine", choices=["forest", "worklist"], default="forest",
                    help="--mode search: forest = one in-kernel DFS per open node of a frontier, no exchange (default; the only engine for --domains set); "
                         "worklist = batched rounds with the open-node stacks balanced GPU-to-GPU over RCCL")
    ap.add_argument("--trees", type=int, default=0, help="--mode search, forest: trees (workgroups) per GPU (0 = 4096 for intervals, 512 for sets)")
    ap.add_argument("--steps-per-launch", type=int, default=0, help="--mode search, forest: nodes per tree and launch (0 = 1024 for intervals, 2048 for sets)")
    ap.add_argument("--domains", choices=["interval", "set"], default="interval",
                    help="--mode search: Interval<i32> domains, or IntervalSet<i32> (the reference's FDSpace: what example/src/nqueens.rs runs)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="(experiments) pcp_set_option on the context after the model is set, e.g. --opt neq_persist=0; repeatable")
    ap.add_argument("--spawn-dry-run", action="store_true",
                    help="(tests) the self-spawned ranks only bring up a gloo group on CPU, all_reduce their ranks and rank 0 prints a small JSON line: checks the launcher without a GPU")
    args = ap.parse_args()

    # `python bench.py --gpus N` by itself (no torchrun: WORLD_SIZE unset) starts its own N ranks, one process per device; under
    # torch.distributed.run the environment is already there and this process IS a rank.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_local_ranks(args))
    # The ONE JSON line is the only thing that reaches the real stdout: RCCL prints a version banner on fd 1 when a process group comes
    # up (gloo its connection report), so fd 1 is pointed at stderr for the rest of the run and the line is written to the saved descriptor.
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if args.spawn_dry_run:
        return spawn_dry_run_rank(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as g
    if rank == 0:
        g.build()  # a no-op when the prebuilt libraries match the sources (content hash)
    if world > 1:
        dist.barrier()
    if args.mode == "search":
        run_search_mode(args, torch, dist, world, rank, dev)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    import pcp_amd.engine as E
    from pcp_amd import model as M
    from pcp_amd import workloads as W

    n = args.n
    props = M.nqueens_props(n)
    ctx = E.Context(local_rank)
    ctx.set_model(n, props)
    ctx.set_hull(1, n)  # the queens were allocated with Interval(1, n)  (example/src/nqueens.rs:32-35)
    ctx.set_option("block_threads", args.block_threads)
    ctx.set_option("nodes_per_block", args.nodes_per_block)
    implicit = args.active == "implicit"
    V, words = n, ctx.words

    # ---- synthetic input: this rank's share of the breadth-first frontier (pcp_amd.workloads.nqueens_frontier) --------
    shares = max(8, world)
    share = rank if args.share < 0 else args.share % shares
    L, U, A = W.nqueens_frontier(ctx, n, args.nodes, share=share, shares=shares, implicit=implicit)
    t_lb_in = torch.from_numpy(L).to(dev)
    t_ub_in = torch.from_numpy(U).to(dev)
    t_act_in = None if A is None else torch.from_numpy(A.view(np.int64)).to(dev)
    t_status = torch.zeros(args.nodes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    n_probe = min(args.steps, 10)  # untimed steps after the timed ones, for the per-launch HIP-event times
    total_steps = args.warmup + args.steps + n_probe
    # In place, as the reference's Store::consistency(&mut vstore) works and as the search loop calls the engine: every
    # step gets its OWN copy of the frontier, staged in HBM before the timed region.  If K exceeds what fits, the pool is
    # cycled and the later steps see already-propagated nodes — reported as fresh_inputs < steps.
    copy_bytes = t_lb_in.numel() * 8 + (0 if t_act_in is None else t_act_in.numel() * 8)
    free_b, _ = torch.cuda.mem_get_info(dev)
    n_pool = max(1, min(total_steps, int(free_b * 0.5 // copy_bytes)))
    pool = [(t_lb_in.clone(), t_ub_in.clone(), None if t_act_in is None else t_act_in.clone()) for _ in range(n_pool)]
    step_no = [0]

    def step():
        lb_, ub_, act_ = pool[step_no[0] % len(pool)]
        step_no[0] += 1
        ctx.propagate_device(args.nodes, lb_, ub_, lb_, ub_, act_, act_, t_status, stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region runs the call as a search would issue it: the kernel and nothing else in the queue (option time_kernels 0: no HIP events
    # around the launch).  The per-launch kernel times of `roofline` come from the untimed probe pass below, with the events back on.
    ctx.set_option("time_kernels", 0)
    for _ in range(args.warmup):
        step()
    barrier()
    ctx.stats_reset(stream)
    torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # (torch's current stream: the stream the launches go to)
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events over the timed region: the kernel's average launch duration, back to back
    ctx.set_option("time_kernels", 1)
    # HIP-event time of the fixpoint kernel of each step would need a sync per step; take it from a second, untimed pass
    # over the same steps so that the timed region stays free of host syncs.
    kernel_ms = []
    for _ in range(n_probe):
        step()
        kernel_ms.append(ctx.last_kernel_ms())
    st = ctx.stats_read(stream)
    plan = ctx.last_plan()
    per_step = {k: v / (args.steps + len(kernel_ms)) for k, v in st.items()}  # stats cover steps + probes, all identical
    steps_rank = (per_step["steps"] + per_step["steps3"]) * args.steps
    eval_rank = per_step["evaluated"] * args.steps
    full_rank = per_step["full_evals"] * args.steps

    t_dt = torch.tensor([dt], dtype=torch.float64, device=dev)
    t_cnt = torch.tensor([steps_rank, eval_rank, full_rank], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_dt, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_cnt, op=dist.ReduceOp.SUM)
    dt_max = float(t_dt.item())
    steps_all, eval_all, full_all = (float(x) for x in t_cnt.tolist())

    status = t_status.cpu().numpy()
    out, flat, legs = None, {}, []
    if rank == 0:
        k_med = float(np.median(kernel_ms))  # one event pair around each of the probe launches (a pair costs the queue a few microseconds)
        k_ms = region_ms                     # the roofline's duration: HIP events over the timed region / launches
        compulsory = args.nodes * node_bytes(V, words, not implicit) + 8 * V * min(args.nodes, per_step["narrowings"])
        tr = profiled_traffic({"n": n, "nodes_per_launch": args.nodes, "active": args.active})
        achieved = compulsory / (k_ms * 1e-3) / 1e9
        box = box_ceilings(torch, pool[0][0], pool[0][1], stream) if world == 1 else {}
        out = {
            "metric": "propagator filter-steps/sec to fixpoint, N-queens-1000",
            "value": eval_all / dt_max,
            "unit": "filter-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32",
            "data": "synthetic",
            "config": {
                "workload": f"N-queens n={n} (V={n}, P={len(props)} XNeqY), Interval<i32> domains (not the reference's default IntervalSet mode: see legs), "
                            f"{args.nodes} open nodes per GPU per step = this rank's share of the BFS frontier, 1 launch per step, in place on a fresh copy; "
                            + ("implicit-active nodes (domains only)" if implicit else "explicit `active` rows (183 KB per node)"),
                "value_is": "filter steps EXECUTED per second: (propagator, node) pairs tested one by one on the node's own domains (pcp_stats.evaluated); "
                            "steps_reference_equivalent_per_s = pairs the reference's scheduler would pop for the same fixpoints (bookkeeping, not work)",
                "steps_reference_equivalent_per_s": steps_all / dt_max,
                "steps_evaluated_per_s": eval_all / dt_max,
                "full_filter_evals_per_s": full_all / dt_max,
                "nodes_per_s": args.nodes * world * args.steps / dt_max,
                "nodes_per_gpu": args.nodes,
                "active_rows": args.active,
                "plan": plan,
                "domain_cells": "i32 bounds in/out; 16-bit packed (-lb, ub) LDS cells under the declared hull [1,n]",
                "fresh_inputs": max(0, min(args.steps, n_pool - args.warmup)),
                "filter_steps_per_step_per_gpu": per_step["steps"] + per_step["steps3"],
                "evaluated_per_step_per_gpu": per_step["evaluated"],
                "full_evals_per_step_per_gpu": per_step["full_evals"],
                "narrowings_per_step_per_gpu": per_step["narrowings"],
                "fixpoint_waves_per_node": per_step["waves"] / args.nodes,
                "status_counts_false_true_unknown": np.bincount(status, minlength=3)[:3].tolist(),
                "kernel_ms_per_launch": {"min": float(min(kernel_ms)), "median": k_med, "max": float(max(kernel_ms)), "launches": len(kernel_ms),
                                         "note": "untimed probe launches, each bracketed by its own pair of HIP events"},
                "kernel_ms_timed_region": region_ms,
                "parallelism": f"nodes sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": (tr or {}).get("traffic_bytes"),
                "traffic_source": (tr or {}).get("source"),
                "kernel": "pcp::neqfix_kernel" if plan.get("path") == 1 else "pcp::fixpoint_kernel", "kernel_ms": k_ms,
                "compulsory_bytes_per_launch": compulsory,
                "ceiling_gbs": box.get("ceiling_gbs"), "frac_of_ceiling": (achieved / box["ceiling_gbs"]) if box.get("ceiling_gbs") else None,
                "ceiling_note": box.get("ceiling_note"),
                "valu_cycles_per_wave_inst": box.get("valu_cycles_per_inst"), "valu_pk16_cycles_per_wave_inst": box.get("valu_pk16_cycles_per_inst"),
                "model": "achieved = compulsory HBM bytes per launch (every node's lb/ub rows read once"
                         + ("" if implicit else " + its `active` row")
                         + ", the rows of changed nodes written once) / the kernel's average launch duration = HIP events around the K timed launches / K (kernel_ms); traffic = 2 x FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 PMC passes, per launch",
                "algorithmic_note": "SURVEY.md 8d prices a filter step at 28 B (12 B descriptor + two 8 B domains): those bytes are LDS reads here (the domains of a tile "
                                    "stay in LDS for the whole fixpoint), not HBM traffic; the roofline is the bytes the contract forces across HBM",
            },
        }
        legs_req = args.legs
        if legs_req == "auto":
            legs_req = "cells,wide,mix,explicit,c2,forest,deep500,deep3000,set,setsearch,c3,c4,f4" if world == 1 else "none"
        legs = []
        if world == 1 and args.cpu_budget > 0:
            out["cpu_baseline"], ref = cpu_baseline(n, props, L, U, A, args.cpu_budget)
            # parity of the measured launch itself: the nodes the CPU leg just computed, against the GPU's results for the
            # same nodes (pool[0] was propagated in place by the first warm-up step)
            k = len(ref)
            g_lb, g_ub = pool[0][0][:k].cpu().numpy(), pool[0][1][:k].cpu().numpy()
            g_act = None if pool[0][2] is None else pool[0][2][:k].cpu().numpy().view(np.uint64)
            if implicit:  # materialise the `active` rows of those nodes on request
                _, _, g_act, _, _ = ctx.propagate_implicit(L[:k], U[:k], want_active=True)
            for i, (r_lb, r_ub, r_act, r_st) in enumerate(ref):
                if int(r_st[0]) != int(status[i]):
                    raise SystemExit(f"PARITY FAILURE: node {i} status {int(status[i])} != oracle {int(r_st[0])}")
                if int(r_st[0]) != 0 and not (np.array_equal(r_lb[0], g_lb[i]) and np.array_equal(r_ub[0], g_ub[i]) and np.array_equal(r_act[0], g_act[i])):
                    raise SystemExit(f"PARITY FAILURE: node {i} differs from the oracle")
            out["config"]["parity_checked_nodes"] = k
            cb = out["cpu_baseline"]
            gpu_nps = out["config"]["nodes_per_s"]
            out["config"]["time_to_fixpoint_vs_cpu"] = {
                "gpu_nodes_per_s": gpu_nps, "cpu_nodes_per_s": cb["nodes_per_s"], "cpu_nodes_per_s_noassert": cb["nodes_per_s_noassert"],
                "speedup": gpu_nps / cb["nodes_per_s"], "speedup_noassert": gpu_nps / cb["nodes_per_s_noassert"],
                "note": "same nodes, bit-identical fixpoints (parity_checked_nodes): GPU nodes/s of the timed steps vs the single-thread restatement of libpcp on this host",
            }
        if legs_req != "none":
            legs = side_legs(ctx, torch, dev, n, props, args, set(legs_req.split(",")), L, U)
        # The ONE JSON line carries every BASELINE configuration as SHORT flat scalar keys, most important first and ahead of the long
        # text fields (a parser that truncates keys or caps their number then still keeps them); the full leg records go to stderr
        # and to gpurun_out/bench_legs.json.
        by = {l["name"]: l for l in legs}
        flat = {}
        def put_ms(key, name):
            if name in by and isinstance(by[name].get("kernel_ms"), dict):
                flat[key] = round(by[name]["kernel_ms"]["median"], 4)
        def put_k(key, name, field, digits=4):
            if name in by and by[name].get(field) is not None:
                flat[key] = float(f"{by[name][field]:.{digits}g}")
        put_ms("mix_ms", "MIX-nodes-along-the-dfs"); put_k("mix_nps", "MIX-nodes-along-the-dfs", "nodes_per_s"); put_k("mix_sps", "MIX-nodes-along-the-dfs", "evaluated_per_s")
        put_k("mix_frac", "MIX-nodes-along-the-dfs", "hbm_frac", 3)
        put_ms("pk_ms", "C2-frontier-resident-as-packed-cells"); put_k("pk_nps", "C2-frontier-resident-as-packed-cells", "nodes_per_s"); put_k("pk_frac", "C2-frontier-resident-as-packed-cells", "hbm_frac", 3)
        put_ms("w8_ms", "C2-frontier-eight-shares-one-launch"); put_k("w8_nps", "C2-frontier-eight-shares-one-launch", "nodes_per_s"); put_k("w8_frac", "C2-frontier-eight-shares-one-launch", "hbm_frac", 3)
        put_ms("mixh_ms", "MIXH-children-with-dirty-var-hints"); put_ms("mixc_ms", "MIXC-children-no-hints"); put_k("mixh_nps", "MIXH-children-with-dirty-var-hints", "nodes_per_s")
        put_ms("d500_ms", "C2-deep-dive-500"); put_ms("d3000_ms", "C2-deep-dive-3000")
        put_ms("c3_ms", "C3-random-binary-csp-50k-vars-500k-props"); put_k("c3_sps", "C3-random-binary-csp-50k-vars-500k-props", "evaluated_per_s")
        put_ms("c4_ms", "C4-golomb-distinct-sum-network"); put_ms("f4_ms", "F4-cumulative-reified-layer")
        put_ms("set_ms", "C2-set-mode-IntervalSet-frontier"); put_k("set_frac", "C2-set-mode-IntervalSet-frontier", "hbm_frac", 3)
        put_ms("expl_ms", "C2-frontier-explicit-active-rows"); put_k("expl_frac", "C2-frontier-explicit-active-rows", "hbm_frac", 3)
        put_k("forest_nps", "C5-interval-forest", "nodes_per_s"); put_k("setforest_nps", "C2-set-mode-device-search", "nodes_per_s")
        put_k("forest8k_nps", "C5-interval-forest-8192-trees", "nodes_per_s")
        put_k("forest_chk", "C5-interval-forest", "parity_checked_nodes"); put_k("setf_chk", "C2-set-mode-device-search", "parity_checked_nodes")
        put_k("dfs_us_node", "C2-dfs-256-device-side-stack", "us_per_node"); put_k("c2_us_node", "C2-dfs-256-one-node-per-call", "us_per_node")
        # The legs that HBM does not bound get their roofline against the bound they do have: integer VALU issue.  <leg>_vfrac = VALU
        # wave-instructions per launch (committed PMC pass of the same launch, profiles/*valu_counts.json) / this run's launch time / (the VALU
        # issue rate measured on this box just now x CUs).  1.0 would be every SIMD issuing a VALU instruction whenever it can.
        pv = profiled_valu()
        rate = box.get("valu_wave_inst_per_s_per_cu") if world == 1 else None
        if pv and rate:
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            def put_v(key, prof_leg, ms):
                if prof_leg in pv and ms and pv[prof_leg].get("valu_per_launch"):
                    flat[key] = float(f"{pv[prof_leg]['valu_per_launch'] / (ms * 1e-3) / (rate * ncu):.3g}")
            put_v("hl_vfrac", "frontier", k_ms)
            for key, prof_leg, name in (("mix_vfrac", "mix", "MIX-nodes-along-the-dfs"), ("d500_vfrac", "deep500", "C2-deep-dive-500"), ("d3000_vfrac", "deep3000", "C2-deep-dive-3000"),
                                        ("c3_vfrac", "c3", "C3-random-binary-csp-50k-vars-500k-props"), ("c4_vfrac", "c4", "C4-golomb-distinct-sum-network")):
                if name in by and isinstance(by[name].get("kernel_ms"), dict):
                    put_v(key, prof_leg, by[name]["kernel_ms"]["median"])
            out["roofline"]["valu_bound_note"] = (f"*_vfrac keys in config: bound 'valu' — VALU wave-instructions per launch from {pv.get('_file')} over this run's launch time, against "
                                                  f"{rate:.3e} wave-instructions/s/CU x {ncu} CUs measured on this box (v_add_u32, four wavefronts per SIMD: {box.get('valu_cycles_per_inst'):.2f} SIMD cycles "
                                                  f"per instruction at the reported clock; v_pk_add_u16: {box.get('valu_pk16_cycles_per_inst') or 0:.2f})")
    # ---- --gpus N > 1: the config-5 legs that exercise RCCL, AFTER the headline record is complete and under a watchdog: if a rank fails or a
    # collective hangs (this path cannot be run on more than one GPU where it was written), rank 0 still prints the headline line, with
    # `c5_error` in place of the c5 keys, and every rank leaves.
    c5 = {}
    if args.c5_budget > 0 and not args.no_c5 and (world > 1 or args.c5_single or args.legs == "auto"):
        import threading

        def finish(extra):
            if rank == 0:
                out["config"] = {**front_keys({**flat, **extra}), **out["config"]}
                emit_json(out)
            _flush_c_stdio()
            os._exit(0)

        dog = threading.Timer(args.c5_timeout, lambda: finish({"c5_error": f"no result after {args.c5_timeout:.0f} s (a collective did not return)"}))
        dog.daemon = True
        dog.start()
        try:
            if not dist.is_initialized():  # --c5-single: the same legs through a one-rank RCCL group (a 1-GPU box can check everything but the transfers)
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29543")
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            pool.clear()
            torch.cuda.empty_cache()
            c5 = c5_legs(args, torch, dist, world, rank, dev, n)
        except Exception as e:  # this rank is out of the collectives: it reports (rank 0) and leaves; the others' watchdogs end them
            dog.cancel()
            print(f"config-5 legs failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
            finish({"c5_error": f"rank {rank}: {type(e).__name__}: {str(e)[:160]}"})
        dog.cancel()
    if rank == 0:
        flat.update(c5)
        out["config"] = {**front_keys(flat), **out["config"]}
        if legs:
            full = json.dumps({"legs": legs})
            print(full, file=sys.stderr, flush=True)
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "bench_legs.json"), "w") as f:
                    f.write(full + "\n")
            except OSError:
                pass
        _flush_c_stdio()
        emit_json(out)
    if dist.is_initialized():
        dist.destroy_process_group()
    _flush_c_stdio()


# The flat keys in the order of what a reader of the ONE line needs first — the driver's parser keeps only the first two dozen keys of `config`
# (BENCH_r05.json: everything behind `forest_nps` was cut): config 5's worklist engine (VERDICT r5 #1c), the search-node and wide-batch legs, then
# one figure per BASELINE configuration, then the rest.
FRONT_KEYS = ("c5_nps", "c5_i32_nps", "c5_same_tree", "c5_xchg_share", "c5f_nps", "c5_error", "mix_ms", "mixh_ms", "w8_frac", "w8_ms", "pk_ms", "pk_frac",
              "c3_ms", "d500_ms", "d3000_ms", "c4_ms", "f4_ms", "set_ms", "set_frac", "expl_ms", "expl_frac", "forest_nps", "setforest_nps", "forest8k_nps")


def front_keys(flat):
    return {**{k: flat[k] for k in FRONT_KEYS if k in flat}, **{k: v for k, v in flat.items() if k not in FRONT_KEYS}}


def side_legs(ctx, torch, dev, n, props, args, want, L, U):
    """The other BASELINE configurations, each timed by HIP events on a few launches."""
    import pcp_amd.engine as E
    from pcp_amd import model as M
    from pcp_amd import workloads as W
    from pcp_amd.search_device import DeviceSearch
    legs = []
    V, words = n, ctx.words

    def reset_opts():
        for k, v in {"force_path": 0, "nodes_per_block": 0, "team": 0, "global_dom": 0, "packed": 1, "word_level": 1}.items():
            ctx.set_option(k, v)

    if "cells" in want and ctx.last_plan().get("path") == 1:
        # the headline frontier RESIDENT AS PACKED CELLS (pcp_device_batch.cell_format PCP_CELLS_PACKED16): 4 B per variable in HBM instead of
        # 8, the rows are the kernel's LDS format.  Same nodes as the headline, so same results: unpacked and compared with an int32 launch.
        reset_opts()
        ctx.set_option("nodes_per_block", args.nodes_per_block)
        stream = torch.cuda.current_stream().cuda_stream
        lb_h, ub_h = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
        cells0 = ctx.pack_rows(lb_h, ub_h, stream_ptr=stream)
        nn = L.shape[0]
        st_c = torch.zeros(nn, dtype=torch.uint8, device=dev)
        copies = [cells0.clone() for _ in range(7)]
        ctx.propagate_device(nn, copies[0], None, copies[0], None, None, None, st_c, stream, cells=True)
        torch.cuda.synchronize()
        ctx.stats_reset(stream)
        ms = []
        for cpy in copies[1:]:
            ctx.propagate_device(nn, cpy, None, cpy, None, None, None, st_c, stream, cells=True)
            ms.append(ctx.last_kernel_ms())
        stc = ctx.stats_read(stream)
        st_i = torch.zeros(nn, dtype=torch.uint8, device=dev)
        ctx.propagate_device(nn, lb_h, ub_h, lb_h, ub_h, None, None, st_i, stream)
        ul, uu = ctx.unpack_rows(copies[-1], stream_ptr=stream)
        torch.cuda.synchronize()
        okc = st_i != 0
        if not (torch.equal(st_c, st_i) and torch.equal(ul[okc], lb_h[okc]) and torch.equal(uu[okc], ub_h[okc])):
            raise SystemExit("PARITY FAILURE (cells leg): the packed-cell launch differs from the int32 launch")
        med = float(np.median(ms))
        nbytes = nn * 4 * V + 4 * V * min(nn, stc["narrowings"] / len(ms))
        legs.append({"name": "C2-frontier-resident-as-packed-cells", "nodes": nn, "launches": len(ms), "kernel_ms": {"min": float(min(ms)), "median": med, "max": float(max(ms))},
                     "nodes_per_s": nn / (med * 1e-3), "compulsory_bytes_per_launch": nbytes, "hbm_frac": nbytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "identical_to_int32_launch": True,
                     "note": "the headline frontier kept in HBM as rows of 32-bit cells (-lb & 0xffff | ub << 16): half the bytes per node; hbm_frac is of THESE bytes"})
        del copies, cells0, lb_h, ub_h, ul, uu
    if "wide" in want and ctx.last_plan().get("path") == 1:
        # the headline launch at EIGHT times its width: the 8 shares of the frontier (8 x args.nodes open nodes of one depth, ~1 GB of rows) in one
        # launch — 16 tiles per persistent workgroup instead of 2.  Separates what a launch pays once (start-up, the first generation of tiles
        # staged by every CU at the same moment, the late tiles of the few nodes that narrow) from the kernel's steady state.
        reset_opts()
        ctx.set_option("nodes_per_block", args.nodes_per_block)
        stream = torch.cuda.current_stream().cuda_stream
        parts = [W.nqueens_frontier(ctx, n, args.nodes, share=s_, shares=8, implicit=True)[:2] for s_ in range(8)]
        Lw = torch.from_numpy(np.concatenate([p_[0] for p_ in parts])).to(dev)
        Uw = torch.from_numpy(np.concatenate([p_[1] for p_ in parts])).to(dev)
        nw = Lw.shape[0]
        st_w = torch.zeros(nw, dtype=torch.uint8, device=dev)
        copies = [(Lw.clone(), Uw.clone()) for _ in range(6)]
        ctx.propagate_device(nw, *copies[0], *copies[0], None, None, st_w, stream)
        torch.cuda.synchronize()
        ctx.stats_reset(stream)
        ms = []
        for l_, u_ in copies[1:]:
            ctx.propagate_device(nw, l_, u_, l_, u_, None, None, st_w, stream)
            ms.append(ctx.last_kernel_ms())
        stw = ctx.stats_read(stream)
        # the same nodes one share per launch: identical rows and statuses
        k0 = args.nodes
        l1, u1 = Lw[:k0].clone(), Uw[:k0].clone()
        st_1 = torch.zeros(k0, dtype=torch.uint8, device=dev)
        ctx.propagate_device(k0, l1, u1, l1, u1, None, None, st_1, stream)
        torch.cuda.synchronize()
        if not (torch.equal(st_1, st_w[:k0]) and torch.equal(l1, copies[-1][0][:k0]) and torch.equal(u1, copies[-1][1][:k0])):
            raise SystemExit("PARITY FAILURE (wide leg): a share of the wide launch differs from the same share launched alone")
        med = float(np.median(ms))
        nbytes = nw * node_bytes(V, words, False) + 8 * V * min(nw, stw["narrowings"] / len(ms))
        legs.append({"name": "C2-frontier-eight-shares-one-launch", "nodes": nw, "launches": len(ms), "kernel_ms": {"min": float(min(ms)), "median": med, "max": float(max(ms))},
                     "nodes_per_s": nw / (med * 1e-3), "evaluated_per_s": stw["evaluated"] / len(ms) / (med * 1e-3), "compulsory_bytes_per_launch": nbytes,
                     "hbm_frac": nbytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, "us_per_headline_batch": med * 1e3 * args.nodes / nw, "share0_identical_to_its_own_launch": True,
                     "plan": {k: v for k, v in ctx.last_plan().items() if k in ("nodes_per_block", "packed", "implicit_active", "grid")},
                     "note": "the 8 shares of the frontier (what 8 ranks would each take one of) in ONE launch on one GPU: 16 tiles per persistent workgroup; "
                             "us_per_headline_batch = this launch's time per 16384 nodes, to be read against the headline's kernel_ms"})
        del copies, Lw, Uw, parts, l1, u1
    if "explicit" in want:  # the same frontier with explicit `active` rows (round 1's node format and headline)
        reset_opts()
        ctx.set_option("nodes_per_block", args.nodes_per_block)
        shares = 8
        Le, Ue, Ae = W.nqueens_frontier(ctx, n, args.nodes, share=0 if args.share < 0 else args.share % shares, shares=shares, implicit=False)
        leg = Leg(ctx, torch, "C2-frontier-explicit-active-rows", torch.from_numpy(Le).to(dev), torch.from_numpy(Ue).to(dev), torch.from_numpy(Ae.view(np.int64)).to(dev),
                  Le.shape[0] * node_bytes(V, words, True), "the same frontier with node = domains + 183 KB `active` row (round 1's headline configuration)")
        legs.append(leg.run(launches=3, warmup=1))
        del leg, Le, Ue, Ae
    if "c2" in want:
        # C2 as surveyed: DFS with a node limit of 256, ONE pcp_propagate_device(n_nodes=1) per node (team path), branching on the device
        reset_opts()
        ds = DeviceSearch(ctx, batch=1, capacity=1024, implicit=True)
        lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
        ds.run(lb0, ub0, all_solutions=False, node_limit=32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ds.run(lb0, ub0, all_solutions=False, node_limit=256)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        legs.append({"name": "C2-dfs-256-one-node-per-call", "nodes": st.num_nodes, "seconds": dt, "us_per_node": dt / st.num_nodes * 1e6,
                     "steps_per_s": st.filter_steps / dt, "evaluated_per_s": st.evaluated / dt, "last_kernel_us": ctx.last_kernel_ms() * 1e3,
                     "plan": {k: v for k, v in ctx.last_plan().items() if k in ("nodes_per_block", "team", "packed", "implicit_active", "grid")},
                     "note": "per node: one pcp_propagate_device(n_nodes=1) + pcp_branch_device + a 16-byte D2H of the counters (the round's only sync)"})
        del ds
        # the same 256-node DFS with the stack, the branching and the stop test on the device (pcp_dfs_device): the host enqueues
        # steps and reads 8 bytes of state per 64 steps
        ctx.stats_reset()
        ctx.dfs_device(lb0, ub0, 32, capacity=1024, node_limit=32, chunk=64)
        torch.cuda.synchronize()
        ctx.stats_reset()
        t0 = time.perf_counter()
        r = ctx.dfs_device(lb0, ub0, 256, capacity=1024, node_limit=256, chunk=64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sd = ctx.stats_read()
        if r["nodes"] != st.num_nodes or r["failed"] != st.num_failed_node:
            raise SystemExit(f"pcp_dfs_device disagrees with the host-driven DFS: {r['nodes']}/{r['failed']} vs {st.num_nodes}/{st.num_failed_node}")
        legs.append({"name": "C2-dfs-256-device-side-stack", "nodes": r["nodes"], "seconds": dt, "us_per_node": dt / r["nodes"] * 1e6,
                     "steps_per_s": sd["steps"] / dt, "evaluated_per_s": sd["evaluated"] / dt,
                     "note": "pcp_dfs_device: per node a team-scratch memset + the fixpoint kernel + a 1-workgroup branch/stack kernel, no host sync inside a 64-step chunk; "
                             "bound by the kernels (the fixpoint of a node this deep in the dive), not by the enqueue rate"})
    if "mix" in want:
        # one launch whose nodes are what a SEARCH hands the engine: 16384 open nodes sampled along the reference's DFS (every 12th of its
        # first ~200 000 nodes: the dive, then the bottom of the tree) — not one depth, the depths in the proportions a search visits them
        reset_opts()
        lbm, ubm, dep = W.nqueens_dfs_samples(ctx, n, args.nodes, 12)
        reset_opts()
        assigned = (lbm == ubm).sum(dim=1).float()
        leg = Leg(ctx, torch, "MIX-nodes-along-the-dfs", lbm, ubm, None, lbm.shape[0] * node_bytes(V, words, False),
                  f"{lbm.shape[0]} open nodes sampled along the reference's depth-first search (every 12th node of its first {12 * lbm.shape[0]} nodes): stack depth "
                  f"{int(dep.min())}..{int(dep.max())} (median {int(np.median(dep))}), {assigned.mean().item():.0f} queens assigned on average ({int(assigned.min().item())}..{int(assigned.max().item())})")
        res = leg.run(launches=5, warmup=1)
        if args.cpu_budget > 0:  # parity of the TIMED launch on nodes spread over the batch (the deep ones cost the oracle ~0.3 s each)
            from oracle import oracle as orc
            pick = np.linspace(0, lbm.shape[0] - 1, 12).astype(np.int64)
            Lh, Uh = lbm[pick].cpu().numpy(), ubm[pick].cpu().numpy()
            refm = orc.OracleModel(n, props).consistency(Lh, Uh, None)
            g_lb, g_ub, g_st = leg.last_out[0][pick].cpu().numpy(), leg.last_out[1][pick].cpu().numpy(), leg.status[pick].cpu().numpy()
            ok = np.array_equal(refm[3], g_st) and all(refm[3][i] == 0 or (np.array_equal(refm[0][i], g_lb[i]) and np.array_equal(refm[1][i], g_ub[i])) for i in range(len(pick)))
            if not ok:
                raise SystemExit("PARITY FAILURE (mix leg): the timed launch differs from the oracle")
            res["parity_checked_nodes"] = int(len(pick))
        legs.append(res)
        # The same nodes one step further, as a search's batched loop really produces them: every node of the launch above is a propagated
        # row now; pcp_branch_device_hint makes the children of the Unknown ones and names, for every child, the one variable it was branched
        # on (pcp_device_batch.dirty_var).  The children are timed with their hints ("mixh": the first round is that variable's lists) and
        # without ("mixc": every node restarts from all of its assigned queens, what the mix leg above measures for the parents).
        ph_lb, ph_ub = leg.last_out
        nm = ph_lb.shape[0]
        cl = torch.empty((2 * nm, n), dtype=torch.int32, device=dev); cu = torch.empty_like(cl)
        cd = torch.full((2 * nm,), -1, dtype=torch.int32, device=dev)
        cnt = torch.zeros(5, dtype=torch.int32, device=dev)
        ctx.branch_device(nm, ph_lb, ph_ub, None, leg.status, cl, cu, None, cnt, torch.cuda.current_stream().cuda_stream, child_dirty=cd)
        kc = min(int(cnt[0].item()), args.nodes)
        if kc >= 1024:
            cl, cu, cd = cl[:kc].clone(), cu[:kc].clone(), cd[:kc].clone()
            results = {}
            for name, hint in (("MIXC-children-no-hints", None), ("MIXH-children-with-dirty-var-hints", cd)):
                lg = Leg(ctx, torch, name, cl, cu, None, kc * node_bytes(V, words, False),
                         f"{kc} children of the mix leg's propagated nodes (pcp_branch_device_hint), " + ("with" if hint is not None else "without") + " their dirty-variable hints")
                lg.dirty = hint
                results[name] = (lg, lg.run(launches=5, warmup=1))
            (lc, rc_), (lh, rh) = results["MIXC-children-no-hints"], results["MIXH-children-with-dirty-var-hints"]
            same = bool(torch.equal(lc.status, lh.status))
            okm = lc.status != 0
            same = same and bool(torch.equal(lc.last_out[0][okm], lh.last_out[0][okm])) and bool(torch.equal(lc.last_out[1][okm], lh.last_out[1][okm]))
            if not same:
                raise SystemExit("PARITY FAILURE (mixh leg): hinted and unhinted launches differ")
            rh["identical_to_unhinted_launch"] = True
            if args.cpu_budget > 0:
                from oracle import oracle as orc
                pick = np.linspace(0, kc - 1, 12).astype(np.int64)
                refm = orc.OracleModel(n, props).consistency(cl[pick].cpu().numpy(), cu[pick].cpu().numpy(), None)
                g_lb, g_ub, g_st = lh.last_out[0][pick].cpu().numpy(), lh.last_out[1][pick].cpu().numpy(), lh.status[pick].cpu().numpy()
                ok = np.array_equal(refm[3], g_st) and all(refm[3][i] == 0 or (np.array_equal(refm[0][i], g_lb[i]) and np.array_equal(refm[1][i], g_ub[i])) for i in range(len(pick)))
                if not ok:
                    raise SystemExit("PARITY FAILURE (mixh leg): the hinted launch differs from the oracle")
                rh["parity_checked_nodes"] = int(len(pick))
            legs.append(rc_); legs.append(rh)
            del lc, lh, results
        del leg, lbm, ubm, cl, cu, cd
    if "forest" in want:
        # the search loop itself on the device, interval domains: the root expanded to 2048 open nodes, one in-kernel DFS per node
        reset_opts()
        from pcp_amd.search_forest import forest_search
        lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
        capf = -(-500_000 // 2048) + 64  # (the warm-up allocates the timed search's stacks: they come out of the allocator's cache then)
        forest_search(ctx, lb0, ub0, node_limit=50_000, n_trees=2048, steps_per_launch=16, capacity=capf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fr = forest_search(ctx, lb0, ub0, node_limit=500_000, n_trees=2048, steps_per_launch=256, capacity=capf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        checked = None
        if args.cpu_budget > 0:
            # parity of the engine just timed, at this size and in its launch shape: four of the forest's own roots (the open nodes of the same
            # expansion), six nodes each, node for node against the oracle's DFS from those roots — one node per launch and six in one launch
            # (oracle/forest_check.py; raises on any difference)
            from oracle import oracle as orc
            from oracle import forest_check as FC
            from pcp_amd.search_forest import seed_roots_interval
            rl, ru, _ = seed_roots_interval(ctx, lb0, ub0, 2048)
            pick = np.linspace(0, rl.shape[0] - 1, 4).astype(np.int64)
            try:
                checked = FC.check_interval_forest(ctx, orc.OracleModel(n, props), rl[pick].cpu().numpy(), ru[pick].cpu().numpy(), K=6)
            except AssertionError as e:
                raise SystemExit(f"PARITY FAILURE (interval forest leg): {e}")
            del rl, ru
        # the same engine at the operating point of `--mode search`: 2 M nodes, 8192 trees (128-thread trees, eight to a CU)
        capw = -(-2_000_000 // 8192) + 64
        forest_search(ctx, lb0, ub0, node_limit=100_000, n_trees=8192, steps_per_launch=8, capacity=capw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fw = forest_search(ctx, lb0, ub0, node_limit=2_000_000, n_trees=8192, steps_per_launch=256, capacity=capw)
        torch.cuda.synchronize()
        dtw = time.perf_counter() - t0
        legs.append({"name": "C5-interval-forest-8192-trees", "nodes": fw["nodes"], "seconds": dtw, "us_per_node": dtw / fw["nodes"] * 1e6, "nodes_per_s": fw["nodes"] / dtw,
                     "trees": fw["trees"], "launches": fw["launches"], "error": fw["error"], "plan": {k: v for k, v in ctx.last_plan().items() if k in ("block", "grid", "lds_bytes", "packed")},
                     "note": "pcp_dfs_forest_device: first 2 000 000 nodes, 8192 trees of 128 threads (eight to a CU), 256 nodes per tree and launch; the timed region includes the expansion"})
        legs.append({"name": "C5-interval-forest", "nodes": fr["nodes"], "seconds": dt, "us_per_node": dt / fr["nodes"] * 1e6, "nodes_per_s": fr["nodes"] / dt,
                     "trees": fr["trees"], "launches": fr["launches"], "error": fr["error"], "parity_checked_nodes": checked,
                     "note": "pcp_dfs_forest_device: first 500 000 nodes, 2048 trees of 256 threads, 256 nodes per tree and launch; the timed region includes the expansion"})
    for dive in (500, 3000):
        if f"deep{dive}" in want:
            reset_opts()
            lb, ub, _ = W.nqueens_deep(ctx, n, dive, 4096, implicit=True)
            reset_opts()
            assigned = float((lb == ub).sum(dim=1).float().mean().item())
            leg = Leg(ctx, torch, f"C2-deep-dive-{dive}", lb, ub, None, lb.shape[0] * node_bytes(V, words, False),
                      f"4096 open nodes {dive} nodes down a left-first DFS dive ({assigned:.0f} queens assigned on average)")
            legs.append(leg.run(launches=5, warmup=1))
            del leg, lb, ub
    if "set" in want:
        # the same model over IntervalSet<i32> domains — the reference's default FDSpace, in which XNeqY removes interior values
        reset_opts()
        sw = (n + 63) // 64
        ctx.set_model(n, props, set_words=sw)
        ctx.set_hull(1, n)
        Bs, exp_nodes, exp_failed = W.nqueens_frontier_set(ctx, n, min(n, 1024))  # depth ~10: the first queen has just been assigned in every node
        Ns = Bs.shape[0]
        t_bits = torch.from_numpy(Bs.view(np.int64)).to(dev)
        t_lb = torch.zeros((Ns, n), dtype=torch.int32, device=dev)
        t_ub = torch.zeros_like(t_lb)
        t_st = torch.zeros(Ns, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        ms = []
        copies = [t_bits.clone() for _ in range(4)]
        ctx.propagate_device(Ns, None, None, t_lb, t_ub, None, None, t_st, stream, bits_in=copies[0], bits_out=copies[0])
        torch.cuda.synchronize()
        ctx.stats_reset(stream)
        for c_ in copies[1:]:
            ctx.propagate_device(Ns, None, None, t_lb, t_ub, None, None, t_st, stream, bits_in=c_, bits_out=c_)
            ms.append(ctx.last_kernel_ms())
        stt = ctx.stats_read(stream)
        med = float(np.median(ms))
        cbytes = Ns * (2 * n * sw * 8 + 2 * n * 4)  # the sets in and out (the set kernel always writes back), the bounds out
        out_bits = copies[1].cpu().numpy().view(np.uint64)
        removed = int((Bs != out_bits).sum())
        cpu_set = None
        if args.cpu_budget > 0:  # the oracle over IntervalSet domains on the first nodes of this batch: rate + parity of the launch
            from oracle import oracle as orc
            om = orc.OracleModel(n, props)
            t0c, k, osteps = time.perf_counter(), 0, 0
            while k < Ns and time.perf_counter() - t0c < args.cpu_budget / 4:
                r = om.consistency_set(Bs[k:k + 1], 1, None, check_dup=False)
                if int(r[4][0]) != int(t_st[k].item()) or (int(r[4][0]) != 0 and not np.array_equal(r[2][0], out_bits[k])):
                    raise SystemExit(f"PARITY FAILURE (set mode): node {k} differs from the oracle")
                osteps += r[5]["steps"]
                k += 1
            dtc = time.perf_counter() - t0c
            cpu_set = {"value": osteps / dtc, "unit": "filter-steps/s", "cores": 1, "kind": "port", "nodes_per_s": k / dtc,
                       "sample": f"first {k} nodes of this batch, {dtc:.1f} s, oracle over IntervalSet<i32> without the duplicate-subscription assert", "parity_checked_nodes": k}
        legs.append({"name": "C2-set-mode-IntervalSet-frontier", "nodes": Ns, "launches": len(ms), "kernel_ms": {"min": float(min(ms)), "median": med, "max": float(max(ms))},
                     "steps_per_launch": (stt["steps"] + stt["steps3"]) / len(ms), "evaluated_per_launch": stt["evaluated"] / len(ms), "full_evals_per_launch": stt["full_evals"] / len(ms),
                     "narrowings_per_launch": stt["narrowings"] / len(ms), "waves_per_node": stt["waves"] / len(ms) / Ns,
                     "steps_per_s": (stt["steps"] + stt["steps3"]) / len(ms) / (med * 1e-3), "evaluated_per_s": stt["evaluated"] / len(ms) / (med * 1e-3), "nodes_per_s": Ns / (med * 1e-3),
                     "compulsory_bytes_per_launch": cbytes, "hbm_frac": cbytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "status_false_true_unknown": np.bincount(t_st.cpu().numpy(), minlength=3)[:3].tolist(),
                     "plan": {k: v for k, v in ctx.last_plan().items() if k in ("nodes_per_block", "team", "packed", "word_level", "global_dom", "implicit_active", "set_mode", "grid")},
                     "cpu_baseline": cpu_set,
                     "note": f"N-queens n={n} over IntervalSet<i32> domains ({sw} u64 words per variable, 125 KB of sets per node in LDS), {Ns} open nodes of the breadth-first "
                             f"frontier of the FDSpace tree ({exp_nodes} nodes expanded, {exp_failed} failed); one workgroup per node; {removed} set words changed by the launch"})
        del t_bits, copies
        ctx.set_model(n, props)
        ctx.set_hull(1, n)
    if "setsearch" in want:
        # the reference's own workload end to end: example/src/nqueens.rs over FDSpace, entirely on the GPU
        reset_opts()
        sw = (n + 63) // 64
        ctx.set_model(n, props, set_words=sw)
        ctx.set_hull(1, n)
        lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
        # (a) one tree per workgroup, the current node in LDS, an undo trail (pcp_dfs_forest_device_set)
        from pcp_amd.search_forest import forest_search_set
        budget_f, trees = 1_000_000, 512
        forest_search_set(ctx, lb0, ub0, 1, node_limit=4 * trees, n_trees=trees, steps_per_launch=4)
        torch.cuda.synchronize()
        ctx.stats_reset()
        finfo = {}
        t0 = time.perf_counter()
        fr = forest_search_set(ctx, lb0, ub0, 1, node_limit=budget_f, n_trees=trees, steps_per_launch=2048, info=finfo)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fs = ctx.stats_read()
        checked = None
        if args.cpu_budget > 0:  # as in the interval forest leg: three of this forest's roots, five nodes each, against the oracle's FDSpace DFS
            from oracle import oracle as orc
            from oracle import forest_check as FC
            from pcp_amd.search_forest import seed_roots
            roots, _ = seed_roots(ctx, lb0, ub0, 1, trees)
            pick = np.linspace(0, roots.shape[0] - 1, 3).astype(np.int64)
            try:
                checked = FC.check_set_forest(ctx, orc.OracleModel(n, props), roots[pick].cpu().numpy().view(np.uint64).reshape(3, n, sw), 1, K=5)
            except AssertionError as e:
                raise SystemExit(f"PARITY FAILURE (set forest leg): {e}")
            del roots
        legs.append({"name": "C2-set-mode-device-search", "nodes": fr["nodes"], "seconds": dt, "us_per_node": dt / fr["nodes"] * 1e6, "nodes_per_s": fr["nodes"] / dt,
                     "parity_checked_nodes": checked,
                     "steps_per_s": (fs["steps"] + fs["steps3"]) / dt, "evaluated_per_s": fs["evaluated"] / dt, "last_kernel_us": ctx.last_kernel_ms() * 1e3,
                     "failed_nodes": fr["failed"], "solutions": fr["solutions"], "trees": fr["trees"], "seeded_nodes": fr["seeded_nodes"], "launches": fr["launches"],
                     "error": fr["error"], "trail_entries_max": finfo.get("trail_max"), "levels_max": finfo.get("levels_max"),
                     "plan": {k: v for k, v in ctx.last_plan().items() if k in ("nodes_per_block", "implicit_active", "set_mode", "grid", "lds_bytes")},
                     "note": f"N-queens n={n} over IntervalSet<i32> domains (FDSpace): first {budget_f} nodes of the tree; the root is expanded breadth-first by the batched "
                             f"search to {fr['trees']} open nodes ({fr['seeded_nodes']} nodes), then one tree per workgroup: the current node stays in LDS, "
                             "backtracking undoes a trail (what VStoreTrail does); the timed region includes the expansion"})
        # (b) the batched search of round 2 on the same workload: every node a 133 KB row in HBM, copied for each child
        budget, sb = 100_000, 1024
        ds = DeviceSearch(ctx, batch=sb, capacity=budget + 8 * sb, implicit=True)
        ds.run(lb0, ub0, all_solutions=True, node_limit=4 * sb, base=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ds.run(lb0, ub0, all_solutions=True, node_limit=budget, base=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        legs.append({"name": "C2-set-mode-batched-search", "nodes": st.num_nodes, "seconds": dt, "us_per_node": dt / st.num_nodes * 1e6, "nodes_per_s": st.num_nodes / dt,
                     "steps_per_s": st.filter_steps / dt, "evaluated_per_s": st.evaluated / dt, "last_kernel_us": ctx.last_kernel_ms() * 1e3,
                     "failed_nodes": st.num_failed_node, "solutions": st.num_solution, "rounds": st.rounds,
                     "plan": {k: v for k, v in ctx.last_plan().items() if k in ("nodes_per_block", "team", "packed", "implicit_active", "set_mode", "grid")},
                     "note": f"the same search with pcp_propagate_device + pcp_branch_device_set per round: first {budget} nodes, batch {sb} nodes per round, implicit nodes of 133 KB (sets + bounds)"})
        del ds
        ctx.set_model(n, props)
        ctx.set_hull(1, n)
    if "c3" in want:
        reset_opts()
        V3, P3, N3 = 50_000, 500_000, 4096
        p3, lb3, ub3, sol3 = W.planted_binary_csp(0xC3, V3, P3)
        L3, U3 = W.unit_narrowing_prefix(0xC3 + 1, lb3, ub3, sol3, N3)
        ctx.set_model(V3, p3)
        ctx.set_hull(0, 999)  # the variables are allocated with Interval(0, 999): 10-bit LDS cells instead of HBM-resident domains
        leg = Leg(ctx, torch, "C3-random-binary-csp-50k-vars-500k-props", torch.from_numpy(L3).to(dev), torch.from_numpy(U3).to(dev), None,
                  N3 * node_bytes(V3, ctx.words, False), "4096 nodes, planted solution, unit-narrowing prefixes; 400 KB of bounds per node, kept in LDS as 10-bit cells (declared hull [0, 999])")
        res3 = leg.run(launches=3, warmup=1)
        res3["plan_path"] = ctx.last_plan()["path"]
        if res3["plan_path"] != 2:
            raise SystemExit("config 3 leg: the launch did not take the 10-bit-cell kernel (plan.path != 2)")
        if args.cpu_budget > 0:  # parity of the TIMED launch on its first nodes (the oracle needs ~10 s for the model + ~1 s per node)
            from oracle import oracle as orc
            k3 = 8
            ref3 = orc.OracleModel(V3, p3).consistency(L3[:k3], U3[:k3], None)
            g_lb, g_ub = leg.last_out[0][:k3].cpu().numpy(), leg.last_out[1][:k3].cpu().numpy()
            g_st = leg.status[:k3].cpu().numpy()
            ok = np.array_equal(ref3[3], g_st) and all(ref3[3][i] == 0 or (np.array_equal(ref3[0][i], g_lb[i]) and np.array_equal(ref3[1][i], g_ub[i])) for i in range(k3))
            if not ok:
                raise SystemExit("PARITY FAILURE (config 3 leg): the timed launch differs from the oracle")
            res3["parity_checked_nodes"] = k3
        legs.append(res3)
        del leg
    if "c4" in want:
        reset_opts()
        p4, L4, U4, A4 = W.golomb_frontier(ctx, 4096)
        V4 = L4.shape[1]
        leg = Leg(ctx, torch, "C4-golomb-distinct-sum-network", torch.from_numpy(L4).to(dev), torch.from_numpy(U4).to(dev),
                  torch.from_numpy(A4.view(np.int64)).to(dev), L4.shape[0] * node_bytes(V4, ctx.words, True),
                  f"{L4.shape[0]} open nodes of the BinarySplit expansion, V={V4}, {len(p4)} elementary filters in {ctx.n_units} units (one Distinct of 990)")
        legs.append(leg.run(launches=5, warmup=1))
        del leg
    if "f4" in want:
        # the reified layer: Cumulative (propagators/cumulative.rs:59-114) = Booleans, equivalences over conjunctions, XEqYMulZ, Sum views —
        # formula units, formfix_kernel (plan.path 3).  Nodes: the start windows narrowed at random (a scheduler's open nodes); everything else is derived by the propagation.
        reset_opts()
        T4, N4 = 8, 4096
        vs4, cs4, Lf, Uf = W.cumulative_nodes(N4, tasks=T4, horizon=15, seed=0xF4)
        V4f = len(vs4)
        M.push_model(ctx, cs4, V4f)
        leg = Leg(ctx, torch, "F4-cumulative-reified-layer", torch.from_numpy(Lf).to(dev), torch.from_numpy(Uf).to(dev), None, N4 * node_bytes(V4f, ctx.words, False),
                  f"{N4} nodes with random start windows over Cumulative with {T4} tasks (V={V4f}, {ctx.n_units} units: formula trees of equivalences, XEqYMulZ, sums over Sum views), implicit nodes")
        res = leg.run(launches=5, warmup=1)
        res["plan_path"] = ctx.last_plan()["path"]
        if args.cpu_budget > 0:  # parity of the launch on its first nodes
            from oracle import oracle as orc
            om4 = orc.OracleModel(V4f)
            M.push_model(om4, cs4, V4f)
            k4 = 64
            ref4 = om4.consistency(Lf[:k4], Uf[:k4], None)
            got4 = ctx.propagate_implicit(Lf[:k4], Uf[:k4])
            ok = np.array_equal(ref4[3], got4[3]) and all(ref4[3][i] == 0 or (np.array_equal(ref4[0][i], got4[0][i]) and np.array_equal(ref4[1][i], got4[1][i])) for i in range(k4))
            if not ok:
                raise SystemExit("PARITY FAILURE (cumulative leg): the launch differs from the oracle")
            res["parity_checked_nodes"] = k4
        legs.append(res)
        del leg
    # back to the headline model for anything that follows
    ctx.set_model(n, props)
    ctx.set_hull(1, n)
    return legs


if __name__ == "__main__":
    main()
