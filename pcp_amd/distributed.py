"""Multi-GPU subtree search: the open-node worklist sharded over the ranks of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL over xGMI on MI355X, ``gloo`` in the CPU
tests).  The propagation data path has NO collective: open search nodes are independent spaces over one
immutable model, every rank holds the whole model and propagates its own nodes.  The only exchange step is
work balancing:

  X1  all_gather of the worklist lengths                      (world x 8 bytes)
  X2  pairwise send/recv of whole node records, richest -> poorest, one batch_isend_irecv — on xGMI every
      pair of GPUs has its own link, so direct pairwise transfers use all links at once where a ring would
      be bound by one
  X3  all_reduce(SUM) of counters, all_reduce(MAX) of the stop flag

A node record is what the reference's branch label holds (search/branching/branch.rs:29-49): the vstore label
= the domains (copy memory, variable/memory/copy_memory.rs:141-151) and the cstore label = the ``active``
BitSet (propagation/store.rs:315-317); the model itself (``propagators``) is shared and never moves.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import numpy as np

from .model import FALSE, TRUE, UNKNOWN
from .search import branch


@dataclass
class Worklist:
    """LIFO stack of open nodes (gcollections::VectorStack in the reference), stored as row blocks."""
    n_vars: int
    words: int
    lb: np.ndarray = None
    ub: np.ndarray = None
    act: np.ndarray = None

    def __post_init__(self):
        if self.lb is None:
            self.lb = np.zeros((0, self.n_vars), np.int32)
            self.ub = np.zeros((0, self.n_vars), np.int32)
            self.act = np.zeros((0, self.words), np.uint64)

    def __len__(self):
        return self.lb.shape[0]

    def push(self, lb, ub, act):
        self.lb = np.concatenate([self.lb, lb.reshape(-1, self.n_vars)])
        self.ub = np.concatenate([self.ub, ub.reshape(-1, self.n_vars)])
        self.act = np.concatenate([self.act, act.reshape(-1, self.words)])

    def pop(self, k: int):
        """Take the top k nodes (top of the stack first)."""
        k = min(k, len(self))
        n = len(self)
        out = (self.lb[n - k:][::-1].copy(), self.ub[n - k:][::-1].copy(), self.act[n - k:][::-1].copy())
        self.lb, self.ub, self.act = self.lb[: n - k], self.ub[: n - k], self.act[: n - k]
        return out

    def take_bottom(self, k: int):
        """Give away the k OLDEST nodes (closest to the root: the largest subtrees, the classic work-stealing end)."""
        k = min(k, len(self))
        out = (self.lb[:k].copy(), self.ub[:k].copy(), self.act[:k].copy())
        self.lb, self.ub, self.act = self.lb[k:], self.ub[k:], self.act[k:]
        return out


def plan_moves(lengths: List[int]) -> List[Tuple[int, int, int]]:
    """Deterministic balancing plan from the gathered worklist lengths: repeatedly move nodes from the richest
    to the poorest rank until every rank is within one node of the mean.  Returns (src, dst, count) triples;
    every rank computes the same plan from the same all_gather result."""
    n = len(lengths)
    cur = list(lengths)
    total = sum(cur)
    target = [total // n + (1 if r < total % n else 0) for r in range(n)]
    surplus = [[r, cur[r] - target[r]] for r in range(n) if cur[r] > target[r]]
    deficit = [[r, target[r] - cur[r]] for r in range(n) if cur[r] < target[r]]
    surplus.sort(key=lambda t: (-t[1], t[0]))
    deficit.sort(key=lambda t: (-t[1], t[0]))
    moves = []
    i = j = 0
    while i < len(surplus) and j < len(deficit):
        k = min(surplus[i][1], deficit[j][1])
        if k > 0:
            moves.append((surplus[i][0], deficit[j][0], k))
        surplus[i][1] -= k
        deficit[j][1] -= k
        if surplus[i][1] == 0:
            i += 1
        if deficit[j][1] == 0:
            j += 1
    return moves


def balance(wl: Worklist, dist, device=None) -> int:
    """X1 + X2.  Returns the number of nodes this rank sent (+) or received (-)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    moves = plan_moves(_gather_sizes(len(wl), dist, dev))
    ops, recv_bufs, delta = [], [], 0
    rec_words = 2 * wl.n_vars * 4 + wl.words * 8  # bytes per node record
    for src, dst, k in moves:
        if rank == src:
            lb, ub, act = wl.take_bottom(k)
            payload = np.concatenate([lb.view(np.uint8).reshape(k, -1), ub.view(np.uint8).reshape(k, -1), act.view(np.uint8).reshape(k, -1)], axis=1)
            t = torch.from_numpy(np.ascontiguousarray(payload)).to(dev)
            ops.append(dist.P2POp(dist.isend, t, dst))
            delta += k
        elif rank == dst:
            t = torch.empty((k, rec_words), dtype=torch.uint8, device=dev)
            ops.append(dist.P2POp(dist.irecv, t, src))
            recv_bufs.append(t)
            delta -= k
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for t in recv_bufs:
        raw = t.cpu().numpy()
        nv4 = wl.n_vars * 4
        lb = np.ascontiguousarray(raw[:, :nv4]).view(np.int32)
        ub = np.ascontiguousarray(raw[:, nv4:2 * nv4]).view(np.int32)
        act = np.ascontiguousarray(raw[:, 2 * nv4:]).view(np.uint64)
        # received nodes go to the BOTTOM: they are old, large subtrees; the local dive continues on top
        wl.lb = np.concatenate([lb, wl.lb])
        wl.ub = np.concatenate([ub, wl.ub])
        wl.act = np.concatenate([act, wl.act])
    return delta


@dataclass
class ParallelStats:
    num_nodes: int = 0
    num_solution: int = 0
    num_failed_node: int = 0
    filter_steps: int = 0
    rounds: int = 0
    moved: int = 0
    solutions: List[np.ndarray] = field(default_factory=list)


def parallel_search(ctx, lb0, ub0, dist, batch: int = 64, all_solutions: bool = True, node_limit: int = 0,
                    balance_every: int = 1, device=None, full_active: Optional[Callable] = None) -> ParallelStats:
    """Batched subtree search with the worklist sharded over ranks.  Rank 0 starts with the root; every round
    each rank propagates the top ``batch`` nodes of its stack in one launch, branches the Unknown ones
    (FirstSmallestVar/MiddleVal/BinarySplit, folded) and pushes the children; every ``balance_every`` rounds
    the stacks are balanced (X1+X2) and termination / counters are agreed on (X3).  The set of solutions and
    every node's fixpoint are schedule-independent; the exploration order is not."""
    import torch
    from .engine import full_active as _fa
    fa = full_active or _fa
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    n_vars = int(np.asarray(lb0).shape[-1])
    wl = Worklist(n_vars, ctx.words)
    if rank == 0:
        wl.push(np.ascontiguousarray(lb0, np.int32), np.ascontiguousarray(ub0, np.int32), fa(1, ctx.n_units))
    st = ParallelStats()
    while True:
        if len(wl):
            L, U, A = wl.pop(batch)
            ok = (L <= U).all(axis=1)
            st.num_nodes += L.shape[0]
            st.num_failed_node += int((~ok).sum())
            if ok.any():
                lb, ub, act, status, s = ctx.propagate(L[ok], U[ok], A[ok])
                st.filter_steps += s["steps"] + s.get("steps3", 0)
                st.num_failed_node += int((status == FALSE).sum())
                for r in np.nonzero(status == TRUE)[0]:
                    st.num_solution += 1
                    st.solutions.append(lb[r].copy())
                unk = np.nonzero(status == UNKNOWN)[0]
                if len(unk):
                    cl, cu, ca = branch(lb[unk], ub[unk], act[unk])
                    # children in reverse so that the first node's left child ends on top of the stack
                    wl.push(cl[::-1], cu[::-1], ca[::-1])
        st.rounds += 1
        if st.rounds % balance_every == 0:
            sent = balance(wl, dist, dev)
            st.moved += max(sent, 0)
            # X3: agree on termination (global open count, solutions, node budget)
            flags = torch.tensor([len(wl), st.num_solution, st.num_nodes], dtype=torch.int64, device=dev)
            dist.all_reduce(flags, op=dist.ReduceOp.SUM)
            open_total, sol_total, nodes_total = (int(x) for x in flags.tolist())
            if open_total == 0 or (not all_solutions and sol_total > 0) or (node_limit and nodes_total >= node_limit):
                break
    tot = torch.tensor([st.num_nodes, st.num_solution, st.num_failed_node, st.filter_steps, st.moved], dtype=torch.int64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    g = ParallelStats(*[int(x) for x in tot.tolist()[:4]], rounds=st.rounds, moved=int(tot[4]))
    g.solutions = st.solutions  # local solutions only
    return g


# ---------------------------------------------------------------------------------------------------------------
# Device-resident variant: the open-node stacks live in HBM (pcp_amd.search_device.DeviceSearch) and whole node records
# move GPU-to-GPU — over RCCL/xGMI with backend "nccl", over gloo with CPU tensors in the tests.  Same plan, same
# three exchange steps as above; the only host traffic is the all_gather of the stack sizes and the counters.
# ---------------------------------------------------------------------------------------------------------------
def _gather_sizes(size: int, dist, dev) -> List[int]:
    """X1: every rank's stack size — ONE all_gather and one device-to-host read."""
    import torch
    world = dist.get_world_size()
    mine = torch.tensor([size], dtype=torch.int64, device=dev)
    gathered = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(gathered, mine)
    return [int(x) for x in torch.cat(gathered).tolist()]


class _Exchange:
    """X1 and X3 in ONE collective (round 6): every rank contributes (open nodes, solutions, nodes) and gets everybody's — an all_gather of
    3 x 8 bytes per rank into buffers that live as long as the search (device tensors plus pinned host mirrors on a GPU: no allocation, no
    pageable copy), and ONE host wait.  Round 5 paid an all_gather with its device-to-host read, then an all_reduce with another, then two
    small host-to-device copies, 36 times per 2 M nodes: 18 % of the worklist engine's time on one rank with nothing to move."""

    def __init__(self, dist, dev):
        import torch
        self.dist, self.dev, self.world = dist, dev, dist.get_world_size()
        cuda = dev.type == "cuda"
        self.h_in = torch.zeros(3, dtype=torch.int64, pin_memory=cuda)
        self.h_out = torch.zeros(3 * self.world, dtype=torch.int64, pin_memory=cuda)
        self.d_in = torch.zeros(3, dtype=torch.int64, device=dev) if cuda else self.h_in
        self.d_out = torch.zeros(3 * self.world, dtype=torch.int64, device=dev) if cuda else self.h_out

    def gather(self, size: int, sols: int, nodes: int):
        import torch
        self.h_in[0], self.h_in[1], self.h_in[2] = int(size), int(sols), int(nodes)
        if self.d_in is not self.h_in:
            self.d_in.copy_(self.h_in, non_blocking=True)
        self.dist.all_gather_into_tensor(self.d_out, self.d_in)
        if self.d_out is not self.h_out:
            self.h_out.copy_(self.d_out, non_blocking=True)
            torch.cuda.current_stream(self.dev).synchronize()
        v = self.h_out.tolist()
        return v[0::3], v[1::3], v[2::3]  # per rank: open nodes, solutions, nodes


def balance_stacks(stack, dist, info: Optional[dict] = None, sizes: Optional[List[int]] = None) -> int:
    """X1 + X2 on a stack object with tensors ``lb [cap,V]``, ``ub [cap,V]``, ``act [cap,W]`` (and ``bits`` in set mode).  The open
    nodes are either rows [0, size) (a plain stack: ``size``) or DeviceSearch's segments ``segs`` = [start, length] bottom to top.
    Senders give away their OLDEST rows (bottom of the stack: closest to the root, the largest subtrees); receivers put them at the
    bottom too.  ONLY THE MOVED ROWS ARE TOUCHED: a sender packs the k rows it gives into one send buffer and advances its bottom
    segment (no shift of what stays); a receiver writes the k rows into the hole below its bottom segment (DeviceSearch starts its
    stack an eighth into the buffer for that) and shifts only when that hole is used up.  Returns rows sent (+) or received (-);
    ``info`` accumulates moved_rows / moved_bytes / record_bytes."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = stack.lb.device
    segmented = hasattr(stack, "segs")
    segs = stack.segs if segmented else ([[0, int(stack.size)]] if stack.size else [])
    size = sum(l for _, l in segs)
    moves = plan_moves(sizes if sizes is not None else _gather_sizes(size, dist, dev))  # (`sizes`: the caller has gathered them already)
    # implicit-active stacks carry no `active` rows; set-mode stacks carry the sets as well
    rows = tuple(stack._rows()) if hasattr(stack, "_rows") else tuple(t for t in (stack.lb, stack.ub, stack.act) if t is not None)
    cap = int(stack.lb.shape[0])
    rec_bytes = sum(int(np.prod(t.shape[1:])) * t.element_size() for t in rows)
    ops, incoming, delta = [], [], 0
    for src, dst, k in moves:
        if rank == src:
            # the k oldest rows: from the bottom segments, packed into one buffer per tensor
            pieces, need = [], k
            while need:
                s0, l0 = segs[0]
                take = min(need, l0)
                pieces.append((s0, take))
                need -= take
                if take == l0:
                    segs.pop(0)
                else:
                    segs[0] = [s0 + take, l0 - take]
            for t in rows:
                buf = t[pieces[0][0]:pieces[0][0] + pieces[0][1]].contiguous() if len(pieces) == 1 else torch.cat([t[a:a + n] for a, n in pieces])
                ops.append(dist.P2POp(dist.isend, buf, dst))
            delta += k
        elif rank == dst:
            bufs = [torch.empty((k,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in rows]
            for b in bufs:
                ops.append(dist.P2POp(dist.irecv, b, src))
            incoming.append(bufs)
            delta -= k
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for bufs in incoming:  # received rows go below the bottom segment
        k = bufs[0].shape[0]
        cur = sum(l for _, l in segs)
        if cur + k > cap:
            raise RuntimeError("open-node stack overflow while receiving work; raise `capacity`")
        if not segs:
            at = min(cap // 8, cap - k)
            segs.append([at, 0])
        if segs[0][0] < k:
            # the hole below the stack is used up: move the stack up once, leaving half of the free rows below it
            gap = max(k, (cap - cur) // 2)
            for t in rows:
                packed = torch.cat([t[a:a + n] for a, n in segs]) if len(segs) > 1 else t[segs[0][0]:segs[0][0] + segs[0][1]].clone()
                t[gap:gap + cur] = packed
            segs[:] = [[gap, cur]]
        s0, l0 = segs[0]
        for t, b in zip(rows, bufs):
            t[s0 - k:s0] = b
        segs[0] = [s0 - k, l0 + k]
    if info is not None:
        info["moved_rows"] = info.get("moved_rows", 0) + max(delta, 0)
        info["moved_bytes"] = info.get("moved_bytes", 0) + max(delta, 0) * rec_bytes
        info["record_bytes"] = rec_bytes
    if segmented:
        stack.segs = [sg for sg in segs if sg[1] > 0]
    else:
        # a plain stack keeps its open nodes in rows [0, size)
        pos = 0
        for s0, l0 in segs:
            if l0 and s0 != pos:
                for t in rows:
                    t[pos:pos + l0] = t[s0:s0 + l0].clone()
            pos += l0
        stack.size = pos
    return delta


def seed_frontier(search, lb0, ub0, dist, target: int, all_solutions: bool = True, base: int = 0, node_limit: int = 0) -> int:
    """"the frontier is first expanded breadth-first to >= 8*64 open nodes, then sharded".  Every rank runs the SAME
    expansion of the root (no communication: it is a few rounds of at most `batch` nodes) and keeps the open nodes r, r + world,
    r + 2 world, ...; the nodes of the expansion are counted once, on rank 0.  No GPU waits for the first exchange."""
    world, rank = dist.get_world_size(), dist.get_rank()
    search.reset(lb0, ub0, base) if getattr(search, "bits", None) is not None else search.reset(lb0, ub0)
    # (the expansion is part of the search: it stops at the node limit like everything else — StopNode, stop_node.rs:57-62)
    while 0 < search.size < target and not (node_limit and search.stats.num_nodes >= node_limit):
        if search.advance(all_solutions=all_solutions, max_rounds=1, keep_solutions=0, node_limit=node_limit):
            break
    seeded = int(search.stats.num_nodes)  # (the same number on every rank)
    if world == 1:
        return seeded
    found = (not all_solutions) and search.stats.num_solution > 0
    if rank != 0:  # the expansion's nodes, failures and solutions are rank 0's to report
        search.stats = type(search.stats)()
        search.ctx.stats_reset(search._stream())
    if found or (node_limit and seeded >= node_limit):
        search.segs = []
        return seeded
    search.compact()
    n = search.size
    import torch
    idx = torch.arange(rank, n, world, device=search.lb.device)
    for t in search._rows():
        t[:idx.numel()] = t[idx]
    search.segs = [[0, int(idx.numel())]] if idx.numel() else []
    return seeded


def parallel_search_device(search, lb0, ub0, dist, all_solutions: bool = True, node_limit: int = 0, rounds_per_exchange: int = 4, info: Optional[dict] = None,
                           base: int = 0, seed_nodes: int = 8 * 64):
    """Sharded subtree search with device-resident stacks: ``search`` is a pcp_amd.search_device.DeviceSearch of this
    rank's GPU.  The root is expanded to a frontier that is dealt out over the ranks (seed_frontier); every
    ``rounds_per_exchange`` rounds the stacks are balanced GPU-to-GPU (X1+X2) and termination / totals agreed on (X3).
    Returns the global (nodes, solutions, failures, filter steps, moved records)."""
    import time
    import torch
    dev = search.lb.device
    world = dist.get_world_size()
    seed_frontier(search, lb0, ub0, dist, min(seed_nodes, search.batch * world) if world > 1 else 0, all_solutions=all_solutions, base=base, node_limit=node_limit)
    moved = 0
    exchange_s, exchanges = 0.0, 0
    xinfo = {}
    xch = _Exchange(dist, dev)
    # The exchange step is off the rounds' path where it has nothing to do: its interval starts at `rounds_per_exchange` and DOUBLES (up to 8 x)
    # whenever an exchange found every rank with at least a round's worth of open nodes and moved nothing; it falls back as soon as a rank runs
    # short or records move.  Every rank derives the interval from the same gathered numbers, so all of them enter the next collective together.
    interval = max(1, int(rounds_per_exchange))
    while True:
        if search.size > 0 and (all_solutions or search.stats.num_solution == 0):
            # a rank's share of what is left of the node budget (the budget is global; it is checked at every exchange)
            # One rank: the budget IS the search's StopNode limit (its limit node is counted as a node and as nothing else, stop_node.rs:57-62).
            # Several ranks: the limit is checked on the gathered total at every exchange, so there is no single "node that reaches it";
            # a rank's share only ends its chunk (stop_at) and every node it explored counts with its status.
            if world == 1:
                search.advance(all_solutions=all_solutions, max_rounds=interval, keep_solutions=0, node_limit=node_limit)
            else:
                left = max(1, (node_limit - search.stats.num_nodes * world) // world) if node_limit else 0
                search.advance(all_solutions=all_solutions, max_rounds=interval, keep_solutions=0,
                               stop_at=(search.stats.num_nodes + left) if node_limit else 0)
        t0 = time.perf_counter()
        st = search.stats
        sizes, sols, nodes = xch.gather(search.size, st.num_solution, st.num_nodes)  # X1 + X3: one collective, one host wait
        open_total, sol_total, nodes_total = sum(sizes), sum(sols), sum(nodes)
        over = open_total == 0 or (not all_solutions and sol_total > 0) or (node_limit and nodes_total >= node_limit)
        sent = 0
        # X2, pairwise — only when some rank is about to run short (fewer than two rounds' worth of open nodes): while everybody has work for the
        # rounds ahead, moving records buys nothing (records moved now are records the receiver would not have touched before the next exchange)
        need = min(sizes) < 2 * search.batch
        if not over and need:
            sent = balance_stacks(search, dist, xinfo, sizes=sizes)
            moved += max(sent, 0)
        exchange_s += time.perf_counter() - t0
        exchanges += 1
        if over:
            break
        quiet = not need
        interval = min(8 * max(1, int(rounds_per_exchange)), 2 * interval) if quiet else max(1, int(rounds_per_exchange))
    if info is not None:
        info.update(exchange_s=exchange_s, exchanges=exchanges, moved_bytes=xinfo.get("moved_bytes", 0), record_bytes=xinfo.get("record_bytes", 0))
    st = search.stats
    tot = torch.tensor([st.num_nodes, st.num_solution, st.num_failed_node, st.filter_steps, moved, getattr(st, "evaluated", 0)],
                       dtype=torch.int64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    vals = [int(x) for x in tot.tolist()]
    if info is not None:
        info["evaluated"] = vals[5]  # (propagator, node) pairs tested one by one, all ranks (pcp_stats.evaluated)
    return tuple(vals[:5])
