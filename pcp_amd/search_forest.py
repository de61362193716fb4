"""Device-resident search over FDSpace (IntervalSet<i32> domains) as a FOREST: the batched device search (search_device.DeviceSearch:
pcp_propagate_device + pcp_branch_device_set, 133 KB rows per node) only expands the root breadth-first until there are enough open
nodes; every open node then becomes the root of a tree that ONE workgroup searches depth-first with its current node in LDS and an
undo trail in HBM (pcp_dfs_forest_device_set) — what the reference's VStoreTrail does (variable/memory/trail_memory.rs:100-104),
once per CU.  The union of the expansion and the trees is exactly the reference's search tree (tests/test_set_forest.py).

Several ranks: every rank runs the same expansion (no communication) and takes the open nodes r, r + world, ...; counters are summed
by the caller (one all_reduce).  Between launches a finished tree takes the oldest open node of the tree with the most open nodes —
of its own GPU first, and (interval forest, with a process group) of another rank when its whole GPU ran dry: run_forest_loop /
refill_across_ranks."""
from __future__ import annotations

import numpy as np


def share_of_budget(node_limit: int, seeded_nodes: int, rank: int, world: int) -> int:
    """This rank's part of what is left of `node_limit` after the expansion: the shares differ by at most one and add up exactly."""
    left = max(int(node_limit) - int(seeded_nodes), 0)
    return left // world + (1 if rank < left % world else 0)


def seed_roots(ctx, lb0, ub0, base: int, want: int):
    """Breadth-first expansion of the root to at least `want` open nodes (or until the tree is exhausted).
    Returns (roots [k, V, set_words] int64 CUDA tensor, stats of the expansion)."""
    from .search_device import DeviceSearch
    ds = DeviceSearch(ctx, batch=max(want, 1), capacity=4 * max(want, 1) + 64, implicit=True)
    ds.reset(lb0, ub0, base)
    while 0 < ds.size < want:
        if ds.advance(all_solutions=True, max_rounds=1, keep_solutions=0):
            break
    ds.compact()
    k = ds.size
    roots = ds.bits[:k].clone()
    st = ds.stats
    return roots, st


def trail_bound(ctx) -> int:
    """The most entries a tree's undo trail can hold: one per value of the root's sets (see forest_search_set)."""
    return int(ctx.n_vars) * int(ctx.set_words) * 64 + 16


def forest_search_set(ctx, lb0, ub0, base: int = 0, node_limit: int = 0, n_trees: int = 512, steps_per_launch: int = 2048, rank: int = 0, world: int = 1,
                      trail_capacity: int = 0, level_capacity: int = 0, info: dict | None = None) -> dict:
    """All solutions of the space below (lb0, ub0) — or the first `node_limit` nodes of this rank's share of it.
    trail_capacity / level_capacity 0 = derived from the model (trail_bound): every trail entry takes at least one value out of one
    set and is popped before that value can come back, so a tree's trail never holds more entries than the root has values —
    n_vars * set_words * 64 at most.  (N-queens-1000: 1 024 000 entries = 16 MB per tree; a 100-variable model: 100 KB.)
    Returns dict(nodes, solutions, failed, error, seeded_nodes, trees, launches); with world > 1 the expansion's counters are
    reported by rank 0 only, so that a sum over the ranks counts every node once."""
    roots, st = seed_roots(ctx, lb0, ub0, base, n_trees * world)
    mine = roots[rank::world].contiguous()
    out = {"seeded_nodes": st.num_nodes if rank == 0 else 0, "trees": int(mine.shape[0]), "launches": 0,
           "nodes": st.num_nodes if rank == 0 else 0, "solutions": st.num_solution if rank == 0 else 0, "failed": st.num_failed_node if rank == 0 else 0,
           "error": 0}
    budget = 0
    if node_limit:
        budget = share_of_budget(node_limit, st.num_nodes, rank, world)
        if budget == 0:
            return out
    if mine.shape[0] == 0:
        return out
    bound = trail_bound(ctx)
    r = ctx.dfs_forest_set(mine, node_limit=budget, steps_per_launch=steps_per_launch, trail_capacity=trail_capacity or bound,
                           level_capacity=level_capacity or min(bound, 1 << 14),
                           want_solution=False, info=info)
    out["nodes"] += r["nodes"]; out["solutions"] += r["solutions"]; out["failed"] += r["failed"]
    out["error"] = r["error"]; out["launches"] = r["launches"]
    return out


def plan_refill(idle, donors):
    """Who sends how many open nodes to whom: ``idle[r]`` trees of rank r have nothing left, ``donors[r]`` trees of rank r have two or
    more open nodes (each can give one).  Only a rank without idle trees gives (one with both fixes itself locally first).  The plan is a
    pure function of the two gathered lists, so every rank computes the same one.  Returns [(src, dst, k)], k >= 1."""
    give = [(r, int(d)) for r, d in enumerate(donors) if d > 0 and idle[r] == 0]
    moves, gi = [], 0
    for dst, need in enumerate(idle):
        need = int(need)
        while need > 0 and gi < len(give):
            src, have = give[gi]
            k = min(need, have)
            moves.append((src, dst, k))
            need -= k
            if have == k:
                gi += 1
            else:
                give[gi] = (src, have - k)
    return moves


def refill_across_ranks(lb, ub, sp, spc, dist, info: dict | None = None, dirty=None) -> int:
    """One exchange of the interval forest: ranks whose trees ran dry receive open nodes from ranks that have trees with two or more.
    ``lb``/``ub``: [T, capacity, V] stacks, ``sp``: [T] int32 stack sizes (device or CPU tensors), ``spc``: this rank's sizes as a numpy
    array (updated in place).  A donor tree gives its BOTTOM row (its oldest open node: the subtree nearest its root), richest trees
    first; a receiving tree starts from that row.  ONE all_gather of two integers per rank, then pairwise send/recv of the rows
    (RCCL over xGMI with backend nccl; gloo with CPU tensors in the tests).  Returns rows sent (+) or received (-)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = lb.device
    idle_t = np.nonzero(spc == 0)[0]
    donor_t = [int(i) for i in np.argsort(-spc, kind="stable") if spc[i] >= 2]
    mine = torch.tensor([len(idle_t), len(donor_t)], dtype=torch.int64, device=dev)
    gathered = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(gathered, mine)
    g = torch.stack(gathered).cpu().numpy()
    moves = plan_refill(g[:, 0].tolist(), g[:, 1].tolist())
    if not moves:
        return 0
    V = lb.shape[2]
    ops, incoming, delta, gave, took = [], [], 0, 0, 0
    for src, dst, k in moves:
        if rank == src:
            trees = donor_t[gave:gave + k]
            gave += k
            idx = torch.tensor(trees, dtype=torch.int64, device=dev)
            for t in (lb, ub) if dirty is None else (lb, ub, dirty):
                ops.append(dist.P2POp(dist.isend, t[idx, 0].contiguous(), dst))
            for d in trees:  # the donor's stack moves down by one row
                n = int(spc[d])
                lb[d, 0:n - 1] = lb[d, 1:n].clone(); ub[d, 0:n - 1] = ub[d, 1:n].clone()
                if dirty is not None:
                    dirty[d, 0:n - 1] = dirty[d, 1:n].clone()
                spc[d] = n - 1
            sp[idx] -= 1
            delta += k
        elif rank == dst:
            bufs = [torch.empty((k, V), dtype=lb.dtype, device=dev) for _ in range(2)]
            if dirty is not None:
                bufs.append(torch.empty((k,), dtype=dirty.dtype, device=dev))  # the rows' hints travel with them
            for b in bufs:
                ops.append(dist.P2POp(dist.irecv, b, src))
            incoming.append((bufs, idle_t[took:took + k]))
            took += k
            delta -= k
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for bufs, trees in incoming:
        idx = torch.tensor(np.asarray(trees), dtype=torch.int64, device=dev)
        lb[idx, 0] = bufs[0]; ub[idx, 0] = bufs[1]
        if dirty is not None:
            dirty[idx, 0] = bufs[2]
        sp[idx] = 1
        spc[trees] = 1
    if info is not None:
        info["moved_rows"] = info.get("moved_rows", 0) + max(delta, 0)
        info["moved_bytes"] = info.get("moved_bytes", 0) + max(delta, 0) * V * 8
    return delta


class ForestStacks:
    """The interval forest's state: stacks ``lb``/``ub`` [T, capacity, V], ``status`` [T, capacity], sizes ``sp`` [T], flags ``stop`` [T],
    ``counters`` [T, 5] (nodes, solutions, failed, error, ..).  ``grow()`` doubles the rows per tree (a tree's rows stay where they are
    within its slice) up to ``max_capacity``; ``on_grow`` lets the owner re-point its pcp_dfs_state."""

    def __init__(self, lb, ub, status, sp, stop, counters, max_capacity: int = 0, on_grow=None, dirty=None):
        self.lb, self.ub, self.status, self.sp, self.stop, self.counters = lb, ub, status, sp, stop, counters
        self.dirty = dirty  # [T, capacity] int32 or None: per stack row the variable it was branched on (pcp_dfs_state.dirty); moves with its row
        self.max_capacity = int(max_capacity) if max_capacity else int(lb.shape[1])
        self.on_grow = on_grow
        self.grown = 0

    @property
    def capacity(self) -> int:
        return int(self.lb.shape[1])

    def grow(self) -> bool:
        import torch
        cap = self.capacity
        new = min(2 * cap, self.max_capacity)
        if new <= cap:
            return False
        T, _, V = self.lb.shape
        for name in ("lb", "ub"):
            old = getattr(self, name)
            t = torch.empty((T, new, V), dtype=old.dtype, device=old.device)
            t[:, :cap] = old
            setattr(self, name, t)
            del old
        if self.status is not None:
            t = torch.zeros((T, new), dtype=self.status.dtype, device=self.status.device)
            t[:, :cap] = self.status
            self.status = t
        if self.dirty is not None:
            t = torch.full((T, new), -1, dtype=self.dirty.dtype, device=self.dirty.device)
            t[:, :cap] = self.dirty
            self.dirty = t
        self.grown += 1
        if self.on_grow:
            self.on_grow(self)
        return True


def run_forest_loop(launch, fs: ForestStacks, stop_on_solution: bool = False, node_budget: int = 0, rebalance: bool = True,
                    max_launches: int = 1 << 30, dist=None) -> dict:
    """The host side of the interval forest (engine.Context.dfs_forest): ``launch()`` enqueues one launch of the forest kernel on the
    stacks ``fs``; between launches a tree whose stack is full gets more rows (ForestStacks.grow: memory follows the depth actually
    reached instead of being reserved for the worst case) and finished trees take work — from trees of this GPU, then (``dist``) from
    other ranks.  With ``dist`` every rank runs the same number of launches and the end of the search / node_budget are decided on
    the all-reduced summary.  Returns dict(launches, steals, moved_rows, moved_bytes, exchange_s, grown, capacity)."""
    import time
    import torch
    T = int(fs.sp.shape[0])
    launches = steals = fatal = 0
    xinfo = {"moved_rows": 0, "moved_bytes": 0}
    exchange_s = 0.0
    multi = dist is not None and dist.get_world_size() > 1
    while launches < max_launches:
        launch()
        launches += 1
        sp, stop, counters = fs.sp, fs.stop, fs.counters
        live = (sp > 0) & (stop == 0)
        err = counters[:, 3]
        full = err == 1  # stack full: the node stayed on top of its stack, uncounted (pcp_hip.h, pcp_dfs_device)
        summary = torch.stack([live.sum(), counters[:, 0].sum(), counters[:, 1].sum(), full.sum(), ((err != 0) & ~full).sum() + fatal]).to(torch.int64)
        mine = summary.cpu().tolist()  # (the launch's only synchronisation on one GPU)
        vals = mine
        if multi:
            t0 = time.perf_counter()
            dist.all_reduce(summary, op=dist.ReduceOp.SUM)
            vals = summary.cpu().tolist()
            exchange_s += time.perf_counter() - t0
        if vals[4] or (stop_on_solution and vals[2]) or (node_budget and vals[1] >= node_budget):
            break
        if mine[3]:
            if fs.grow():
                idx = torch.nonzero(full).flatten()
                fs.counters[idx, 3] = 0
                fs.stop[idx] = 0
            elif not multi:
                break  # the ceiling is reached: the trees keep error 1 (the caller reports it)
            else:
                fatal = 1  # (several ranks leave the loop TOGETHER: the flag travels with the next launch's summary)
        elif vals[0] == 0 and vals[3] == 0:
            break
        if not rebalance:
            continue
        lb, ub, sp = fs.lb, fs.ub, fs.sp
        spc = None
        if mine[0] < T:
            # Finished trees take work from the others: the BOTTOM row of a tree's stack is its oldest open node (the subtree nearest
            # its root); it moves to the finished tree's row 0 and the donor's stack shifts down by one row.  Counters stay per
            # tree, so their sum is the search's.  (With a per-tree node limit a tree that reached it must stay stopped.)
            spc = sp.cpu().numpy().copy()
            idle = [int(i) for i in np.nonzero(spc == 0)[0]]
            donors = [int(i) for i in np.argsort(-spc, kind="stable") if spc[i] >= 2][:len(idle)]
            for d, r in zip(donors, idle):
                k = int(spc[d])
                lb[r, 0] = lb[d, 0]; ub[r, 0] = ub[d, 0]
                lb[d, 0:k - 1] = lb[d, 1:k].clone(); ub[d, 0:k - 1] = ub[d, 1:k].clone()
                if fs.dirty is not None:  # a row's hint moves with it
                    fs.dirty[r, 0] = fs.dirty[d, 0]
                    fs.dirty[d, 0:k - 1] = fs.dirty[d, 1:k].clone()
                sp[d] = k - 1; sp[r] = 1
                spc[d] = k - 1; spc[r] = 1
                steals += 1
        if multi:
            t0 = time.perf_counter()
            if spc is None:
                spc = sp.cpu().numpy().copy()
            refill_across_ranks(lb, ub, sp, spc, dist, xinfo, dirty=fs.dirty)
            exchange_s += time.perf_counter() - t0
    return {"launches": launches, "steals": steals, "moved_rows": xinfo["moved_rows"], "moved_bytes": xinfo["moved_bytes"], "exchange_s": exchange_s,
            "grown": fs.grown, "capacity": fs.capacity}


def seed_roots_interval(ctx, lb0, ub0, want: int):
    """Interval mode: breadth-first expansion of the root to at least `want` open nodes.  Returns (lb rows, ub rows, stats)."""
    from .search_device import DeviceSearch
    ds = DeviceSearch(ctx, batch=max(want, 1), capacity=4 * max(want, 1) + 64, implicit=True)
    ds.reset(lb0, ub0)
    while 0 < ds.size < want:
        if ds.advance(all_solutions=True, max_rounds=1, keep_solutions=0):
            break
    ds.compact()
    k = ds.size
    return ds.lb[:k].clone(), ds.ub[:k].clone(), ds.stats


STACK_BYTES_CAP = 64 << 30  # default ceiling of the forest's stack rows per GPU (both bound arrays together); rows are allocated as trees get deep


def forest_search(ctx, lb0, ub0, node_limit: int = 0, n_trees: int = 768, steps_per_launch: int = 512, rank: int = 0, world: int = 1, capacity: int = 0,
                  dist=None, info: dict | None = None, stack_bytes: int = STACK_BYTES_CAP) -> dict:
    """Interval mode, all-XNeqY models: expansion + one in-kernel DFS per open node (pcp_dfs_forest_device).  The node limit is checked
    between launches, so `nodes` can exceed it by less than one launch; it is the exact count.  Without ``dist`` every tree also stops at
    its share of the limit; with ``dist`` (torch.distributed, world ranks) the limit is global, finished trees are refilled (also across
    ranks) and the returned counters are still this rank's.  A tree's stack holds its OPEN nodes (one right branch per level of
    its current path); the rows are allocated on demand (ForestStacks.grow) up to ``stack_bytes`` for the whole forest — a tree
    that needs more reports error 1."""
    rl, ru, st = seed_roots_interval(ctx, lb0, ub0, n_trees * world)
    ml, mu = rl[rank::world].contiguous(), ru[rank::world].contiguous()
    out = {"seeded_nodes": st.num_nodes if rank == 0 else 0, "trees": int(ml.shape[0]), "launches": 0,
           "nodes": st.num_nodes if rank == 0 else 0, "solutions": st.num_solution if rank == 0 else 0, "failed": st.num_failed_node if rank == 0 else 0, "error": 0}
    budget = share_of_budget(node_limit, st.num_nodes, rank, world) if node_limit else 0
    multi = dist is not None and world > 1
    sp0 = None
    if multi:
        # the loop below is collective (an all_reduce and an all_gather per launch): every rank enters it or none does — decided on what is
        # left of the GLOBAL budget, the same number on every rank.  A rank without a share, or without a tree, still takes part: it
        # brings one tree with an empty stack, which the cross-rank refill can hand work to.
        if (node_limit and node_limit - st.num_nodes <= 0) or rl.shape[0] == 0:  # (the expansion used the budget up / finished the tree)
            return out
        if ml.shape[0] == 0:
            ml, mu = rl[:1].clone(), ru[:1].clone()
            sp0 = [0]
    elif (node_limit and budget == 0) or ml.shape[0] == 0:
        return out
    per_tree = -(-budget // ml.shape[0]) if budget else 0
    # a tree's stack grows by at most one row per node it explores: never more than its share of the budget; it starts at 128 rows
    # (1 MB per tree and bound array at V = 1000) and is doubled when a tree fills it, up to `stack_bytes` for the whole forest
    ceiling = max(64, int(stack_bytes) // (int(ml.shape[0]) * int(ml.shape[1]) * 8))
    if per_tree and not multi:
        ceiling = min(ceiling, per_tree + 64)
    if not capacity:
        # a KNOWN bound (one rank: a tree never holds more rows than its share of the node budget) is allocated at once when it is at most
        # an eighth of the free HBM — growing to it step by step costs three allocations and two copies of a short search; without a
        # bound (no budget, or several ranks with a global budget and refill) a tree starts at 128 rows and grows on demand
        row_bytes = int(ml.shape[0]) * int(ml.shape[1]) * 8
        upfront = 8 << 30
        if ml.is_cuda:
            import torch
            upfront = max(upfront, torch.cuda.mem_get_info(ml.device)[0] // 8)
        capacity = ceiling if (per_tree and not multi and ceiling * row_bytes <= upfront) else min(128, ceiling)
    r = ctx.dfs_forest(ml, mu, node_limit_per_tree=0 if multi else per_tree, steps_per_launch=steps_per_launch, capacity=capacity, max_capacity=max(ceiling, capacity),
                       node_budget=max(node_limit - st.num_nodes, 1) if multi else budget, dist=dist if multi else None, info=info, sp0=sp0)
    out["nodes"] += r["nodes"]; out["solutions"] += r["solutions"]; out["failed"] += r["failed"]
    out["error"] = r["error"]; out["launches"] = r["launches"]
    return out
