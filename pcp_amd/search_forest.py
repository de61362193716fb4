"""Device-resident search over FDSpace (IntervalSet<i32> domains) as a FOREST: the batched device search (search_device.DeviceSearch:
pcp_propagate_device + pcp_branch_device_set, 133 KB rows per node) only expands the root breadth-first until there are enough open
nodes; every open node then becomes the root of a tree that ONE workgroup searches depth-first with its current node in LDS and an
undo trail in HBM (pcp_dfs_forest_device_set) — what the reference's VStoreTrail does (variable/memory/trail_memory.rs:100-104),
once per CU.  The union of the expansion and the trees is exactly the reference's search tree (tests/test_set_forest.py).

Several ranks: every rank runs the same expansion (no communication) and takes the open nodes r, r + world, ...; counters are summed
by the caller (one all_reduce).  Subtrees of very different sizes are not re-balanced between trees yet: a finished tree's CU idles
until the launch ends."""
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


def forest_search_set(ctx, lb0, ub0, base: int = 0, node_limit: int = 0, n_trees: int = 512, steps_per_launch: int = 2048, rank: int = 0, world: int = 1,
                      trail_capacity: int = 1 << 21, level_capacity: int = 1 << 14, info: dict | None = None) -> dict:
    """All solutions of the space below (lb0, ub0) — or the first `node_limit` nodes of this rank's share of it.
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
    r = ctx.dfs_forest_set(mine, node_limit=budget, steps_per_launch=steps_per_launch, trail_capacity=trail_capacity, level_capacity=level_capacity,
                           want_solution=False, info=info)
    out["nodes"] += r["nodes"]; out["solutions"] += r["solutions"]; out["failed"] += r["failed"]
    out["error"] = r["error"]; out["launches"] = r["launches"]
    return out


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


def forest_search(ctx, lb0, ub0, node_limit: int = 0, n_trees: int = 768, steps_per_launch: int = 512, rank: int = 0, world: int = 1, capacity: int = 0) -> dict:
    """Interval mode, all-XNeqY models: expansion + one in-kernel DFS per open node (pcp_dfs_forest_device).  The node limit is checked
    between launches (each tree may also stop at its share of it), so `nodes` can exceed it by less than one launch; it is the exact count."""
    rl, ru, st = seed_roots_interval(ctx, lb0, ub0, n_trees * world)
    ml, mu = rl[rank::world].contiguous(), ru[rank::world].contiguous()
    out = {"seeded_nodes": st.num_nodes if rank == 0 else 0, "trees": int(ml.shape[0]), "launches": 0,
           "nodes": st.num_nodes if rank == 0 else 0, "solutions": st.num_solution if rank == 0 else 0, "failed": st.num_failed_node if rank == 0 else 0, "error": 0}
    budget = share_of_budget(node_limit, st.num_nodes, rank, world) if node_limit else 0
    if (node_limit and budget == 0) or ml.shape[0] == 0:
        return out
    per_tree = -(-budget // ml.shape[0]) if budget else 0
    if not capacity:  # a tree's stack grows by one row per open node it explores: its share of the budget, or 2048 rows without one
        capacity = per_tree + 64 if per_tree else 2048
    r = ctx.dfs_forest(ml, mu, node_limit_per_tree=per_tree, steps_per_launch=steps_per_launch, capacity=capacity, node_budget=budget)
    out["nodes"] += r["nodes"]; out["solutions"] += r["solutions"]; out["failed"] += r["failed"]
    out["error"] = r["error"]; out["launches"] = r["launches"]
    return out
