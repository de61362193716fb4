"""Test infrastructure (like everything under oracle/): the device-side search forests against the oracle's DFS, NODE FOR NODE.

Used by tests/test_forest_n1000.py and by bench.py's forest legs (`parity_checked_nodes`); never by the product.  Two engines:
  * pcp_dfs_forest_device (interval domains, pcp_neq.hip DFS = true): one workgroup per tree, the open nodes of a tree in stack rows;
  * pcp_dfs_forest_device_set (IntervalSet domains, pcp_set.hip setdfs_kernel): the current node in LDS, an undo trail.
Each is checked in two launch shapes against `OracleModel.search` / `search_set` run from the same subtree roots
(search/engine/one_solution.rs:46-105: branches pushed reversed onto a LIFO => left first; stop_node.rs:47-62 is NOT applied here:
the trees run a fixed number of steps, every status counts):
  * stepwise — ONE node per launch: the node a tree is about to visit (top stack row / persisted current node) is the oracle's k-th
    visited node, bit for bit, and the status the step leaves is the oracle's;
  * burst — K nodes in ONE launch (what the bench runs: children continue in LDS from the variable branched on): the state the launch
    leaves behind — every open stack row / the persisted node and the open levels — and the per-tree counters are what the oracle's
    first K nodes imply.  Every visited node's fixpoint is in that state: an Unknown node's right child IS its fixpoint with one bound
    moved (binary_split.rs:46-57), and the next node's input is the last left child.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


def middle_val(lb: int, ub: int) -> int:
    """MiddleVal (search/branching/middle_val.rs:25-27): (lower + upper) / 2, Rust's `/` truncates toward zero."""
    s = int(lb) + int(ub)
    return -((-s) // 2) if s < 0 else s // 2


def first_smallest_var(size: np.ndarray) -> int:
    """FirstSmallestVar (first_smallest_var.rs:30-39): minimal size > 1, the first index wins."""
    cand = np.where(size > 1, size, np.iinfo(np.int64).max)
    v = int(np.argmin(cand))
    assert size[v] > 1, "Unknown node without a variable to branch on"
    return v


def simulate_interval_stack(rec, K: int):
    """The stack pcp_dfs_forest_device leaves after the oracle's first K visited nodes of one tree: rows bottom to top, and the counters
    (nodes, solutions, failed).  Also asserts that the oracle's own k-th input is the simulated top row (the records are consistent)."""
    stack = [(rec["lb_in"][0].copy(), rec["ub_in"][0].copy())]
    sols = fails = 0
    for k in range(K):
        lb_in, ub_in = stack.pop()
        assert np.array_equal(lb_in, rec["lb_in"][k]) and np.array_equal(ub_in, rec["ub_in"][k]), f"oracle record {k} is not the simulated top row"
        st = int(rec["status"][k])
        if st == 0:
            fails += 1
        elif st == 1:
            sols += 1
        else:
            lo, uo = rec["lb_out"][k], rec["ub_out"][k]
            var = first_smallest_var(uo.astype(np.int64) - lo.astype(np.int64) + 1)
            val = middle_val(lo[var], uo[var])
            rl, ru = lo.copy(), uo.copy()
            rl[var] = max(int(lo[var]), val + 1)  # the right child x > val takes the parent's row
            ll, lu = lo.copy(), uo.copy()
            lu[var] = min(int(uo[var]), val)      # the left child x <= val goes on top
            stack.append((rl, ru))
            stack.append((ll, lu))
        if not stack:
            assert k == K - 1, "the oracle's tree ended before K nodes"
    return stack, (K, sols, fails)


def check_interval_forest(ctx, om, root_lb, root_ub, K: int, capacity: int = 64, expect_block: int = 256, hints: bool = True) -> int:
    """root_lb / root_ub: [T, V] int32 numpy — subtree roots, not yet propagated.  Runs both launch shapes on fresh stacks and compares with
    the oracle's DFS from every root.  Returns the number of (tree, node) pairs compared.  Raises AssertionError on any difference."""
    import torch
    import pcp_amd.engine as E
    T, V = root_lb.shape
    dev = torch.device("cuda", ctx.device)
    recs = []
    for t in range(T):
        _, _, rec, _ = om.search(root_lb[t], root_ub[t], all_solutions=True, node_limit=K, check_dup=False, max_records=K)
        assert rec["status"].shape[0] == K, f"tree {t}: the oracle visited {rec['status'].shape[0]} < {K} nodes"
        recs.append(rec)

    def fresh():
        lb = torch.zeros((T, capacity, V), dtype=torch.int32, device=dev)
        ub = torch.zeros((T, capacity, V), dtype=torch.int32, device=dev)
        lb[:, 0] = torch.from_numpy(root_lb).to(dev); ub[:, 0] = torch.from_numpy(root_ub).to(dev)
        sp = torch.ones(T, dtype=torch.int32, device=dev)
        stop = torch.zeros(T, dtype=torch.int32, device=dev)
        status = torch.full((T, capacity), 255, dtype=torch.uint8, device=dev)
        counters = torch.zeros((T, 5), dtype=torch.int64, device=dev)
        # (hints: the stack keeps, per row, the variable it was branched on — pcp_dfs_state.dirty, what engine.dfs_forest passes — and a popped
        # row is propagated from that variable alone; the stepwise shape pops EVERY node from the stack)
        dirty = torch.full((T, capacity), -1, dtype=torch.int32, device=dev) if hints else None
        st = E.DfsState(lb.data_ptr(), ub.data_ptr(), capacity, sp.data_ptr(), stop.data_ptr(), status.data_ptr(), counters.data_ptr(), None,
                        dirty.data_ptr() if hints else None)
        st._keep = dirty
        return lb, ub, sp, stop, status, counters, st

    def launch(st, steps):
        ctx._check(ctx._L.pcp_dfs_forest_device(ctx._h, C.byref(st), T, steps, 0, 0, None))
        torch.cuda.synchronize()
        pl = ctx.last_plan()
        want = (1, 1, 0, T, expect_block if expect_block else pl["block"]) if T > 1 else (1, 1, 0, 1, pl["block"])
        assert (pl["path"], pl["nodes_per_block"], pl["packed"], pl["grid"], pl["block"]) == want, f"forest launch off the measured shape: {pl}"

    # ---- stepwise: one node per launch -------------------------------------------------------------------------------------------
    lb, ub, sp, stop, status, counters, st = fresh()
    for k in range(K):
        spc = sp.cpu().numpy()
        assert (spc >= 1).all(), f"step {k}: a tree ran out of nodes"
        tops = torch.from_numpy((spc - 1).astype(np.int64)).to(dev)
        idx = torch.arange(T, device=dev)
        in_lb, in_ub = lb[idx, tops].cpu().numpy(), ub[idx, tops].cpu().numpy()
        for t in range(T):
            assert np.array_equal(in_lb[t], recs[t]["lb_in"][k]) and np.array_equal(in_ub[t], recs[t]["ub_in"][k]), f"tree {t} node {k}: input differs from the oracle's"
        launch(st, 1)
        got = status[idx, tops].cpu().numpy()
        for t in range(T):
            assert int(got[t]) == int(recs[t]["status"][k]), f"tree {t} node {k}: status {int(got[t])} != oracle {int(recs[t]['status'][k])}"
    cn = counters.cpu().numpy()
    for t in range(T):
        want = (K, int((recs[t]["status"] == 1).sum()), int((recs[t]["status"] == 0).sum()))
        assert tuple(int(x) for x in cn[t, :3]) == want and int(cn[t, 3]) == 0, f"tree {t}: counters {cn[t].tolist()} != {want}"

    # ---- burst: K nodes in one launch (children continue in LDS) ----------------------------------------------------------------------
    lb, ub, sp, stop, status, counters, st = fresh()
    launch(st, K)
    spc, cn = sp.cpu().numpy(), counters.cpu().numpy()
    for t in range(T):
        stack, want = simulate_interval_stack(recs[t], K)
        assert int(spc[t]) == len(stack), f"tree {t}: {int(spc[t])} open nodes, the oracle's first {K} nodes leave {len(stack)}"
        assert tuple(int(x) for x in cn[t, :3]) == want and int(cn[t, 3]) == 0, f"tree {t}: counters {cn[t].tolist()} != {want}"
        g_lb, g_ub = lb[t, :len(stack)].cpu().numpy(), ub[t, :len(stack)].cpu().numpy()
        for r, (el, eu) in enumerate(stack):
            assert np.array_equal(g_lb[r], el) and np.array_equal(g_ub[r], eu), f"tree {t}: stack row {r} of {len(stack)} differs from the oracle's"
    return 2 * T * K


def _popcounts(bits: np.ndarray) -> np.ndarray:
    """[V, sw] uint64 -> cardinality per variable."""
    b = np.ascontiguousarray(bits).view(np.uint8)
    return np.unpackbits(b, axis=-1).reshape(bits.shape[0], -1).sum(axis=1).astype(np.int64)


def simulate_set_levels(rec, K: int):
    """The open levels [(var, val)] setdfs_kernel holds after the oracle's first K visited nodes of one tree (a level = a branching whose
    right child has not been taken yet) and the counters.  A failed node / a solution takes the deepest open level's right branch."""
    levels, sols, fails = [], 0, 0
    for k in range(K):
        st = int(rec["status"][k])
        if st == 2:
            var = first_smallest_var(_popcounts(rec["bits_out"][k]))
            levels.append((var, middle_val(rec["lb_out"][k][var], rec["ub_out"][k][var])))
        else:
            if st == 0:
                fails += 1
            else:
                sols += 1
            assert levels or k == K - 1, "the oracle's tree ended before K nodes"
            if levels:
                levels.pop()
    return levels, (K, sols, fails)


def check_set_forest(ctx, om, root_bits: np.ndarray, base: int, K: int) -> int:
    """root_bits: [T, V, sw] uint64 numpy — subtree roots of the FDSpace search, not yet propagated.  Needs K + 1 oracle records per tree
    (the node the forest is ABOUT to visit after K steps is compared too).  Returns the number of (tree, node) pairs compared."""
    import torch
    import pcp_amd.engine as E
    T, V, sw = root_bits.shape
    dev = torch.device("cuda", ctx.device)
    recs = []
    for t in range(T):
        _, _, rec, _ = om.search_set(None, None, sw, base, all_solutions=True, node_limit=K + 1, check_dup=False, max_records=K + 1, root_bits=root_bits[t])
        assert rec["status"].shape[0] == K + 1, f"tree {t}: the oracle visited {rec['status'].shape[0]} < {K + 1} nodes"
        recs.append(rec)
    bound = V * sw * 64 + 16

    def fresh():
        bits = torch.from_numpy(np.ascontiguousarray(root_bits).view(np.int64)).to(dev).clone()
        tree = torch.zeros((T, 4), dtype=torch.int32, device=dev)
        tree[:, 2] = -1  # PCP_DFS_FULL: a root runs the full sweep
        levels = torch.zeros((T, 1 << 12, 4), dtype=torch.int32, device=dev)
        trail = torch.zeros((T, bound, 4), dtype=torch.int32, device=dev)
        counters = torch.zeros((T, 4), dtype=torch.int64, device=dev)
        glob = torch.zeros(4, dtype=torch.int64, device=dev)
        st = E.ForestState(T, 1 << 12, bound, 0, bits.data_ptr(), tree.data_ptr(), levels.data_ptr(), trail.data_ptr(), counters.data_ptr(),
                           glob.data_ptr(), glob.data_ptr() + 8, None, None)
        return bits, tree, levels, trail, counters, glob, st

    def launch(st, steps):
        ctx._check(ctx._L.pcp_dfs_forest_device_set(ctx._h, C.byref(st), steps, 0, 0, None))
        torch.cuda.synchronize()
        pl = ctx.last_plan()
        assert (pl["set_mode"], pl["grid"], pl["block"], pl["nodes_per_block"]) == (1, T, 1024, 1), f"set forest launch off the measured shape: {pl}"

    def node_bits(bits):
        return bits.cpu().numpy().view(np.uint64).reshape(T, V, sw)

    # ---- stepwise: the persisted current node before step k is the oracle's k-th visited node; the step's status shows in the counters ----
    bits, tree, levels, trail, counters, glob, st = fresh()
    prev = np.zeros((T, 4), np.int64)
    for k in range(K):
        nb = node_bits(bits)
        for t in range(T):
            assert np.array_equal(nb[t], recs[t]["bits_in"][k]), f"tree {t} node {k}: input sets differ from the oracle's"
        launch(st, 1)
        cn = counters.cpu().numpy()
        for t in range(T):
            d = cn[t] - prev[t]
            got = 1 if d[1] else 0 if d[2] else 2
            assert int(d[0]) == 1 and int(cn[t, 3]) == 0, f"tree {t} node {k}: counters {cn[t].tolist()}"
            assert got == int(recs[t]["status"][k]), f"tree {t} node {k}: status {got} != oracle {int(recs[t]['status'][k])}"
        prev = cn.copy()
    nb = node_bits(bits)
    for t in range(T):
        assert np.array_equal(nb[t], recs[t]["bits_in"][K]), f"tree {t}: the node after {K} steps differs from the oracle's"

    # ---- burst: K nodes in one launch -------------------------------------------------------------------------------------------------
    bits, tree, levels, trail, counters, glob, st = fresh()
    launch(st, K)
    nb, cn, tr, lv = node_bits(bits), counters.cpu().numpy(), tree.cpu().numpy(), levels.cpu().numpy()
    for t in range(T):
        want_levels, want = simulate_set_levels(recs[t], K)
        assert tuple(int(x) for x in cn[t, :3]) == want and int(cn[t, 3]) == 0, f"tree {t}: counters {cn[t].tolist()} != {want}"
        assert np.array_equal(nb[t], recs[t]["bits_in"][K]), f"tree {t}: the node after a {K}-step launch differs from the oracle's"
        assert int(tr[t, 0]) == len(want_levels), f"tree {t}: {int(tr[t, 0])} open levels, the oracle's first {K} nodes leave {len(want_levels)}"
        got_levels = [(int(lv[t, i, 0]), int(lv[t, i, 1])) for i in range(len(want_levels))]
        assert got_levels == want_levels, f"tree {t}: levels {got_levels} != {want_levels}"
    return 2 * T * K
