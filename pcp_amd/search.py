"""Host-side search driver over the batched HIP engine — the *caller* side of the hot path.

The reference keeps the search tree on the host (north_star) and calls ``Space::consistency`` once per node
(search/propagation.rs:42-55).  This driver keeps that structure but hands the engine many open nodes per
call.  Branching follows the reference's default engine (search/mod.rs:45-52):

* variable  ``FirstSmallestVar``  (search/branching/first_smallest_var.rs:30-39): first index among the
  variables of minimal size > 1;
* value     ``MiddleVal``         (search/branching/middle_val.rs:25-27): (lb+ub)/2, truncating toward zero;
* split     ``BinarySplit``       (search/branching/binary_split.rs:33-60): children ``x <= v`` and ``x > v``.

A branch constraint is a var-vs-constant ``XLessY`` that narrows its one variable on its first run and is then
entailed and unlinked (x_less_y.rs:87-93, propagation/store.rs:171), so it is folded into the child's bounds
(SURVEY.md §8b "per-node propagators"); children inherit the parent's ``active`` row, exactly the cstore label
``(len, active.clone())`` of propagation/store.rs:315-317.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .model import FALSE, TRUE, UNKNOWN


def first_smallest_var(lb: np.ndarray, ub: np.ndarray) -> np.ndarray:
    """Row-wise FirstSmallestVar.  Rows where every variable is assigned return -1 (the reference panics)."""
    size = ub.astype(np.int64) - lb.astype(np.int64) + 1
    big = np.iinfo(np.int64).max
    key = np.where(size > 1, size, big)
    idx = key.argmin(axis=1)  # argmin returns the FIRST minimum, as min_by_key does
    none = key[np.arange(key.shape[0]), idx] == big
    return np.where(none, -1, idx)


def middle_val(lb: np.ndarray, ub: np.ndarray) -> np.ndarray:
    s = lb.astype(np.int64) + ub.astype(np.int64)
    return (np.sign(s) * (np.abs(s) // 2)).astype(np.int32)  # Rust `/` truncates toward zero


def branch(lb: np.ndarray, ub: np.ndarray, active: Optional[np.ndarray]):
    """BinarySplit children of each (Unknown) row, folded: returns (lb2, ub2, active2) with 2 rows per input row,
    ordered left child then right child."""
    n = lb.shape[0]
    var = first_smallest_var(lb, ub)
    if (var < 0).any():
        raise RuntimeError("Cannot select a variable in a space where all variables are assigned.")
    rows = np.arange(n)
    v = middle_val(lb[rows, var], ub[rows, var])
    L = np.repeat(lb, 2, axis=0)
    U = np.repeat(ub, 2, axis=0)
    U[2 * rows, var] = np.minimum(U[2 * rows, var], v)          # x <= v
    L[2 * rows + 1, var] = np.maximum(L[2 * rows + 1, var], v + 1)  # x > v
    A = None if active is None else np.repeat(active, 2, axis=0)
    return L, U, A


@dataclass
class SearchStats:
    num_nodes: int = 0
    num_solution: int = 0
    num_failed_node: int = 0
    launches: int = 0
    filter_steps: int = 0
    solutions: List[np.ndarray] = field(default_factory=list)


def bfs_frontier(ctx, lb0: np.ndarray, ub0: np.ndarray, n_open: int, max_rounds: int = 64, active0: Optional[np.ndarray] = None,
                 implicit: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray, SearchStats]:
    """Expand the search tree breadth-first until at least ``n_open`` open (branched, not yet propagated) nodes
    exist; returns their folded (lb, ub, active) rows, at most ``n_open`` of them, in tree order.  ``lb0/ub0`` may be
    one root or a block of open nodes (with their ``active0`` rows) to continue from.  ``implicit``: nodes are domains
    only (no `active` rows; the engine derives liveness from the domains, include/pcp_hip.h) and A is None."""
    from .engine import full_active
    st = SearchStats()
    L = np.ascontiguousarray(lb0, np.int32)
    L = L.reshape(1, -1) if L.ndim == 1 else L
    U = np.ascontiguousarray(ub0, np.int32).reshape(L.shape)
    if implicit:
        A = None
    else:
        A = full_active(L.shape[0], ctx.n_units) if active0 is None else np.ascontiguousarray(active0, np.uint64).reshape(L.shape[0], -1)
    for _ in range(max_rounds):
        if L.shape[0] >= n_open or L.shape[0] == 0:
            break
        ok = (L <= U).all(axis=1)  # a folded branch can be empty: that child is failed without a launch
        st.num_failed_node += int((~ok).sum())
        L, U, A = L[ok], U[ok], (None if A is None else A[ok])
        if L.shape[0] == 0:
            break
        lb, ub, act, status, s = ctx.propagate(L, U, A)
        st.launches += 1
        st.num_nodes += L.shape[0]
        st.filter_steps += s["steps"] + s["steps3"]
        st.num_failed_node += int((status == FALSE).sum())
        for r in np.nonzero(status == TRUE)[0]:
            st.num_solution += 1
            st.solutions.append(lb[r].copy())
        unk = status == UNKNOWN
        L, U, A = branch(lb[unk], ub[unk], None if act is None else act[unk])
    ok = (L <= U).all(axis=1)
    return L[ok][:n_open], U[ok][:n_open], (None if A is None else A[ok][:n_open]), st


def dfs(ctx, lb0: np.ndarray, ub0: np.ndarray, all_solutions: bool = False, node_limit: int = 0, batch: int = 1) -> SearchStats:
    """Depth-first search with a LIFO stack of open nodes (gcollections::VectorStack in the reference).  With
    ``batch=1`` the node order is exactly the reference's left-first DFS (one_solution.rs:46-51, 92-105); with
    ``batch>1`` the top ``batch`` open nodes are propagated in one launch (batched subtree propagation)."""
    from .engine import full_active
    st = SearchStats()
    stack: List[Tuple[np.ndarray, np.ndarray, np.ndarray]] = [
        (np.ascontiguousarray(lb0, np.int32), np.ascontiguousarray(ub0, np.int32), full_active(1, ctx.n_units)[0])
    ]
    while stack:
        take = stack[-batch:][::-1]  # top of the stack first
        del stack[-batch:]
        if node_limit:
            take = take[: max(0, node_limit - st.num_nodes)]
            if not take:
                break
        L = np.stack([t[0] for t in take])
        U = np.stack([t[1] for t in take])
        A = np.stack([t[2] for t in take])
        ok = (L <= U).all(axis=1)
        lb, ub, act, status = L.copy(), U.copy(), A.copy(), np.zeros(L.shape[0], np.uint8)
        if ok.any():
            plb, pub, pact, pst, s = ctx.propagate(L[ok], U[ok], A[ok])
            lb[ok], ub[ok], act[ok], status[ok] = plb, pub, pact, pst
            st.launches += 1
            st.filter_steps += s["steps"] + s["steps3"]
        st.num_nodes += L.shape[0]
        if node_limit and st.num_nodes >= node_limit:
            # StopNode hands EndOfSearch to the monitor for the node that reaches the limit (stop_node.rs:57-62 under Monitor,
            # stop_node.rs:90-97): it is counted as a node, never as a solution or a failure
            status = status.copy()
            status[-1] = UNKNOWN
        st.num_failed_node += int((status == FALSE).sum())
        done = False
        for r in np.nonzero(status == TRUE)[0]:
            st.num_solution += 1
            st.solutions.append(lb[r].copy())
            if not all_solutions:
                done = True
        if done or (node_limit and st.num_nodes >= node_limit):
            break
        unk = np.nonzero(status == UNKNOWN)[0]
        if len(unk):
            cl, cu, ca = branch(lb[unk], ub[unk], act[unk])
            # push so that the first taken node's left child ends on top: iterate parents in reverse, right then left
            for k in range(len(unk) - 1, -1, -1):
                stack.append((cl[2 * k + 1], cu[2 * k + 1], ca[2 * k + 1]))
                stack.append((cl[2 * k], cu[2 * k], ca[2 * k]))
    return st


# ---------------------------------------------------------------------------------------------------------------------
# Set mode (IntervalSet<i32> domains as bitsets — the reference's default FDSpace, search/mod.rs:41-43).  The same engine,
# with the selectors the reference applies to sets: FirstSmallestVar compares CARDINALITIES (first_smallest_var.rs:30-39:
# `v.size()`), MiddleVal is (lower + upper) / 2 (middle_val.rs:25-27), BinarySplit keeps the values <= v resp. > v.
# ---------------------------------------------------------------------------------------------------------------------
def _popcount64(a: np.ndarray) -> np.ndarray:
    a = a.astype(np.uint64)
    m1, m2, m4 = np.uint64(0x5555555555555555), np.uint64(0x3333333333333333), np.uint64(0x0F0F0F0F0F0F0F0F)
    a = a - ((a >> np.uint64(1)) & m1)
    a = (a & m2) + ((a >> np.uint64(2)) & m2)
    a = (a + (a >> np.uint64(4))) & m4
    return ((a * np.uint64(0x0101010101010101)) >> np.uint64(56)).astype(np.int64)


def branch_set(bits: np.ndarray, lb: np.ndarray, ub: np.ndarray, base: int, active: Optional[np.ndarray]):
    """BinarySplit children of each (Unknown) set-mode row, folded into the sets: returns (bits2, active2), 2 rows per input
    row (left `x <= v`, then right `x > v`).  bits: [n, V, set_words]; lb/ub: the sets' bounds."""
    from .model import interval_bits
    n, V, sw = bits.shape
    size = _popcount64(bits).sum(axis=2)
    big = np.iinfo(np.int64).max
    key = np.where(size > 1, size, big)
    var = key.argmin(axis=1)
    rows = np.arange(n)
    if (key[rows, var] == big).any():
        raise RuntimeError("Cannot select a variable in a space where all variables are assigned.")
    v = middle_val(lb[rows, var], ub[rows, var]).astype(np.int64)
    B = np.repeat(bits, 2, axis=0)
    lo_mask = interval_bits(np.full(n, base, np.int64), v, sw, base)                      # values <= v
    hi_mask = interval_bits(v + 1, np.full(n, base + 64 * sw - 1, np.int64), sw, base)    # values > v
    B[2 * rows, var] &= lo_mask
    B[2 * rows + 1, var] &= hi_mask
    A = None if active is None else np.repeat(active, 2, axis=0)
    return B, A


def dfs_set(ctx, lb0: np.ndarray, ub0: np.ndarray, base: int, all_solutions: bool = False, node_limit: int = 0, batch: int = 1,
            implicit: bool = True) -> SearchStats:
    """Depth-first search over set-mode nodes (FDSpace): the variables are allocated as IntervalSet::new(lb0, ub0)
    (example/src/nqueens.rs:32-35); with batch = 1 the node order is the reference's left-first DFS."""
    from .engine import full_active
    from .model import interval_bits
    st = SearchStats()
    sw = ctx.set_words
    root = interval_bits(np.asarray(lb0), np.asarray(ub0), sw, base)
    stack: List[Tuple[np.ndarray, Optional[np.ndarray]]] = [(root, None if implicit else full_active(1, ctx.n_units)[0])]
    while stack:
        take = stack[-batch:][::-1]
        del stack[-batch:]
        if node_limit:
            take = take[: max(0, node_limit - st.num_nodes)]
            if not take:
                break
        Bt = np.stack([t[0] for t in take])
        A = None if implicit else np.stack([t[1] for t in take])
        ok = Bt.any(axis=2).all(axis=1)  # a folded branch can empty a set: that child is failed without a launch
        status = np.zeros(Bt.shape[0], np.uint8)
        lb = np.ones(Bt.shape[:2], np.int32); ub = np.zeros(Bt.shape[:2], np.int32)
        act = None if A is None else A.copy()
        if ok.any():
            plb, pub, pbits, pact, pst, s = ctx.propagate_set(Bt[ok], None if A is None else A[ok])
            Bt[ok], lb[ok], ub[ok], status[ok] = pbits, plb, pub, pst
            if act is not None:
                act[ok] = pact
            st.launches += 1
            st.filter_steps += s["steps"] + s["steps3"]
        st.num_nodes += Bt.shape[0]
        at_limit = bool(node_limit and st.num_nodes >= node_limit)
        if at_limit:  # the node that reaches the limit is a node, never a solution or a failure (StopNode under Monitor, stop_node.rs:57-62, 90-97)
            status = status.copy()
            status[-1] = UNKNOWN
        st.num_failed_node += int((status == FALSE).sum())
        done = False
        for r in np.nonzero(status == TRUE)[0]:
            st.num_solution += 1
            st.solutions.append(lb[r].copy())
            if not all_solutions:
                done = True
        if done or at_limit:
            break
        unk = np.nonzero(status == UNKNOWN)[0]
        if len(unk):
            cb, ca = branch_set(Bt[unk], lb[unk], ub[unk], base, None if act is None else act[unk])
            for k in range(len(unk) - 1, -1, -1):
                stack.append((cb[2 * k + 1], None if ca is None else ca[2 * k + 1]))
                stack.append((cb[2 * k], None if ca is None else ca[2 * k]))
    return st
