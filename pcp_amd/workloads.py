"""Synthetic instances of BASELINE.json's configs (SURVEY.md §8d "concrete synthetic inputs"), shared by bench.py,
tools/ and the parity tests so that what is measured is what is checked.  Pure generators: seeded numpy, the model
mirror (pcp_amd.model) and the engine itself for expanding search frontiers; nothing here touches the oracle.

  C2  nqueens_frontier     N-queens n: this rank's share of the breadth-first frontier of the reference's search tree
      nqueens_deep         N-queens n: open nodes on top of the stack after a depth-first dive of D nodes
  C3  planted_binary_csp   50 000 Interval<i32> variables, 500 000 `x ◇ y + c` constraints, planted solution
      unit_narrowing_prefix  its 4096-node batch: node k = the root with a random prefix of variables narrowed
  C4  golomb_frontier      Golomb-ruler distinct + sum network, the BinarySplit frontier of the root
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import model as M


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


# ---------------------------------------------------------------------------------------------------------- C2 / C5
def nqueens_frontier(ctx, n: int, nodes: int, share: int = 0, shares: int = 8, implicit: bool = False):
    """Share `share` of the N-queens-n frontier (FirstSmallestVar / MiddleVal / BinarySplit, example/src/nqueens.rs:28-50
    + search/mod.rs:45-52): the root is expanded breadth-first to 8 subtrees per share; the subtrees share, share+shares,
    ... are expanded breadth-first to `nodes` open nodes.  Share r is the same node set whatever the number of GPUs.
    `ctx` must hold the N-queens model.  Returns (lb, ub, active or None) as numpy arrays, tree order."""
    from . import search as S
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    L0, U0, A0, _ = S.bfs_frontier(ctx, lb0, ub0, 8 * shares, implicit=implicit)
    if L0.shape[0] < shares:
        raise RuntimeError(f"common frontier has only {L0.shape[0]} open nodes for {shares} shares")
    share %= shares
    L, U, A, _ = S.bfs_frontier(ctx, L0[share::shares], U0[share::shares], nodes, active0=None if implicit else A0[share::shares], implicit=implicit)
    if L.shape[0] < nodes:
        raise RuntimeError(f"frontier has only {L.shape[0]} open nodes")
    return L, U, A


def nqueens_deep(ctx, n: int, dive: int, nodes: int, rounds: int = 14, implicit: bool = False):
    """`nodes` open nodes from deep in the tree: a left-first depth-first dive of `dive` nodes (the reference's order,
    one node per round), then `rounds` batched rounds at the deep end of the stack; the top `nodes` rows of the stack.
    Returns device tensors (lb, ub, act-or-None), cloned."""
    from .search_device import DeviceSearch
    ds = DeviceSearch(ctx, batch=nodes, capacity=24 * nodes, implicit=implicit)
    ds.reset(np.ones(n, np.int32), np.full(n, n, np.int32))
    ds.advance(max_rounds=dive, batch=1)
    ds.advance(max_rounds=rounds, batch=nodes)
    lb, ub, act = ds.top(nodes)
    return lb.clone(), ub.clone(), (None if act is None else act.clone())


def nqueens_dfs_samples(ctx, n: int, samples: int, stride: int, capacity: int = 8192):
    """`samples` open nodes taken ALONG the reference's depth-first search of N-queens-n (pcp_dfs_device: OneSolution order, left first):
    after every `stride` nodes the node on top of the stack — the next one the search would propagate — is copied.  The batch therefore
    holds the depths of the tree in the proportions the search visits them (a dive of a few thousand nodes, then the bottom of the
    tree), not one depth.  Returns CUDA tensors (lb, ub) [k, n] with k <= samples (the search may end) and the depth (stack pointer)
    of every sample."""
    import ctypes as C
    import torch
    from . import engine as E
    dev = torch.device("cuda", ctx.device)
    lb = torch.zeros((capacity, n), dtype=torch.int32, device=dev)
    ub = torch.zeros((capacity, n), dtype=torch.int32, device=dev)
    lb[0] = 1
    ub[0] = n
    state = torch.tensor([1, 0], dtype=torch.int32, device=dev)
    status = torch.zeros(capacity, dtype=torch.uint8, device=dev)
    counters = torch.zeros(5, dtype=torch.int64, device=dev)
    dirty = torch.full((capacity,), -1, dtype=torch.int32, device=dev)  # (a popped row restarts from the variable it was branched on)
    st = E.DfsState(lb.data_ptr(), ub.data_ptr(), capacity, state.data_ptr(), state.data_ptr() + 4, status.data_ptr(), counters.data_ptr(), None, dirty.data_ptr())
    out_lb = torch.empty((samples, n), dtype=torch.int32, device=dev)
    out_ub = torch.empty((samples, n), dtype=torch.int32, device=dev)
    depth = np.zeros(samples, np.int32)
    stream = torch.cuda.current_stream(dev).cuda_stream
    k = 0
    while k < samples:
        sp, stop = state.cpu().tolist()
        if sp == 0 or stop:
            break
        out_lb[k].copy_(lb[sp - 1])
        out_ub[k].copy_(ub[sp - 1])
        depth[k] = sp
        k += 1
        ctx._check(ctx._L.pcp_dfs_device(ctx._h, C.byref(st), stride, 0, 0, C.c_void_p(stream)))
    return out_lb[:k].clone(), out_ub[:k].clone(), depth[:k]


def nqueens_frontier_set(ctx, n: int, nodes: int, max_rounds: int = 64):
    """The N-queens-n frontier over FDSpace (IntervalSet domains, what example/src/nqueens.rs:28-50 really allocates): breadth-first
    from the root with set-mode propagation (`ctx` holds the model with set_words = ceil(n/64) and the hull [1, n]) until `nodes`
    open, implicit-active, nodes exist.  Returns their folded sets [nodes, n, set_words] (tree order) and the node / failure
    counts of the expansion."""
    from . import search as S
    sw = ctx.set_words
    B = M.interval_bits(np.ones(n, np.int64), np.full(n, n, np.int64), sw, 1)[None]
    n_nodes = n_failed = 0
    for _ in range(max_rounds):
        if B.shape[0] >= nodes or B.shape[0] == 0:
            break
        ok = B.any(axis=2).all(axis=1)
        n_failed += int((~ok).sum())
        B = B[ok]
        lb, ub, bits, _, status, _ = ctx.propagate_set(B, None)
        n_nodes += B.shape[0]
        n_failed += int((status == M.FALSE).sum())
        unk = status == M.UNKNOWN
        B, _ = S.branch_set(bits[unk], lb[unk], ub[unk], 1, None)
    ok = B.any(axis=2).all(axis=1)
    return B[ok][:nodes], n_nodes, n_failed


# ---------------------------------------------------------------------------------------------------------------- C3
def planted_binary_csp(seed, n_vars=50_000, n_props=500_000, dom=(0, 999)):
    """BASELINE config 3 generator, vectorised: binary props `x ◇ y + c`, endpoints uniform
    without self-loops, ◇ in {< 40 %, <= 20 %, != 30 %, = 10 %} lowered to LT / LT(+1) / NEQ / EQ over Addition
    views; a planted solution s satisfies every constraint (slack U[0,20] for inequalities) so propagation never
    fails and cascades are long."""
    rng = _rng(seed)
    lo, hi = dom
    sol = rng.integers(lo, hi + 1, size=n_vars)
    x = rng.integers(0, n_vars, size=n_props)
    y = rng.integers(0, n_vars - 1, size=n_props)
    y = np.where(y >= x, y + 1, y)  # no self-loops
    op = rng.choice(4, size=n_props, p=[0.4, 0.2, 0.3, 0.1])  # 0 '<', 1 '<=', 2 '!=', 3 '='
    slack = rng.integers(0, 21, size=n_props)
    sx, sy = sol[x], sol[y]
    c = np.zeros(n_props, np.int64)
    c[op == 0] = (sx - sy + 1 + slack)[op == 0]          # x < y + c
    c[op == 1] = (sx - sy + slack)[op == 1]              # x <= y + c  ==  x < y + (c+1)
    neq = op == 2
    c[neq] = rng.integers(-20, 21, size=int(neq.sum()))
    clash = neq & (sx == sy + c)
    c[clash] += 1
    c[op == 3] = (sx - sy)[op == 3]                      # x = y + c
    props = np.zeros(n_props, dtype=M.PROP_DTYPE)
    props["var"][:] = M.PCP_NOVAR
    props["group"] = np.arange(n_props)
    props["kind"] = np.select([op <= 1, op == 2, op == 3], [M.LT, M.NEQ, M.EQ])
    props["var"][:, 0] = x
    props["var"][:, 1] = y
    props["off"][:, 1] = np.where(op == 1, c + 1, c)
    lb = np.full(n_vars, lo, np.int32)
    ub = np.full(n_vars, hi, np.int32)
    return props, lb, ub, sol


def unit_narrowing_prefix(seed, lb, ub, sol, n_nodes, k=64):
    """Node k of the config-3 batch: the root with a random prefix of k variables narrowed around the planted
    solution (so the node stays consistent and differs from its neighbours)."""
    rng = _rng(seed)
    V = lb.shape[0]
    L = np.tile(lb, (n_nodes, 1)).astype(np.int32)
    U = np.tile(ub, (n_nodes, 1)).astype(np.int32)
    for n in range(n_nodes):
        vs = rng.choice(V, size=k, replace=False)
        w = rng.integers(0, 40, size=k)
        L[n, vs] = np.maximum(lb[vs], sol[vs] - w)
        U[n, vs] = np.minimum(ub[vs], sol[vs] + rng.integers(0, 40, size=k))
    return L, U


# ---------------------------------------------------------------------------------------------------------------- C4
def golomb_frontier(ctx, nodes: int = 4096, m: int = 10, length: int = 80, max_rounds: int = 24):
    """BASELINE config 4: the Golomb-style distinct + sum network (pcp_amd.model.golomb) and the open nodes of its
    BinarySplit expansion (breadth-first from the root until `nodes` are open).  `ctx` gets the model.
    Returns (props, lb, ub, active)."""
    from . import search as S
    vs, cs = M.golomb(m, length)
    V = len(vs)
    props = cs.lower(V)
    ctx.set_model(V, props)
    lb0, ub0 = vs.bounds()
    L, U, A, _ = S.bfs_frontier(ctx, lb0, ub0, nodes, max_rounds=max_rounds)
    return props, L, U, A


def cumulative_nodes(n_nodes: int = 4096, tasks: int = 8, horizon: int = 15, seed: int = 0xF4):
    """The reified layer as a batch (bench.py's F4 leg): Cumulative (propagators/cumulative.rs:59-114) over `tasks` tasks with constant durations
    and resources — Booleans, equivalences over conjunctions, XEqYMulZ, Sum views: formula units — and `n_nodes` nodes whose START windows are
    narrowed at random (a scheduler's open nodes); everything else is derived by the propagation.  Returns (vstore, cstore, lb, ub)."""
    from . import model as M
    vs, cs = M.VStore(), M.CStore()
    rng = np.random.default_rng(seed)
    starts = [vs.alloc((0, horizon)) for _ in range(tasks)]
    durs = [M.Constant(int(d)) for d in rng.integers(1, 6, size=tasks)]
    ress = [M.Constant(int(r)) for r in rng.integers(1, 4, size=tasks)]
    cap = vs.alloc((5, 5))
    M.Cumulative(starts, durs, ress, cap).join(vs, cs)
    V = len(vs)
    lb0, ub0 = vs.bounds()
    L = np.tile(lb0, (n_nodes, 1)); U = np.tile(ub0, (n_nodes, 1))
    pick = rng.random((n_nodes, V)) < 0.7
    pick[:, tasks:] = False  # only the start windows: the Booleans and intermediates are what the propagation derives
    a_ = rng.integers(lb0, ub0 + 1, size=(n_nodes, V)); b_ = rng.integers(lb0, ub0 + 1, size=(n_nodes, V))
    L = np.where(pick, np.minimum(a_, b_), L).astype(np.int32); U = np.where(pick, np.maximum(a_, b_), U).astype(np.int32)
    return vs, cs, L, U
