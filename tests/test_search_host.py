"""Host-side search driver (pcp_amd.search) and multi-rank worklist (pcp_amd.distributed) on CPU.  The engine
is replaced by the oracle-backed stand-in of tests/oracle_ctx.py; the N>1 path runs as 2 gloo ranks."""
import json
import os
import socket

import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
from pcp_amd import search as S
from pcp_amd import distributed as D

from oracle_ctx import OracleCtx

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ALL_SOLUTIONS = [1, 0, 0, 2, 10, 4, 40, 92, 352]  # search/engine/all_solution.rs:70


def test_branching_against_reference_tables():
    g = json.load(open(os.path.join(GOLDEN, "engine_kats.json")))["search"]
    for c in g["first_smallest_var"]["cases"]:
        lb = np.array([[d[0] for d in c["vars"]]], np.int32)
        ub = np.array([[d[1] for d in c["vars"]]], np.int32)
        assert S.first_smallest_var(lb, ub)[0] == c["expect"]
    pv = g["first_smallest_var"]["panic"]["vars"]
    assert S.first_smallest_var(np.array([[d[0] for d in pv]], np.int32), np.array([[d[1] for d in pv]], np.int32))[0] == -1
    root = g["binary_split"]["root"]
    # FirstSmallestVar picks var 2 ([1,2]) on the reference's root; check each var's split by masking the others
    for c in g["binary_split"]["cases"]:
        lb = np.array([[5, 5, 5]], np.int32)
        ub = np.array([[5, 5, 5]], np.int32)
        lb[0, c["var"]], ub[0, c["var"]] = root[c["var"]]
        L, U, _ = S.branch(lb, ub, None)
        assert [[int(L[0, c["var"]]), int(U[0, c["var"]])], [int(L[1, c["var"]]), int(U[1, c["var"]])]] == c["children"]
    # MiddleVal truncates toward zero like Rust's `/`
    assert S.middle_val(np.array([-3]), np.array([-2]))[0] == -2 and orc.middle_val(-3, -2) == -2
    assert S.middle_val(np.array([-7]), np.array([2]))[0] == -2 and orc.middle_val(-7, 2) == -2


@pytest.mark.parametrize("n", range(1, 9))
def test_dfs_matches_reference_order(n):
    """batch=1 DFS == the reference's left-first DFS: same solutions, same node and failure counts."""
    props = M.nqueens_props(n) if n > 1 else M.lower_units([], 1)
    ctx = OracleCtx(n, props)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    st = S.dfs(ctx, lb0, ub0, all_solutions=True)
    ss, _, _, _ = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=True)
    assert st.num_solution == ALL_SOLUTIONS[n - 1] == ss["num_solution"]
    assert st.num_nodes == ss["num_nodes"] and st.num_failed_node == ss["num_failed_node"]
    one = S.dfs(ctx, lb0, ub0, all_solutions=False)
    ss1, _, _, sol = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=False)
    assert one.num_nodes == ss1["num_nodes"]
    if ss1["num_solution"]:
        assert np.array_equal(one.solutions[0], sol)


@pytest.mark.parametrize("batch", [2, 7, 64])
def test_batched_dfs_finds_the_same_solutions(batch):
    n = 8
    props = M.nqueens_props(n)
    ctx = OracleCtx(n, props)
    st = S.dfs(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True, batch=batch)
    assert st.num_solution == 92 and st.num_nodes == 779 and st.num_failed_node == 298
    assert len({tuple(s) for s in st.solutions}) == 92
    assert st.launches <= 779 // min(batch, 4) + 1


def test_node_limit_is_stopnode():
    n = 6
    ctx = OracleCtx(n, M.nqueens_props(n))
    st = S.dfs(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True, node_limit=10)
    assert st.num_nodes == 10  # search/stop_node.rs:82-104


def test_bfs_frontier():
    n = 8
    ctx = OracleCtx(n, M.nqueens_props(n))
    L, U, A, st = S.bfs_frontier(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), 32)
    assert L.shape[0] == 32 and (L <= U).all()
    assert len({(tuple(l), tuple(u)) for l, u in zip(L, U)}) == 32  # distinct open nodes
    # finishing the search from the frontier finds what is left of the 92 solutions
    rest = 0
    full = S.bfs_frontier(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), 10 ** 6, max_rounds=40)
    assert full[3].num_solution == 92 and full[0].shape[0] == 0


def test_plan_moves_balances():
    for lengths in ([10, 0], [0, 0, 9, 1], [5, 5, 5, 5], [100, 3, 0, 0, 0, 0, 0, 1], [1, 0, 0]):
        moves = D.plan_moves(lengths)
        cur = list(lengths)
        for s, d, k in moves:
            assert k > 0 and s != d
            cur[s] -= k
            cur[d] += k
        assert sum(cur) == sum(lengths) and max(cur) - min(cur) <= 1 and min(cur) >= 0
    assert D.plan_moves([4, 4]) == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, batch, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = OracleCtx(n, M.nqueens_props(n))
        st = D.parallel_search(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), dist, batch=batch, all_solutions=True,
                               full_active=orc.full_active)
        local = sorted(tuple(int(v) for v in s) for s in st.solutions)
        q.put((rank, st.num_nodes, st.num_solution, st.num_failed_node, st.moved, local))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [1, 16])
def test_two_rank_worklist_gloo(batch):
    """world_size 2 over gloo: the sharded search explores exactly the reference's tree (779 nodes, 298 failures,
    92 distinct solutions for n=8), work really moves between the ranks, and both ranks agree on the totals."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, 8, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1:5] == res[1][1:5]          # identical global totals on both ranks
    assert res[0][1] == 779 and res[0][2] == 92 and res[0][3] == 298
    assert res[0][4] > 0                        # nodes were exchanged
    sols = res[0][5] + res[1][5]
    assert len(sols) == 92 and len(set(sols)) == 92
    assert len(res[0][5]) > 0 and len(res[1][5]) > 0  # both ranks did real work


def _device_worker(rank, world, port, n, batch, implicit, q, node_limit=0, cells=False):
    """parallel_search_device end to end over gloo: DeviceSearch on CPU tensors with the oracle-backed stand-in context."""
    import torch
    import torch.distributed as dist
    from oracle_ctx import OracleDeviceCtx
    from pcp_amd.search_device import DeviceSearch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = OracleDeviceCtx(n, M.nqueens_props(n))
        ds = DeviceSearch(ctx, batch=batch, capacity=4096, device=torch.device("cpu"), implicit=implicit, cells=cells)
        info = {}
        tot = D.parallel_search_device(ds, np.ones(n, np.int32), np.full(n, n, np.int32), dist, all_solutions=True, rounds_per_exchange=2, info=info,
                                       node_limit=node_limit)
        q.put((rank, tot, ds.stats.num_nodes, info["exchanges"], info["moved_bytes"], info["record_bytes"], info["exchange_s"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("limit", [3, 10])
def test_two_rank_small_node_limit_stops_the_expansion_gloo(limit):
    """A node limit below the size of the replicated expansion (seed_frontier): the expansion stops AT the limit (StopNode, stop_node.rs:57-62),
    no rank explores anything after it, and the total is exactly the limit."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_device_worker, args=(r, 2, port, 8, 16, True, q, limit)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and res[0][1][0] == limit and res[0][1][4] == 0
    assert res[0][2] == limit and res[1][2] == 0  # the expansion's nodes are rank 0's to report


@pytest.mark.parametrize("implicit", [True, False])
def test_two_rank_device_search_gloo(implicit):
    """BASELINE config 5's driver (device-resident stacks, worklist balanced by all_gather + pairwise send/recv, totals by
    all_reduce) with world_size 2: exactly the reference's tree, work moves, both ranks propagate nodes."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_device_worker, args=(r, 2, port, 8, 8, implicit, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, tot0, n0, x0, mb0, rb0, _), (r1, tot1, n1, x1, mb1, rb1, _) = res
    assert tot0 == tot1 and tot0[:3] == (779, 92, 298)   # nodes, solutions, failures of the reference's tree (all_solution.rs:70)
    assert tot0[4] > 0 and n0 > 0 and n1 > 0 and n0 + n1 == 779 and x0 == x1 > 1
    assert rb0 == rb1 == 8 * 8 + (4 if implicit else 8 * ((3 * 28 + 63) // 64)) and mb0 + mb1 == tot0[4] * rb0  # only whole records moved (implicit nodes: the bounds + the 4-byte dirty-variable hint)


def test_two_rank_device_search_on_cells_gloo():
    """The same driver with the open nodes kept as rows of packed cells (DeviceSearch(cells=True): pcp_device_batch.cell_format,
    pcp_branch_device_cells): the stack has ONE row tensor per node, so a record that moves between ranks is 4 bytes per variable plus the
    hint; the tree is the reference's."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_device_worker, args=(r, 2, port, 8, 8, True, q, 0, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, tot0, n0, x0, mb0, rb0, _), (r1, tot1, n1, x1, mb1, rb1, _) = res
    assert tot0 == tot1 and tot0[:3] == (779, 92, 298)
    assert tot0[4] > 0 and n0 > 0 and n1 > 0 and n0 + n1 == 779
    assert rb0 == rb1 == 4 * 8 + 4 and mb0 + mb1 == tot0[4] * rb0


def test_device_search_on_cells_stand_in():
    """DeviceSearch(cells=True) on the oracle-backed stand-in: every launch goes through the cell entry points, solutions come back unpacked."""
    import torch
    from oracle_ctx import OracleDeviceCtx
    from pcp_amd.search_device import DeviceSearch
    n = 8
    ctx = OracleDeviceCtx(n, M.nqueens_props(n))
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    base = DeviceSearch(ctx, batch=5, capacity=4096, device=torch.device("cpu"), implicit=True).run(lb0, ub0, all_solutions=True, keep_solutions=92)
    ds = DeviceSearch(ctx, batch=5, capacity=4096, device=torch.device("cpu"), implicit=True, cells=True)
    st = ds.run(lb0, ub0, all_solutions=True, keep_solutions=92)
    assert ds.ub is None and ctx.cell_launches == st.rounds
    assert (st.num_nodes, st.num_solution, st.num_failed_node) == (779, 92, 298)
    assert len(st.solutions) == 92 and all(np.array_equal(a, b) for a, b in zip(st.solutions, base.solutions))
    with pytest.raises(ValueError):
        DeviceSearch(ctx, batch=5, device=torch.device("cpu"), implicit=False, cells=True)


@pytest.mark.parametrize("n,batch", [(8, 4), (9, 8)])
def test_four_rank_device_search_gloo(n, batch):
    """world_size 4, implicit nodes: the frontier is dealt out after a replicated expansion (seed_frontier), the subtrees are of very
    different sizes so work keeps moving; the union is exactly the reference's tree, every rank propagates nodes, and the bytes moved
    are the moved records times the record size (8 bytes per variable: nothing but the rows given away is touched)."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    world = 4
    procs = [ctxm.Process(target=_device_worker, args=(r, world, port, n, batch, True, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ss, _, _, _ = orc.OracleModel(n, M.nqueens_props(n)).search(np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True)
    tots = {r[1] for r in res}
    assert len(tots) == 1
    tot = res[0][1]
    assert tot[:3] == (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
    assert sum(r[2] for r in res) == ss["num_nodes"] and all(r[2] > 0 for r in res)
    assert tot[4] > 0 and sum(r[4] for r in res) == tot[4] * res[0][5] and res[0][5] == 8 * n + 4  # (bounds + the dirty-variable hint)
    share = max(r[6] for r in res)
    print(f"exchange seconds (max over ranks) {share:.3f} over {res[0][3]} exchanges, {tot[4]} records moved")


def _stack_worker(rank, world, port, q):
    """balance_stacks over gloo with CPU tensors: rows are tagged so that nothing is lost, duplicated or torn."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class Stack:
            pass
        V, W, cap = 5, 3, 64
        st = Stack()
        st.lb = torch.zeros((cap, V), dtype=torch.int32)
        st.ub = torch.zeros((cap, V), dtype=torch.int32)
        st.act = torch.zeros((cap, W), dtype=torch.int64)
        st.size = 23 if rank == 0 else 2
        for r in range(st.size):
            tag = 1000 * rank + r
            st.lb[r] = tag
            st.ub[r] = tag + 500000
            st.act[r] = tag + 7
        delta = D.balance_stacks(st, dist)
        tags = st.lb[: st.size, 0].tolist()
        ok = all((st.lb[i] == st.lb[i, 0]).all() and (st.ub[i] == st.lb[i, 0] + 500000).all() and (st.act[i] == st.lb[i, 0] + 7).all()
                 for i in range(st.size))
        q.put((rank, st.size, delta, tags, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_device_stack_balancing_gloo():
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_stack_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, d0, t0, ok0), (r1, s1, d1, t1, ok1) = res
    assert ok0 and ok1
    assert {s0, s1} == {12, 13} and d0 == -d1 and d0 > 0
    assert sorted(t0 + t1) == sorted(list(range(23)) + [1000, 1001])  # every row exactly once
    assert t0 == list(range(d0, 23))  # rank 0 gave away its oldest rows
    assert t1[:d0] == list(range(d0)) and t1[d0:] == [1000, 1001]  # and they sit at the bottom of rank 1's stack


def test_forest_budget_shares_add_up():
    """search_forest.share_of_budget: the ranks' shares of a node limit differ by at most one and add up to what the expansion left."""
    from pcp_amd.search_forest import share_of_budget
    for limit, seeded, world in [(1000, 255, 1), (1000, 255, 8), (1_000_003, 4095, 8), (10, 50, 4), (0, 0, 2), (7, 0, 8)]:
        shares = [share_of_budget(limit, seeded, r, world) for r in range(world)]
        assert sum(shares) == max(limit - seeded, 0)
        assert max(shares) - min(shares) <= 1


# ---------------------------------------------------------------------------------------------------------------
# The interval forest's host loop (pcp_amd.search_forest.run_forest_loop) with a stand-in for pcp_dfs_forest_device that does what the
# kernel does to the stacks (pop, propagate — here with the oracle —, count, push the left child above the right one, error 1 when a
# stack is full): refill within a rank, refill across ranks over gloo, and stacks that grow on demand.
# ---------------------------------------------------------------------------------------------------------------
class _StandInForest:
    def __init__(self, n, trees, capacity, roots, max_capacity):
        import torch
        from pcp_amd.search_forest import ForestStacks
        self.n, self.T = n, trees
        self.ctx = OracleCtx(n, M.nqueens_props(n))
        lb = torch.zeros((trees, capacity, n), dtype=torch.int32)
        ub = torch.zeros((trees, capacity, n), dtype=torch.int32)
        sp = torch.zeros(trees, dtype=torch.int32)
        for t, (l, u) in enumerate(roots):
            lb[t, 0] = torch.from_numpy(l); ub[t, 0] = torch.from_numpy(u); sp[t] = 1
        # one hint word per stack row (pcp_dfs_state.dirty): written with the children like the kernel does, checked on every pop against
        # the reference tree (`hint_map`: node -> the variable its parent was branched on) — a row that moved (steal within a rank, refill
        # across ranks, a grown stack) must still carry ITS word
        self.fs = ForestStacks(lb, ub, torch.zeros((trees, capacity), dtype=torch.uint8), sp, torch.zeros(trees, dtype=torch.int32),
                               torch.zeros((trees, 5), dtype=torch.int64), max_capacity=max_capacity,
                               dirty=torch.full((trees, capacity), -1, dtype=torch.int32))
        self.hint_map = _reference_hints(n)
        self.visited = []  # (lb, ub) of every node this rank counted, as tuples
        self.solutions = []

    def launch(self, steps):
        from pcp_amd import search as S
        fs = self.fs
        for t in range(self.T):
            for _ in range(steps):
                sp = int(fs.sp[t])
                if sp == 0 or int(fs.stop[t]):
                    break
                l, u = fs.lb[t, sp - 1].numpy().copy(), fs.ub[t, sp - 1].numpy().copy()
                pl, pu, _, st, _ = self.ctx.propagate(l[None], u[None])
                key = (tuple(int(x) for x in l), tuple(int(x) for x in u))
                assert int(fs.dirty[t, sp - 1]) == self.hint_map[key], (key, int(fs.dirty[t, sp - 1]))
                if st[0] == 0:
                    fs.counters[t, 0] += 1; fs.counters[t, 2] += 1; fs.sp[t] = sp - 1
                    self.visited.append(key)
                elif st[0] == 1:
                    fs.counters[t, 0] += 1; fs.counters[t, 1] += 1; fs.sp[t] = sp - 1
                    self.visited.append(key)
                    self.solutions.append(tuple(int(x) for x in pl[0]))
                elif sp >= fs.capacity:
                    fs.counters[t, 3] = 1; fs.stop[t] = 1  # stack full: the node stays, uncounted
                else:
                    fs.counters[t, 0] += 1
                    self.visited.append(key)
                    cl, cu, _ = S.branch(pl, pu, None)
                    import torch
                    fs.lb[t, sp - 1] = torch.from_numpy(cl[1]); fs.ub[t, sp - 1] = torch.from_numpy(cu[1])  # the right child takes the parent's row
                    fs.lb[t, sp] = torch.from_numpy(cl[0]); fs.ub[t, sp] = torch.from_numpy(cu[0])          # the left child goes on top
                    var = int(np.nonzero((cl[0] != cl[1]) | (cu[0] != cu[1]))[0][0])
                    fs.dirty[t, sp - 1] = var; fs.dirty[t, sp] = var                                          # both children differ from the parent's fixpoint in `var`
                    fs.sp[t] = sp + 1


_HINTS = {}


def _reference_hints(n):
    """node (as entered) -> the variable its parent was branched on (-1: the root), over the whole n-queens tree."""
    if n not in _HINTS:
        from pcp_amd import search as S
        ctx = OracleCtx(n, M.nqueens_props(n))
        hints, stack = {}, [(np.ones(n, np.int32), np.full(n, n, np.int32), -1)]
        while stack:
            l, u, h = stack.pop()
            hints[(tuple(int(x) for x in l), tuple(int(x) for x in u))] = h
            pl, pu, _, st, _ = ctx.propagate(l[None], u[None])
            if st[0] == 2:
                cl, cu, _ = S.branch(pl, pu, None)
                var = int(np.nonzero((cl[0] != cl[1]) | (cu[0] != cu[1]))[0][0])
                stack.append((cl[1], cu[1], var)); stack.append((cl[0], cu[0], var))
        _HINTS[n] = hints
    return _HINTS[n]


def _reference_tree(n):
    """Every node of the n-queens search tree as the (lb, ub) it is entered with, by the plain recursive definition."""
    from pcp_amd import search as S
    ctx = OracleCtx(n, M.nqueens_props(n))
    nodes, sols, stack = [], [], [(np.ones(n, np.int32), np.full(n, n, np.int32))]
    while stack:
        l, u = stack.pop()
        nodes.append((tuple(int(x) for x in l), tuple(int(x) for x in u)))
        pl, pu, _, st, _ = ctx.propagate(l[None], u[None])
        if st[0] == 1:
            sols.append(tuple(int(x) for x in pl[0]))
        elif st[0] == 2:
            cl, cu, _ = S.branch(pl, pu, None)
            stack.append((cl[1], cu[1])); stack.append((cl[0], cu[0]))
    return nodes, sols


def test_forest_loop_grows_stacks_and_refills_locally():
    """One rank, 4 trees, the whole tree below ONE root: the three empty trees are fed by the first one, the 2-row stacks are doubled
    on demand, and the forest visits exactly the reference's nodes."""
    from pcp_amd.search_forest import run_forest_loop
    n = 7
    ref_nodes, ref_sols = _reference_tree(n)
    f = _StandInForest(n, 4, 2, [(np.ones(n, np.int32), np.full(n, n, np.int32))], max_capacity=64)
    r = run_forest_loop(lambda: f.launch(3), f.fs)
    assert sorted(f.visited) == sorted(ref_nodes) and len(set(f.visited)) == len(f.visited)
    assert sorted(f.solutions) == sorted(ref_sols)
    assert int(f.fs.counters[:, 0].sum()) == len(ref_nodes) and int(f.fs.counters[:, 3].max()) == 0
    assert r["steals"] > 0 and r["grown"] > 0 and 2 < r["capacity"] <= 64
    assert (f.fs.counters[:, 0] > 0).all()  # every tree did real work
    # a ceiling that is too low is reported, not hidden
    g = _StandInForest(n, 1, 2, [(np.ones(n, np.int32), np.full(n, n, np.int32))], max_capacity=2)
    run_forest_loop(lambda: g.launch(3), g.fs)
    assert int(g.fs.counters[0, 3]) == 1


def _forest_worker(rank, world, port, n, q, max_capacity=64):
    import torch.distributed as dist
    from pcp_amd.search_forest import run_forest_loop
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ALL the work starts on rank 0 (one root); the other ranks' trees are empty: they can only work on what the refill brings
        roots = [(np.ones(n, np.int32), np.full(n, n, np.int32))] if rank == 0 else []
        f = _StandInForest(n, 3, min(4, max_capacity), roots, max_capacity=max_capacity)
        r = run_forest_loop(lambda: f.launch(2), f.fs, dist=dist)
        q.put((rank, f.visited, f.solutions, int(f.fs.counters[:, 0].sum()), r["moved_rows"], r["launches"], int(f.fs.counters[:, 3].max())))
    finally:
        dist.destroy_process_group()


def test_four_rank_forest_refill_gloo():
    """world_size 4 over gloo: the forest with cross-rank refill explores EXACTLY the reference's tree (every node once, on some rank),
    rows really move between ranks, every rank works, and all ranks run the same number of launches."""
    import torch.multiprocessing as mp
    n, world = 8, 4
    ref_nodes, ref_sols = _reference_tree(n)
    assert len(ref_nodes) == 779 and len(ref_sols) == 92
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_forest_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    visited = [v for r in res for v in r[1]]
    assert len(visited) == 779 and sorted(visited) == sorted(ref_nodes)
    assert sorted(s for r in res for s in r[2]) == sorted(ref_sols)
    assert sum(r[3] for r in res) == 779 and all(r[3] > 0 for r in res)
    assert res[0][4] > 0 and sum(r[4] for r in res) >= 3  # rank 0 gave rows away; at least one per starved rank
    assert len({r[5] for r in res}) == 1 and all(r[6] == 0 for r in res)


def test_two_rank_forest_ceiling_ends_every_rank_gloo():
    """A stack ceiling that is too low on the rank that holds the work: that rank cannot grow, and BOTH ranks leave the lockstep loop in the
    same iteration (the flag travels with the all-reduced summary) — no rank is left waiting in a collective; the error is reported."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_forest_worker, args=(r, 2, port, 7, q, 2)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({r[5] for r in res}) == 1          # the same number of launches on both ranks
    assert max(r[6] for r in res) == 1            # error 1 (stack full) is reported by the rank that hit the ceiling


def test_plan_refill():
    from pcp_amd.search_forest import plan_refill
    assert plan_refill([0, 3, 0, 2], [4, 0, 2, 0]) == [(0, 1, 3), (0, 3, 1), (2, 3, 1)]
    assert plan_refill([0, 0], [5, 5]) == []
    assert plan_refill([2, 2], [0, 0]) == []
    assert plan_refill([1, 0], [3, 3]) == [(1, 0, 1)]  # a rank with idle trees does not give
    mv = plan_refill([0, 7, 0], [2, 0, 1])
    assert sum(k for _, _, k in mv) == 3 and all(k >= 1 for _, _, k in mv)


@pytest.mark.parametrize("n", [5, 6])
def test_node_limit_on_every_node_of_the_tree(n):
    """StopNode under Monitor (stop_node.rs:57-62, 90-97): whatever node the limit falls on — a failure, a solution or an inner node —
    that node is counted as a node and as nothing else.  The host DFS against the oracle's for EVERY limit up to the size of the tree."""
    props = M.nqueens_props(n)
    ctx = OracleCtx(n, props)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    om = orc.OracleModel(n, props)
    total = om.search(lb0, ub0, all_solutions=True)[0]["num_nodes"]
    kinds = set()
    for limit in range(1, total + 1):
        ss = om.search(lb0, ub0, all_solutions=True, node_limit=limit)[0]
        st = S.dfs(ctx, lb0, ub0, all_solutions=True, node_limit=limit)
        assert (st.num_nodes, st.num_solution, st.num_failed_node) == (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"]), limit
        nxt = om.search(lb0, ub0, all_solutions=True, node_limit=limit + 1)[0]
        kinds.add((nxt["num_solution"] - ss["num_solution"], nxt["num_failed_node"] - ss["num_failed_node"]))
    assert {(1, 0), (0, 1), (0, 0)} <= kinds  # the limit fell on solutions, failures and inner nodes


def _budget_worker(rank, world, port, n, limit, q):
    """parallel_search_device with a node budget over gloo: this rank's own counters and what is left on its stack."""
    import torch
    import torch.distributed as dist
    from oracle_ctx import OracleDeviceCtx
    from pcp_amd.search_device import DeviceSearch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = OracleDeviceCtx(n, M.nqueens_props(n))
        ds = DeviceSearch(ctx, batch=4, capacity=4096, device=torch.device("cpu"), implicit=True)
        tot = D.parallel_search_device(ds, np.ones(n, np.int32), np.full(n, n, np.int32), dist, all_solutions=True, rounds_per_exchange=2, node_limit=limit)
        q.put((rank, tot, ds.stats.num_nodes, ds.stats.num_solution, ds.stats.num_failed_node, ds.size))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("limit", [60, 200, 333])
def test_two_rank_node_budget_drops_no_status_gloo(limit):
    """with several ranks a rank's share of the node budget only ends its chunk — it is not the search's StopNode limit,
    so no explored node loses its status.  Every branching adds one open node and every leaf removes one:
    open = 1 + inner - leaves, nodes = inner + leaves  =>  solutions + failures = leaves = (nodes + 1 - open) / 2, exactly."""
    import torch.multiprocessing as mp
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_budget_worker, args=(r, 2, port, 8, limit, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nodes, sols, fails = res[0][1][:3]
    assert res[0][1] == res[1][1] and nodes >= limit
    open_left = sum(r[5] for r in res)
    assert nodes == sum(r[2] for r in res) and sols == sum(r[3] for r in res) and fails == sum(r[4] for r in res)
    assert 2 * (sols + fails) == nodes + 1 - open_left, (nodes, sols, fails, open_left)


class _FakeGroup:
    """A process group of `world` ranks seen from rank 0, for the exchange POLICY alone (no communication): the other ranks always report
    `others` open nodes each and nothing else."""

    def __init__(self, world, others):
        self.world, self.others, self.calls = world, others, 0

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return 0

    class ReduceOp:
        SUM = "sum"

    def all_reduce(self, t, op=None):  # (the final totals: one rank's are the totals)
        self.final = True

    def all_gather_into_tensor(self, out, inp):
        self.calls += 1
        out.view(self.world, 3)[0] = inp
        for r in range(1, self.world):
            out.view(self.world, 3)[r, 0] = self.others
            out.view(self.world, 3)[r, 1] = 0
            out.view(self.world, 3)[r, 2] = 0


def test_exchange_is_one_collective_and_backs_off_while_nobody_starves():
    """Round 6: an exchange is ONE collective — every rank's (open nodes, solutions, nodes) — and its interval doubles, up to eight
    times `rounds_per_exchange`, while every rank holds at least two rounds' worth of open nodes; the search it drives is still the reference's tree."""
    import torch
    from oracle_ctx import OracleDeviceCtx
    from pcp_amd import distributed as D
    from pcp_amd.search_device import DeviceSearch
    n = 8
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    ctx = OracleDeviceCtx(n, M.nqueens_props(n))
    ds = DeviceSearch(ctx, batch=2, capacity=4096, device=torch.device("cpu"), implicit=True)
    grp = _FakeGroup(1, 0)
    info = {}
    tot = D.parallel_search_device(ds, lb0, ub0, grp, all_solutions=True, rounds_per_exchange=1, info=info)
    assert tot[:3] == (779, 92, 298) and tot[4] == 0
    rounds = ds.stats.rounds
    assert grp.calls == info["exchanges"]  # one collective per exchange, nothing else
    # with an exchange per round there would be `rounds` of them; the interval grew to 8 rounds wherever the stack held >= 2 rounds of nodes
    assert info["exchanges"] < rounds // 3, (info["exchanges"], rounds)
    # the policy itself, on gathered sizes: moves only when somebody is about to run short
    assert D.plan_moves([100, 3]) and D.plan_moves([100, 100]) == []
