"""The BENCHMARKED launch, pinned to the oracle at its own size: N-queens n=1000, declared
hull [1,1000], 16-node packed tiles (16-bit LDS cells), word-group sweep with the level -1 range test, in place —
exactly what bench.py times — on nodes of the bench frontier and on nodes deep in a depth-first dive, bit-exact
against `OracleModel.consistency` (propagation/store.rs:247-257).  The launch geometry is asserted
through pcp_last_plan so that a policy change cannot silently move the test off the measured path."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
from pcp_amd import workloads as W
import pcp_amd.engine as E

from util import assert_parity

pytestmark = pytest.mark.gpu
N = 1000


@pytest.fixture(scope="module")
def env():
    import torch
    ctx = E.Context(0)
    props = M.nqueens_props(N)
    ctx.set_model(N, props)
    ctx.set_hull(1, N)
    om = orc.OracleModel(N, props)
    yield ctx, om, torch
    ctx.close()


def _headline_opts(ctx):
    for k, v in {"force_path": 0, "nodes_per_block": 16, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1}.items():
        ctx.set_option(k, v)


def _assert_headline_plan(ctx, implicit, neq_path=False):
    pl = ctx.last_plan()
    if neq_path:  # what bench.py's headline runs since round 3: the assignment-driven kernel, 16-node tiles of 16-bit cells
        assert (pl["path"], pl["nodes_per_block"], pl["packed"], pl["team"], pl["implicit_active"]) == (1, 16, 1, 1, 1), pl
        return
    assert (pl["nodes_per_block"], pl["packed"], pl["team"], pl["global_dom"], pl["compact"], pl["block"]) == (16, 1, 1, 0, 1, 1024), pl
    assert pl["word_level"] >= 1 and pl["implicit_active"] == int(implicit) and pl["path"] == 0, pl


def _launch_in_place(ctx, torch, L, U, A):
    """One pcp_propagate_device launch over the whole batch, in place, as bench.py's step does."""
    dev = torch.device("cuda", 0)
    t_lb, t_ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    t_act = None if A is None else torch.from_numpy(A.view(np.int64)).to(dev)
    t_st = torch.zeros(L.shape[0], dtype=torch.uint8, device=dev)
    ctx.stats_reset()
    ctx.propagate_device(L.shape[0], t_lb, t_ub, t_lb, t_ub, t_act, t_act, t_st)
    torch.cuda.synchronize()
    st = ctx.stats_read()
    act = None if t_act is None else t_act.cpu().numpy().view(np.uint64)
    return t_lb.cpu().numpy(), t_ub.cpu().numpy(), act, t_st.cpu().numpy(), st


@pytest.mark.parametrize("implicit,neq", [(False, False), (True, False), (True, True)])
def test_bench_frontier_nodes(env, implicit, neq):
    """The bench batch itself (share 0, 16384 open nodes, one launch, in place); 64 of its nodes — spread over the
    batch so that every region of the tree and 64 different tiles are sampled — against the oracle.  neq: the assignment-driven
    kernel (bench.py's headline), else the generic kernels on the same batch."""
    ctx, om, torch = env
    _headline_opts(ctx)
    ctx.set_option("neq_path", int(neq))
    nodes = 16384
    L, U, A = W.nqueens_frontier(ctx, N, nodes, share=0, shares=8, implicit=implicit)
    _headline_opts(ctx)
    if neq:
        ctx.set_option("nodes_per_block", 0)  # bench.py's default policy
    glb, gub, gact, gst, st = _launch_in_place(ctx, torch, L, U, A)
    _assert_headline_plan(ctx, implicit, neq)
    ctx.set_option("neq_path", 1)
    assert st["nodes"] == nodes and st["steps"] >= nodes * (om.n_units // 2)
    assert 0 < st["evaluated"] < st["steps"] and st["full_evals"] <= st["evaluated"]
    pick = np.arange(0, nodes, nodes // 64)[:64] + (np.arange(64) % 16)  # every tile position 0..15 appears
    ref = om.consistency(L[pick], U[pick], None if A is None else A[pick], check_dup=False)
    got_act = gact[pick] if gact is not None else None
    assert_parity((ref[0], ref[1], ref[2] if got_act is not None else None, ref[3]), (glb[pick], gub[pick], got_act, gst[pick]), f"bench frontier implicit={implicit}")


@pytest.mark.parametrize("dive", [500, 3000])
@pytest.mark.parametrize("implicit,neq", [(False, False), (True, False), (True, True)])
def test_deep_dive_nodes(env, dive, implicit, neq):
    """4096 open nodes from `dive` nodes down a left-first DFS (37 / 170 queens assigned: the deep-tile switch, long
    wake-up cascades), launched as one batch on the headline geometry; 64 of them against the oracle."""
    ctx, om, torch = env
    _headline_opts(ctx)
    ctx.set_option("nodes_per_block", 0)
    lb, ub, act = W.nqueens_deep(ctx, N, dive, 4096, implicit=implicit)
    L, U = lb.cpu().numpy(), ub.cpu().numpy()
    A = None if act is None else act.cpu().numpy().view(np.uint64)
    assert L.shape[0] >= 1024
    _headline_opts(ctx)
    ctx.set_option("neq_path", int(neq))
    glb, gub, gact, gst, st = _launch_in_place(ctx, torch, L, U, A)
    _assert_headline_plan(ctx, implicit, neq)
    ctx.set_option("neq_path", 1)
    n = L.shape[0]
    pick = (np.arange(64) * (n // 64) + (np.arange(64) % 16)) % n
    ref = om.consistency(L[pick], U[pick], None if A is None else A[pick], check_dup=False)
    got_act = gact[pick] if gact is not None else None
    assert_parity((ref[0], ref[1], ref[2] if got_act is not None else None, ref[3]), (glb[pick], gub[pick], got_act, gst[pick]), f"dive {dive} implicit={implicit}")
    ctx.set_option("nodes_per_block", 0)


def test_device_side_dfs_at_n1000_node_for_node(env):
    """pcp_dfs_device at the benchmarked size (parity corner b): the first 64 nodes of the reference's DFS on
    N-queens-1000, one step per call — the node on top of the device stack before step k is the oracle's k-th visited node (its
    folded input domains, bit for bit), and the status the step leaves is the oracle's.  A node's input is its parent's
    propagated domains with one bound moved, so this pins the fixpoints and the order (one_solution.rs:46-51: left first)."""
    import ctypes as C
    ctx, om, torch = env
    _headline_opts(ctx)
    ctx.set_option("nodes_per_block", 0)
    K = 64
    lb0, ub0 = np.ones(N, np.int32), np.full(N, N, np.int32)
    _, _, rec, _ = om.search(lb0, ub0, all_solutions=False, node_limit=K, check_dup=False, max_records=K)
    dev = torch.device("cuda", 0)
    cap = 256
    lb = torch.zeros((cap, N), dtype=torch.int32, device=dev)
    ub = torch.zeros((cap, N), dtype=torch.int32, device=dev)
    lb[0] = torch.from_numpy(lb0).to(dev); ub[0] = torch.from_numpy(ub0).to(dev)
    state = torch.tensor([1, 0], dtype=torch.int32, device=dev)
    status = torch.zeros(cap, dtype=torch.uint8, device=dev)
    counters = torch.zeros(5, dtype=torch.int64, device=dev)
    st = E.DfsState(lb.data_ptr(), ub.data_ptr(), cap, state.data_ptr(), state.data_ptr() + 4, status.data_ptr(), counters.data_ptr(), None)
    n_rec = rec["status"].shape[0]
    assert n_rec == K
    for k in range(K):
        sp = int(state[0].item())
        assert sp >= 1
        top = sp - 1
        assert np.array_equal(lb[top].cpu().numpy(), rec["lb_in"][k]) and np.array_equal(ub[top].cpu().numpy(), rec["ub_in"][k]), f"node {k}: input differs from the oracle's"
        ctx._check(ctx._L.pcp_dfs_device(ctx._h, C.byref(st), 1, 0, 0, None))
        torch.cuda.synchronize()
        assert int(status[top].item()) == int(rec["status"][k]), f"node {k}: status"
    assert int(counters[0].item()) == K and int(counters[3].item()) == 0


@pytest.mark.parametrize("implicit", [False, True])
def test_set_mode_nqueens_1000(implicit):
    """N-queens-1000 over IntervalSet<i32> domains (FDSpace, what example/src/nqueens.rs:34 allocates) at the benchmarked size
    (parity corner a): 16 nodes of the set-mode frontier and 16 nodes on top of the stack after a short dive —
    125 KB of sets per node, 16 words per variable — against `consistency_set`, explicit rows and implicit nodes."""
    import torch
    from pcp_amd.search_device import DeviceSearch
    ctx = E.Context(0)
    props = M.nqueens_props(N)
    sw = (N + 63) // 64
    ctx.set_model(N, props, set_words=sw)
    ctx.set_hull(1, N)
    om = orc.OracleModel(N, props)
    B, _, _ = W.nqueens_frontier_set(ctx, N, 64)
    pick = np.arange(0, B.shape[0], max(1, B.shape[0] // 16))[:16]
    ds = DeviceSearch(ctx, batch=16, capacity=1024, implicit=True)
    ds.reset(np.ones(N, np.int32), np.full(N, N, np.int32), 1)
    ds.advance(max_rounds=60, batch=1)
    ds.advance(max_rounds=2, batch=16)
    ds.compact()  # rows [0, size) are the open nodes, bottom to top
    k = min(16, ds.size)
    deep = ds.bits[ds.size - k:ds.size].cpu().numpy().view(np.uint64)
    assert k >= 8
    bits = np.concatenate([B[pick], deep])
    n = bits.shape[0]
    ref = om.consistency_set(bits, 1, None, check_dup=False)
    if implicit:
        got = ctx.propagate_set(bits, None)
        assert ctx.last_plan()["implicit_active"] == 1 and ctx.last_plan()["set_mode"] == 1
        gact = None
    else:
        ctx.set_option("implicit_active", 0)
        got = ctx.propagate_set(bits, E.full_active(n, om.n_units))
        ctx.set_option("implicit_active", 1)
        assert ctx.last_plan()["implicit_active"] == 0
        gact = got[3]
    assert np.array_equal(ref[4], got[4]), (ref[4], got[4])
    ok = ref[4] != 0
    assert np.array_equal(ref[2][ok], got[2][ok]) and np.array_equal(ref[0][ok], got[0][ok]) and np.array_equal(ref[1][ok], got[1][ok])
    if gact is not None:
        assert np.array_equal(ref[3][ok], gact[ok])
    assert (got[2] != bits).any()  # interior values were removed
    ctx.close()
