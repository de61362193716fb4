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
