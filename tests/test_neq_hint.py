"""pcp_device_batch.dirty_var (ABI v7): a node that is a propagated parent's row with ONE variable branched on may be propagated from that
variable alone.  The reference re-schedules every propagator of a node (init_scheduler, propagation/store.rs:144-149) and finds all but
those of the changed variable no-ops (an entailed or quiescent propagator's propagate() does nothing until one of its variables changes:
Store::react, store.rs:191-198), so status, domains and derived `active` rows must be bit-identical with and without the hint, and equal
to the oracle's, which knows nothing of hints.  pcp_branch_device_hint is the producer: it writes the hint of every child."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
import pcp_amd.engine as E

from test_neq_path import neq_model, nodes_with_assignments
from util import assert_parity

pytestmark = pytest.mark.gpu
NOVAR = -1


@pytest.fixture(scope="module")
def ctx():
    c = E.Context(0)
    yield c
    c.close()


def children_with_hints(ctx, L, U):
    """Propagate the nodes (no hints), branch the Unknown ones on the device: (child lb, child ub, child hint, parents' statuses)."""
    import torch
    dev = torch.device("cuda", ctx.device)
    n, V = L.shape
    lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.propagate_device(n, lb, ub, lb, ub, None, None, st)
    cl = torch.zeros((2 * n, V), dtype=torch.int32, device=dev); cu = torch.zeros_like(cl)
    cd = torch.full((2 * n,), -7, dtype=torch.int32, device=dev)
    counts = torch.zeros(5, dtype=torch.int32, device=dev)
    ctx.branch_device(n, lb, ub, None, st, cl, cu, None, counts, child_dirty=cd)
    torch.cuda.synchronize()
    k = int(counts[0].item())
    return cl[:k].cpu().numpy(), cu[:k].cpu().numpy(), cd[:k].cpu().numpy(), st.cpu().numpy(), lb.cpu().numpy(), ub.cpu().numpy()


def launch(ctx, L, U, hint):
    import torch
    dev = torch.device("cuda", ctx.device)
    lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    st = torch.zeros(L.shape[0], dtype=torch.uint8, device=dev)
    act = torch.zeros((L.shape[0], max(ctx.words, 1)), dtype=torch.int64, device=dev)
    d = None if hint is None else torch.from_numpy(np.ascontiguousarray(hint, np.int32)).to(dev)
    ctx.stats_reset()
    ctx.propagate_device(L.shape[0], lb, ub, lb, ub, None, act, st, dirty=d)
    torch.cuda.synchronize()
    return lb.cpu().numpy(), ub.cpu().numpy(), act.cpu().numpy().view(np.uint64)[:, :ctx.words], st.cpu().numpy(), ctx.stats_read()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("hull", [True, False])
def test_hinted_children_of_random_models(ctx, seed, hull):
    """Random XNeqY networks with offsets and Constant operands: the children pcp_branch_device_hint makes of propagated nodes, launched with
    their hints, without them, with hints on every other node only (mixed tiles), and with the hints switched off by option; every tile
    size.  All four bit-identical to the oracle."""
    V, P, dom = 40 + 9 * seed, 220 + 50 * seed, (0, 9 + seed)
    props = neq_model(40 + seed, V, P, dom)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    if hull:
        ctx.set_hull(dom[0], dom[1])
    ctx.set_option("small_path", 0)
    L, U = nodes_with_assignments(300 + seed, V, 400, dom, p_assign=0.1 + 0.04 * seed)
    CL, CU, CD, pst, PL, PU = children_with_hints(ctx, L, U)
    assert CL.shape[0] == 2 * int((pst == 2).sum()) and CL.shape[0] >= 64
    # the hint is the branched variable: FirstSmallestVar of the parent's fixpoint, the same for both children, in tree order
    unk = np.nonzero(pst == 2)[0]
    for j, p in enumerate(unk[:50]):
        size = PU[p].astype(np.int64) - PL[p] + 1
        var = int(np.argmin(np.where(size > 1, size, 1 << 40)))
        assert CD[2 * j] == var and CD[2 * j + 1] == var
        assert CU[2 * j, var] < PU[p, var] and CL[2 * j + 1, var] > PL[p, var]
    ref = om.consistency(CL, CU, None)
    for npb, blk in ((0, 0), (16, 512), (4, 1024), (1, 0)):
        ctx.set_option("nodes_per_block", npb); ctx.set_option("neq_block", blk)
        plain = launch(ctx, CL, CU, None)
        assert ctx.last_plan()["path"] == 1
        assert_parity(ref[:4], plain[:4], f"children, no hints npb={npb}")
        hinted = launch(ctx, CL, CU, CD)
        assert ctx.last_plan()["path"] == 1
        assert_parity(ref[:4], hinted[:4], f"children, hinted npb={npb}")
        assert hinted[4]["evaluated"] <= plain[4]["evaluated"] * 1.02 + 64
        half = CD.copy(); half[::2] = NOVAR; half[5::7] = V + 3  # (>= n_vars: no hint)
        assert_parity(ref[:4], launch(ctx, CL, CU, half)[:4], f"children, mixed tiles npb={npb}")
        ctx.set_option("neq_hint", 0)
        off = launch(ctx, CL, CU, CD)
        ctx.set_option("neq_hint", 1)
        assert_parity(ref[:4], off[:4], f"children, hints ignored npb={npb}")
        assert off[4]["evaluated"] > hinted[4]["evaluated"] or hinted[4]["evaluated"] == plain[4]["evaluated"]  # (counts vary a little with the schedule: never compared exactly)
    ctx.set_option("nodes_per_block", 0); ctx.set_option("neq_block", 0); ctx.set_option("small_path", 1)
    assert (ref[3] == 2).any()


def test_hinted_nqueens_1000_deep_nodes(ctx):
    """N-queens n = 1000 (the benchmarked store, 16-node tiles of 16-bit cells): children of nodes ~3000 nodes down a dive (about 170 queens
    assigned: without a hint a node restarts from all of them).  Hinted and plain launches are bit-identical; 24 nodes against the oracle;
    the hinted launch tests far fewer (entry, node) pairs."""
    from pcp_amd import workloads as W
    N = 1000
    props = M.nqueens_props(N)
    ctx.set_model(N, props)
    ctx.set_hull(1, N)
    lb, ub, _ = W.nqueens_deep(ctx, N, 3000, 2048, implicit=True)
    CL, CU, CD, pst, _, _ = children_with_hints(ctx, lb.cpu().numpy(), ub.cpu().numpy())
    assert CL.shape[0] >= 1024 and (CD >= 0).all() and (CD < N).all()
    plain = launch(ctx, CL, CU, None)
    pl = ctx.last_plan()
    assert (pl["path"], pl["nodes_per_block"], pl["packed"]) == (1, 16, 1), pl
    hinted = launch(ctx, CL, CU, CD)
    assert np.array_equal(plain[3], hinted[3])
    ok = plain[3] != 0
    assert np.array_equal(plain[0][ok], hinted[0][ok]) and np.array_equal(plain[1][ok], hinted[1][ok]) and np.array_equal(plain[2][ok], hinted[2][ok])
    assert hinted[4]["evaluated"] * 20 < plain[4]["evaluated"], (hinted[4]["evaluated"], plain[4]["evaluated"])
    pick = np.linspace(0, CL.shape[0] - 1, 24).astype(np.int64)
    ref = orc.OracleModel(N, props).consistency(CL[pick], CU[pick], None)
    assert_parity(ref[:4], tuple(x[pick] for x in hinted[:4]), "n-queens-1000 hinted children")


@pytest.mark.parametrize("n", [8, 10])
def test_device_search_with_hints_is_the_same_tree(ctx, n):
    """DeviceSearch keeps a hint per open node (children from pcp_branch_device_hint, the root without): the batched search visits exactly
    the oracle's tree, with hints and without, for several batch sizes."""
    from pcp_amd.search_device import DeviceSearch
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    ctx.set_hull(1, n)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    ss = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=True)[0]
    want = (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
    for batch in (1, 7, 64):
        for hints in (True, False):
            ds = DeviceSearch(ctx, batch=batch, capacity=4096, implicit=True, hints=hints)
            assert (ds.dirty is not None) == hints
            st = ds.run(lb0, ub0, all_solutions=True)
            assert (st.num_nodes, st.num_solution, st.num_failed_node) == want, (batch, hints)
