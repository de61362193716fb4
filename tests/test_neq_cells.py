"""pcp_device_batch.cell_format PCP_CELLS_PACKED16 (ABI v7): the all-XNeqY kernel reads and writes rows of packed cells
(-lb & 0xffff | ub << 16, the format it keeps in LDS) instead of two int32 rows per node.  The format changes the bytes a node takes in
HBM, never a result: pack -> launch -> unpack must be bit-identical to the int32 launch and to the oracle (status, domains), for full and
ragged tiles, every tile size, with and without hints, in place and out of place; pack / unpack are exact inverses inside +-16383 and
raise the sticky hull flag outside."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
import pcp_amd.engine as E

from test_neq_path import neq_model, nodes_with_assignments

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = E.Context(0)
    yield c
    c.close()


def launch_i32(ctx, L, U, hint=None):
    import torch
    dev = torch.device("cuda", ctx.device)
    lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    st = torch.zeros(L.shape[0], dtype=torch.uint8, device=dev)
    d = None if hint is None else torch.from_numpy(np.ascontiguousarray(hint, np.int32)).to(dev)
    ctx.propagate_device(L.shape[0], lb, ub, lb, ub, None, None, st, dirty=d)
    torch.cuda.synchronize()
    return lb.cpu().numpy(), ub.cpu().numpy(), st.cpu().numpy()


def launch_cells(ctx, L, U, hint=None, in_place=True):
    import torch
    dev = torch.device("cuda", ctx.device)
    lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    cells = ctx.pack_rows(lb, ub)
    out = cells if in_place else torch.full_like(cells, 0x7fff7fff)
    st = torch.zeros(L.shape[0], dtype=torch.uint8, device=dev)
    d = None if hint is None else torch.from_numpy(np.ascontiguousarray(hint, np.int32)).to(dev)
    ctx.propagate_device(L.shape[0], cells, None, out, None, None, None, st, dirty=d, cells=True)
    assert ctx.last_plan()["path"] == 1
    l2, u2 = ctx.unpack_rows(out)
    torch.cuda.synchronize()
    return l2.cpu().numpy(), u2.cpu().numpy(), st.cpu().numpy()


def test_pack_unpack_round_trip(ctx):
    import torch
    dev = torch.device("cuda", ctx.device)
    rng = np.random.default_rng(5)
    for V, n in ((1000, 64), (37, 5), (4, 1), (2, 3)):
        ctx.set_model(V, neq_model(V, V, 3, (0, 5)))
        L = rng.integers(-16383, 16384, size=(n, V)).astype(np.int32)
        U = rng.integers(-16383, 16384, size=(n, V)).astype(np.int32)
        L[0, 0], U[0, 0] = -16383, 16383
        lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
        cells = ctx.pack_rows(lb, ub)
        want = ((-L.astype(np.int64)) & 0xffff) | ((U.astype(np.int64) & 0xffff) << 16)
        assert np.array_equal(cells.cpu().numpy().view(np.uint32).astype(np.int64), want)
        l2, u2 = ctx.unpack_rows(cells)
        assert np.array_equal(l2.cpu().numpy(), L) and np.array_equal(u2.cpu().numpy(), U)
        # unaligned views take the scalar path
        if n * V > 8:
            fl, fu = lb.reshape(-1)[1:], ub.reshape(-1)[1:]
            k = (fl.numel() // V) * V
            l1, u1 = fl[:k].reshape(-1, V), fu[:k].reshape(-1, V)
            c1 = torch.empty(k + 1, dtype=torch.int32, device=dev)[1:].reshape(-1, V)
            ctx.pack_rows(l1, u1, c1)
            l3, u3 = ctx.unpack_rows(c1)
            assert torch.equal(l3, l1) and torch.equal(u3, u1)
    ctx.stats_reset()
    ctx.stats_read()


def test_pack_outside_the_range_raises_the_hull_flag(ctx):
    import torch
    dev = torch.device("cuda", ctx.device)
    V = 8
    ctx.set_model(V, neq_model(1, V, 10, (0, 5)))
    ctx.stats_reset()
    lb = torch.zeros((2, V), dtype=torch.int32, device=dev)
    ub = torch.full((2, V), 5, dtype=torch.int32, device=dev)
    ub[1, 3] = 16384
    ctx.pack_rows(lb, ub)
    with pytest.raises(E.PcpError):
        ctx.stats_read()
    ctx.stats_reset()
    ctx.stats_read()


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("npb", [0, 16, 8, 1])
def test_cells_equal_int32_rows_random_models(ctx, seed, npb):
    V, P, dom = 40 + 8 * seed, 200 + 60 * seed, (0, 8 + seed)
    props = neq_model(70 + seed, V, P, dom)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(dom[0], dom[1])
    ctx.set_option("small_path", 0)
    ctx.set_option("nodes_per_block", npb)
    n = 16 * 300 + (7 if seed != 2 else 0)  # full tiles and a ragged last one
    L, U = nodes_with_assignments(500 + seed, V, n, dom, p_assign=0.08 + 0.05 * seed)
    Lo, Uo = L.copy(), U.copy()
    L[3, 1], U[3, 1] = 4, 3          # an empty input domain: the node fails (the oracle, like the reference, cannot even allocate it)
    ref = om.consistency(Lo, Uo, None)
    ref[3][3] = 0
    a = launch_i32(ctx, L.copy(), U.copy())
    assert a[2][3] == 0
    for in_place in (True, False):
        c = launch_cells(ctx, L.copy(), U.copy(), in_place=in_place)
        assert np.array_equal(c[2], a[2]) and np.array_equal(c[2], ref[3])
        ok = c[2] != 0  # (a failed node's domains are unspecified: oracle.py)
        assert np.array_equal(c[0][ok], a[0][ok]) and np.array_equal(c[1][ok], a[1][ok])
        assert np.array_equal(c[0][ok], ref[0][ok]) and np.array_equal(c[1][ok], ref[1][ok])
    ctx.set_option("nodes_per_block", 0)


def test_cells_queens_frontier_and_children_with_hints(ctx):
    """N-queens-200: root copies (the frontier shape) and two generations of hinted children, kept as cells between the launches."""
    import torch
    dev = torch.device("cuda", ctx.device)
    n_q = 200
    props = M.nqueens_props(n_q)
    V = n_q
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(1, n_q)
    rng = np.random.default_rng(2)
    n = 512
    L = np.ones((n, V), np.int32); U = np.full((n, V), n_q, np.int32)
    for i in range(n):
        v = rng.integers(0, V); x = rng.integers(1, n_q + 1)
        L[i, v] = U[i, v] = x
    ref = om.consistency(L.copy(), U.copy(), None)
    c = launch_cells(ctx, L.copy(), U.copy())
    assert np.array_equal(c[2], ref[3]) and np.array_equal(c[0], ref[0]) and np.array_equal(c[1], ref[1])
    # children: branch on int32 rows (the brancher's format), pack, launch with hints
    lb, ub = torch.from_numpy(c[0]).to(dev), torch.from_numpy(c[1]).to(dev)
    st = torch.from_numpy(c[2]).to(dev)
    cl = torch.zeros((2 * n, V), dtype=torch.int32, device=dev); cu = torch.zeros_like(cl)
    cd = torch.full((2 * n,), -1, dtype=torch.int32, device=dev)
    counts = torch.zeros(5, dtype=torch.int32, device=dev)
    ctx.branch_device(n, lb, ub, None, st, cl, cu, None, counts, child_dirty=cd)
    k = int(counts[0].item())
    assert k == 2 * int((c[2] == 2).sum()) and k > 0
    CL, CU, CD = cl[:k].cpu().numpy(), cu[:k].cpu().numpy(), cd[:k].cpu().numpy()
    ref2 = om.consistency(CL.copy(), CU.copy(), None)
    for hint in (None, CD):
        c2 = launch_cells(ctx, CL.copy(), CU.copy(), hint=hint)
        assert np.array_equal(c2[2], ref2[3])
        ok = c2[2] != 0
        assert np.array_equal(c2[0][ok], ref2[0][ok]) and np.array_equal(c2[1][ok], ref2[1][ok])


def test_cells_refusals(ctx):
    """What the format cannot carry is refused, not approximated: no declared hull, a model with other propagators, explicit active rows."""
    import torch
    dev = torch.device("cuda", ctx.device)
    V = 24
    props = neq_model(9, V, 60, (0, 6))
    L, U = nodes_with_assignments(9, V, 64, (0, 6), p_assign=0.1)
    lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    st = torch.zeros(64, dtype=torch.uint8, device=dev)
    ctx.set_model(V, props)  # no hull
    ctx.set_option("small_path", 0)
    cells = ctx.pack_rows(lb, ub)
    with pytest.raises(E.PcpError):
        ctx.propagate_device(64, cells, None, cells, None, None, None, st, cells=True)
    ctx.set_hull(0, 6)
    act = torch.zeros((64, max(ctx.words, 1)), dtype=torch.int64, device=dev)
    with pytest.raises(E.PcpError):
        ctx.propagate_device(64, cells, None, cells, None, None, act, st, cells=True)
    ctx.propagate_device(64, cells, None, cells, None, None, None, st, cells=True)
    torch.cuda.synchronize()
    mixed = props.copy()
    mixed["kind"][0] = M.LT
    ctx.set_model(V, mixed)
    ctx.set_hull(0, 6)
    ctx.set_option("small_path", 0)
    with pytest.raises(E.PcpError):
        ctx.propagate_device(64, cells, None, cells, None, None, None, st, cells=True)


@pytest.mark.parametrize("reverse", [0, 1])
def test_branch_on_cells_equals_branch_on_rows(ctx, reverse):
    """pcp_branch_device_cells = pcp_branch_device_hint through pack / unpack: same children in the same rows, same hints, same counts —
    including nodes that are not Unknown (skipped) and negative bounds (MiddleVal truncates toward zero)."""
    import torch
    dev = torch.device("cuda", ctx.device)
    V, dom = 33, (-7, 9)
    props = neq_model(21, V, 90, dom)
    ctx.set_model(V, props)
    ctx.set_hull(dom[0], dom[1])
    ctx.set_option("small_path", 0)
    n = 200
    L, U = nodes_with_assignments(77, V, n, dom, p_assign=0.25)
    lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.propagate_device(n, lb, ub, lb, ub, None, None, st)
    assert len(set(st.cpu().tolist())) >= 2
    ctx.set_option("branch_reverse", reverse)
    cl = torch.zeros((2 * n, V), dtype=torch.int32, device=dev); cu = torch.zeros_like(cl)
    cd = torch.full((2 * n,), -5, dtype=torch.int32, device=dev)
    cnt = torch.zeros(5, dtype=torch.int32, device=dev)
    ctx.branch_device(n, lb, ub, None, st, cl, cu, None, cnt, child_dirty=cd)
    cells = ctx.pack_rows(lb, ub)
    cc = torch.zeros((2 * n, V), dtype=torch.int32, device=dev)
    cd2 = torch.full((2 * n,), -5, dtype=torch.int32, device=dev)
    cnt2 = torch.zeros(5, dtype=torch.int32, device=dev)
    ctx.branch_device_cells(n, cells, st, cc, cnt2, child_dirty=cd2)
    ctx.set_option("branch_reverse", 0)
    torch.cuda.synchronize()
    assert torch.equal(cnt, cnt2)
    k = int(cnt[0].item())
    assert k == 2 * int((st == 2).sum().item()) and k > 0
    l2, u2 = ctx.unpack_rows(cc[:k].contiguous())
    assert torch.equal(l2, cl[:k]) and torch.equal(u2, cu[:k]) and torch.equal(cd[:k], cd2[:k])
    assert (cd2[k:] == -5).all() and (cc[k:] == 0).all()


@pytest.mark.parametrize("n", [8, 10])
def test_device_search_on_cells_is_the_same_tree(ctx, n):
    """DeviceSearch(cells=True): the open nodes stay packed cells from the root to the leaves (pcp_propagate_device with cell_format,
    pcp_branch_device_cells).  Node for node the oracle's tree, and the same solutions in the same order as the search over int32 rows."""
    from pcp_amd.search_device import DeviceSearch
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    ctx.set_hull(1, n)
    ctx.set_option("small_path", 0)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    ss = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=True)[0]
    want = (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
    for batch in (1, 7, 64):
        base = DeviceSearch(ctx, batch=batch, capacity=4096, implicit=True).run(lb0, ub0, all_solutions=True, keep_solutions=50)
        for hints in (True, False):
            ds = DeviceSearch(ctx, batch=batch, capacity=4096, implicit=True, hints=hints, cells=True)
            assert ds.ub is None
            st = ds.run(lb0, ub0, all_solutions=True, keep_solutions=50)
            assert (st.num_nodes, st.num_solution, st.num_failed_node) == want, (batch, hints)
            assert len(st.solutions) == len(base.solutions) and all(np.array_equal(a, b) for a, b in zip(st.solutions, base.solutions))
    lim = DeviceSearch(ctx, batch=7, capacity=4096, implicit=True, cells=True).run(lb0, ub0, all_solutions=True, node_limit=40)
    ref = DeviceSearch(ctx, batch=7, capacity=4096, implicit=True).run(lb0, ub0, all_solutions=True, node_limit=40)
    assert (lim.num_nodes, lim.num_solution, lim.num_failed_node) == (ref.num_nodes, ref.num_solution, ref.num_failed_node)
    ctx.set_option("small_path", 1)


@pytest.mark.parametrize("n", [64, 7])
def test_cell_outside_the_format_refuses_its_node(ctx, n):
    """A cell that pcp_pack_rows could not have written (a half outside +-16384) refuses its node — status PCP_STATUS_HULL, the row untouched,
    the sticky flag raised — in the full-tile and the ragged staging loop; the other nodes of the tile are propagated as usual."""
    import torch
    dev = torch.device("cuda", ctx.device)
    V, dom = 24, (0, 6)
    props = neq_model(9, V, 60, dom)
    ctx.set_model(V, props)
    ctx.set_hull(*dom)
    ctx.set_option("small_path", 0)
    L, U = nodes_with_assignments(9, V, n, dom, p_assign=0.1)
    ref = orc.OracleModel(V, props).consistency(L.copy(), U.copy(), None)
    lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    cells = ctx.pack_rows(lb, ub)
    cells[5, 3] = 0x7fff0000          # ub = 32767
    before = cells[5].clone()
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.stats_reset()
    ctx.propagate_device(n, cells, None, cells, None, None, None, st, cells=True)
    l2, u2 = ctx.unpack_rows(cells)
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    assert st[5] == 0xFE and torch.equal(cells[5], before)
    keep = np.arange(n) != 5
    assert np.array_equal(st[keep], ref[3][keep])
    ok = keep & (st != 0)
    assert np.array_equal(l2.cpu().numpy()[ok], ref[0][ok]) and np.array_equal(u2.cpu().numpy()[ok], ref[1][ok])
    with pytest.raises(E.PcpError):
        ctx.stats_read()
    ctx.stats_reset()
    ctx.stats_read()
    ctx.set_option("small_path", 1)
