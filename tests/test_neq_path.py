"""The assignment-driven kernel of all-XNeqY models (pcp_amd/csrc/pcp_neq.hip, plan.path == 1) against the oracle, through the
C-ABI: XNeqY::propagate acts only with a singleton on one side (propagators/cmp/x_neq_y.rs:82-93), so the initial sweep
(init_scheduler, propagation/store.rs:144-149) is the adjacency lists of the assigned variables.  Bit-exact:
status, bounds and the `active` rows materialised on request; every case also against the generic kernels (neq_path = 0)."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
import pcp_amd.engine as E

from util import assert_parity, splitmix64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = E.Context(0)
    yield c
    c.close()


def neq_model(seed, n_vars, n_props, dom=(0, 12), p_const=0.15, max_off=4):
    """Random all-XNeqY model: x != y + c over Addition views, some operands Constants (either side)."""
    rng = splitmix64(seed)
    props = np.zeros(n_props, dtype=M.PROP_DTYPE)
    props["var"][:] = M.PCP_NOVAR
    props["group"] = np.arange(n_props)
    props["kind"] = M.NEQ
    for r in range(n_props):
        x, y = rng.choice(n_vars, size=2, replace=False)
        ox, oy = int(rng.integers(-max_off, max_off + 1)), int(rng.integers(-max_off, max_off + 1))
        props[r]["var"][0], props[r]["off"][0] = x, ox
        props[r]["var"][1], props[r]["off"][1] = y, oy
        u = rng.random()
        if u < p_const / 2:
            props[r]["var"][0], props[r]["off"][0] = M.PCP_CONST, int(rng.integers(dom[0], dom[1] + 1))
        elif u < p_const:
            props[r]["var"][1], props[r]["off"][1] = M.PCP_CONST, int(rng.integers(dom[0], dom[1] + 1))
    return props


def nodes_with_assignments(seed, n_vars, n_nodes, dom, p_assign=0.3, p_narrow=0.3):
    rng = splitmix64(seed)
    L = np.full((n_nodes, n_vars), dom[0], np.int32)
    U = np.full((n_nodes, n_vars), dom[1], np.int32)
    for k in range(n_nodes):
        pa = rng.random() * p_assign  # from almost free nodes to heavily assigned ones
        for v in range(n_vars):
            u = rng.random()
            if u < pa:
                L[k, v] = U[k, v] = rng.integers(dom[0], dom[1] + 1)
            elif u < pa + p_narrow:
                a = int(rng.integers(dom[0], dom[1] + 1)); b = int(rng.integers(a, dom[1] + 1))
                L[k, v], U[k, v] = a, b
    return L, U


def run_both_paths(ctx, om, L, U, what, in_place=True):
    ref = om.consistency(L, U, None)
    ctx.set_option("neq_path", 1); ctx.set_option("small_path", 0)  # (a handful of nodes of a small store would take pcp_small.hip)
    got = ctx.propagate_implicit(L, U, in_place=in_place)
    pl = ctx.last_plan()
    assert pl["path"] == 1 and pl["implicit_active"] == 1, pl
    assert_parity(ref[:4], got[:4], what + " [assignment-driven]")
    ctx.set_option("neq_path", 0); ctx.set_option("small_path", 0)
    gen = ctx.propagate_implicit(L, U, in_place=in_place)
    assert ctx.last_plan()["path"] == 0
    ctx.set_option("neq_path", 1); ctx.set_option("small_path", 1)
    assert_parity(ref[:4], gen[:4], what + " [generic]")
    return ref, got, pl


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("hull", [True, False])
def test_random_neq_models(ctx, seed, hull):
    """Random sparse XNeqY networks with Addition offsets and Constant operands, many assigned variables (failures,
    cascades, True nodes), 16-bit cells under a declared hull and 32-bit cells without one; every tile size."""
    V, P, dom = 40 + 7 * seed, 200 + 40 * seed, (0, 9 + seed)
    props = neq_model(seed, V, P, dom)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    if hull:
        ctx.set_hull(dom[0], dom[1])
    L, U = nodes_with_assignments(100 + seed, V, 300, dom, p_assign=0.15 + 0.05 * seed)
    for npb, blk in ((0, 0), (16, 512), (8, 256), (4, 1024), (1, 0), (2, 512)):
        ctx.set_option("nodes_per_block", npb)
        ctx.set_option("neq_block", blk)
        ref, got, pl = run_both_paths(ctx, om, L, U, f"neq seed={seed} hull={hull} npb={npb} blk={blk}")
        assert pl["packed"] == int(hull) and (npb == 0 or pl["nodes_per_block"] == npb) and (blk == 0 or pl["block"] == blk), pl
    ctx.set_option("nodes_per_block", 0)
    ctx.set_option("neq_block", 0)
    st = ref[3]
    assert (st == 0).any() and (st == 2).any()


def test_statuses_true_false_unknown(ctx):
    """Hand-made nodes: a full assignment (True: every record entailed), a clash (False), a free node (Unknown), an unassigned
    variable whose every record is entailed (True although a variable is not a singleton)."""
    V = 4
    vs, cs = M.VStore(), M.CStore()
    xs = [vs.alloc((0, 9)) for _ in range(V)]
    cs.alloc(M.XNeqY(xs[0], xs[1]))
    cs.alloc(M.XNeqY(xs[1], xs[2]))
    cs.alloc(M.XNeqY(xs[2], M.Addition(xs[3], 1)))
    props = cs.lower(V)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(0, 9)
    L = np.array([[1, 2, 3, 5], [1, 1, 3, 5], [0, 0, 0, 0], [0, 3, 0, 7], [2, 3, 4, 3]], np.int32)
    U = np.array([[1, 2, 3, 5], [1, 1, 3, 5], [9, 9, 9, 9], [2, 6, 2, 9], [2, 3, 4, 3]], np.int32)
    ref, got, _ = run_both_paths(ctx, om, L, U, "hand-made statuses")
    assert ref[3].tolist() == [1, 0, 2, 1, 0]


@pytest.mark.parametrize("n", [8, 20, 50])
def test_nqueens_dfs_nodes_assignment_driven(ctx, n):
    """The first 200 nodes of the reference's DFS on N-queens-n (folded input domains), as implicit nodes, in place and not."""
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    _, _, rec, _ = om.search(np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True, node_limit=200, max_records=200)
    keep = (rec["lb_in"] <= rec["ub_in"]).all(axis=1)
    ctx.set_model(n, props)
    ctx.set_hull(1, n)
    for in_place in (True, False):
        run_both_paths(ctx, om, rec["lb_in"][keep], rec["ub_in"][keep], f"nqueens({n}) in_place={in_place}", in_place=in_place)


def test_unaligned_rows_and_ragged_tiles(ctx):
    """V not a multiple of four (no 16-byte row loads), a last tile with missing nodes, a refused node (bound outside the hull)."""
    V, dom = 37, (0, 15)
    props = neq_model(77, V, 260, dom, p_const=0.1)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(dom[0], dom[1])
    L, U = nodes_with_assignments(78, V, 53, dom)
    ctx.set_option("nodes_per_block", 16)
    run_both_paths(ctx, om, L, U, "unaligned rows")
    # a node outside +-16383 under a declared 16-bit hull: refused (PCP_STATUS_HULL), outputs untouched, sticky violation
    L2, U2 = L.copy(), U.copy()
    U2[5, 3] = 20000
    ctx.set_option("neq_path", 1)
    import torch
    dev = torch.device("cuda", 0)
    t_lb, t_ub = torch.from_numpy(L2).to(dev), torch.from_numpy(U2).to(dev)
    t_st = torch.zeros(L2.shape[0], dtype=torch.uint8, device=dev)
    ctx.propagate_device(L2.shape[0], t_lb, t_ub, t_lb, t_ub, None, None, t_st)
    torch.cuda.synchronize()
    st = t_st.cpu().numpy()
    assert st[5] == 0xFE and np.array_equal(t_ub[5].cpu().numpy(), U2[5])
    ref = om.consistency(L, U, None)
    others = np.arange(L.shape[0]) != 5
    assert np.array_equal(st[others], ref[3][others])
    with pytest.raises(E.PcpError):
        ctx.stats_read()
    ctx.stats_read()  # the flag was consumed
    ctx.set_option("nodes_per_block", 0)


def test_counters(ctx):
    """steps = every propagator of every node once + the wake-ups; evaluated = item tests; full_evals <= evaluated."""
    n = 30
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    ctx.set_hull(1, n)
    L = np.ones((7, n), np.int32); U = np.full((7, n), n, np.int32)
    L[:, 0] = U[:, 0] = np.arange(1, 8)
    got = ctx.propagate_implicit(L, U, want_active=False)
    st = got[4]
    assert st["nodes"] == 7 and st["steps"] >= 7 * len(props) and st["evaluated"] >= 7 * 3 * (n - 1) and st["full_evals"] <= st["evaluated"]
    assert st["narrowings"] > 0


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_round0_list_built_by_staging(ctx, seed):
    """16-node tiles (pcp_neq.hip, round 0's list built by the staging loop for up to 32 assigned variables per tile, the scan beyond): tiles
    with 1-3, with exactly 32 and with 33-60 distinct assigned variables, variables assigned in several nodes of a tile (shared masks), Constant
    neighbours (seed_always), failing nodes and — in a tile of their own — nodes outside the hull (refused: they must not appear in any mask).
    Against the oracle and the generic kernel; the debug bit that turns the direct list off gives the same results."""
    n, dom = 96, (0, 40)
    props = neq_model(seed, n, 700, dom=dom, p_const=0.1, max_off=3)
    om = orc.OracleModel(n, props)
    ctx.set_model(n, props)
    ctx.set_hull(dom[0], dom[1])
    rng = np.random.default_rng(seed)
    N = 16 * 6
    L = np.full((N, n), dom[0], np.int32); U = np.full((N, n), dom[1], np.int32)
    per_tile = [int(rng.integers(1, 4)), 32, int(rng.integers(33, 61)), int(rng.integers(1, 33)), 5, int(rng.integers(1, 33))]
    for t, k in enumerate(per_tile):
        vs = rng.choice(n, size=k, replace=False)              # the tile's assigned variables ...
        for b in range(16):
            mine = vs[rng.random(k) < 0.7] if k > 1 else vs     # ... each node has most of them
            L[16 * t + b, mine] = U[16 * t + b, mine] = rng.integers(dom[0], dom[1] + 1, size=len(mine))
            w = rng.choice(n, size=6, replace=False)
            a_ = rng.integers(dom[0], dom[1] + 1, size=6); b_ = rng.integers(dom[0], dom[1] + 1, size=6)
            free = L[16 * t + b, w] != U[16 * t + b, w]
            L[16 * t + b, w[free]] = np.minimum(a_, b_)[free]; U[16 * t + b, w[free]] = np.maximum(a_, b_)[free]
    for k, v in {"nodes_per_block": 16, "neq_debug": 0}.items():
        ctx.set_option(k, v)
    try:
        ref, got, pl = run_both_paths(ctx, om, L, U, f"round-0 list seed={seed}")
        assert pl["nodes_per_block"] == 16
        assert (ref[3] == 0).any() and (ref[3] != 0).any()
        ctx.set_option("neq_debug", 16384)  # the scan builds every round's list
        ctx.set_option("small_path", 0)
        alt = ctx.propagate_implicit(L, U)
        assert_parity(ref[:4], alt[:4], "round-0 list by the scan")
        # refused nodes (a bound beyond the 16-bit cells) sharing tiles with ordinary ones: left out of every mask, the others unaffected
        import torch
        L2, U2 = L[:32].copy(), U[:32].copy()
        U2[3, 7] = 20000; L2[20, 9] = -20000
        ctx.set_option("neq_debug", 0); ctx.set_option("neq_path", 1)
        dev = torch.device("cuda", 0)
        t_lb, t_ub = torch.from_numpy(L2).to(dev), torch.from_numpy(U2).to(dev)
        t_st = torch.zeros(32, dtype=torch.uint8, device=dev)
        ctx.propagate_device(32, t_lb, t_ub, t_lb, t_ub, None, None, t_st)
        torch.cuda.synchronize()
        st = t_st.cpu().numpy()
        assert ctx.last_plan()["path"] == 1
        assert st[3] == 0xFE and st[20] == 0xFE
        others = np.ones(32, bool); others[[3, 20]] = False
        assert np.array_equal(st[others], ref[3][:32][others])
        ok = others & (ref[3][:32] != 0)
        assert np.array_equal(t_lb.cpu().numpy()[ok], ref[0][:32][ok]) and np.array_equal(t_ub.cpu().numpy()[ok], ref[1][:32][ok])
        with pytest.raises(E.PcpError):
            ctx.stats_read()
        ctx.stats_read()  # the flag was consumed
    finally:
        for k, v in {"nodes_per_block": 0, "neq_debug": 0, "small_path": 1}.items():
            ctx.set_option(k, v)
