"""Parity of the 10-bit-cell kernel (pcp_big.hip, pcp_plan.path == 2) — the kernel behind BASELINE config 3 as bench.py runs it:
implicit-active nodes through pcp_propagate_device under a declared hull of at most 1024 values.  Every test ASSERTS the path
(`last_plan()["path"] == 2`) before comparing with the oracle: a model that silently takes another kernel proves nothing here.

Reference semantics these pin: XLessY x_less_y.rs:104-109, XEqY x_eq_y.rs:102-107, XNeqY x_neq_y.rs:82-93 (a value is removed only
at a bound), variable::Store::update variable/store.rs:151-166, the engine propagation/store.rs:151-198.  Bit-exact (integer work)."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
import pcp_amd.engine as E

from util import assert_parity, planted_binary_csp, random_csp, random_nodes, unit_narrowing_prefix

pytestmark = pytest.mark.gpu

DEFAULTS = {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1,
            "big_path": 1, "big_round": 0, "dom10": 1, "neq_path": 1, "implicit_active": 1}


@pytest.fixture(scope="module")
def ctx():
    c = E.Context(0)
    yield c
    for k, v in DEFAULTS.items():
        c.set_option(k, v)
    c.close()


def run_big(ctx, om, L, U, what, **opts):
    """The batch as implicit nodes on path 2 (asserted), `active` rows materialised (derive_active after bigfix) and compared."""
    for k, v in {**DEFAULTS, **opts}.items():
        ctx.set_option(k, v)
    ctx.stats_reset()
    got = ctx.propagate_implicit(L, U, want_active=True)
    plan = ctx.last_plan()
    assert plan["path"] == 2 and plan["implicit_active"] == 1, (what, plan)
    ref = om.consistency(L, U, None)
    assert_parity(ref[:4], got[:4], what)
    return ref, got, ctx.debug_counters()


def test_config3_full_size_on_bigfix(ctx):
    """BASELINE config 3 exactly as bench.py's C3 leg launches it: V = 50 000, P = 500 000, seed 0xC3, implicit nodes, hull [0, 999]
    — no option forces anything: the store does not fit LDS as pairs and the batch has more nodes than half the CUs (the plan's rule,
    pcp_api.hip), so the plan must pick path 2 by itself.  The oracle needs ~1 s per node: 24 of the 136 nodes are compared."""
    V, P, N = 50_000, 500_000, 136
    props, lb, ub, sol = planted_binary_csp(0xC3, V, P)
    L, U = unit_narrowing_prefix(0xC3 + 1, lb, ub, sol, N)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(0, 999)
    for k, v in DEFAULTS.items():
        ctx.set_option(k, v)
    ctx.stats_reset()
    got = ctx.propagate_implicit(L, U, want_active=True)
    plan = ctx.last_plan()
    assert plan["path"] == 2 and plan["implicit_active"] == 1 and plan["grid"] == N, plan
    dbg = ctx.debug_counters()
    pick = np.r_[0:12, N - 12:N]
    ref = om.consistency(L[pick], U[pick], None)
    assert_parity(ref[:4], tuple(g[pick] for g in got[:4]), "config 3 on bigfix")
    assert (ref[3] == 2).all() and ((ref[0] != L[pick]) | (ref[1] != U[pick])).sum() > V  # long cascades, never fails
    assert (got[3] == 2).all()
    assert dbg["big_dense"] + dbg["big_sparse"] >= N  # every node ran wake-up rounds
    # the same launch with each form of wake-up round forced: both must reach the same fixpoint
    for mode in (1, 2):
        _, _, d = run_big(ctx, om, L[:8], U[:8], f"config 3 on bigfix, big_round={mode}", big_round=mode, global_dom=2)
        assert (d["big_dense"] > 0) == (mode == 1) and (d["big_sparse"] > 0) == (mode == 2), d
    ctx.set_model(V, props)  # forgets the hull


@pytest.mark.parametrize("seed", range(8))
def test_binary_random_csps_on_bigfix(ctx, seed):
    """Binary-only random CSPs (constants, negative offsets, all three kinds), planted and failing, hull widths from 2 to 1024 with
    bounds AT lo and lo + 1023 (the un-narrowed variables sit on both ends of the cell range), V and P not multiples of 3 / 64."""
    width = [1024, 2, 1023, 17, 1024, 300, 64, 5][seed]
    lo = [-512, 0, -1023, 7, 100000, -300, 1, -2][seed]
    planted = seed not in (2, 5, 7)
    V, P, N = 211 + 37 * seed, 1700 + 331 * seed, 70
    kinds = [None, None, None, [M.NEQ], [M.LT], [M.EQ, M.NEQ], None, None][seed]
    props, lb, ub, sol = random_csp(7700 + seed, V, P, dom=(lo, lo + width - 1), p_const=0.12, p_tern=0.0, planted=planted, kinds=kinds)
    assert (props["kind"] <= M.LT).all()
    L, U = random_nodes(7800 + seed, lb, ub, N, sol if planted else None, p_narrow=0.25 if planted else 0.03)
    if width >= 1023:
        assert (L == lo).any() and (U == lo + width - 1).any()
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(lo, lo + width - 1)
    tot = {"big_dense": 0, "big_sparse": 0}
    for mode in (0, 1, 2):
        ref, _, d = run_big(ctx, om, L, U, f"binary csp seed={seed} big_round={mode}", global_dom=2, big_round=mode)
        if mode == 1:
            assert d["big_sparse"] == 0
        if mode == 2:
            assert d["big_dense"] == 0
        for k in tot:
            tot[k] += d[k]
    if planted:
        assert (ref[3] != 0).all()
        assert tot["big_dense"] > 0 and tot["big_sparse"] > 0, tot  # (a failing store may fail every node in the sweep: no rounds)
    else:
        assert (ref[3] == 0).any()
    ctx.set_model(V, props)


def test_auto_rounds_take_both_forms(ctx):
    """With big_round = 0 (what every caller gets) one batch makes the kernel choose dense rounds for some (node, round) pairs and
    sparse rounds for others — counters, not options, prove both ran."""
    V, P, N = 3000, 30_000, 48
    props, lb, ub, sol = planted_binary_csp(0xB16, V, P, dom=(0, 1023))
    L, U = unit_narrowing_prefix(0xB17, lb, ub, sol, N, k=48)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(0, 1023)
    _, got, d = run_big(ctx, om, L, U, "auto rounds", global_dom=2)
    assert d["big_dense"] > 0 and d["big_sparse"] > 0, d
    assert got[4]["narrowings"] > 0 and got[4]["waves"] >= 2 * N
    ctx.set_model(V, props)


def test_bigfix_refuses_nodes_outside_the_hull(ctx):
    """A bound outside the declared hull: the node is refused (PCP_STATUS_HULL, outputs untouched, sticky violation reported by
    pcp_stats_read), its neighbours in the batch are unaffected (pcp_hip.h)."""
    import torch
    V, P, N = 400, 3000, 24
    props, lb, ub, sol = random_csp(9100, V, P, dom=(0, 1023), p_const=0.1, p_tern=0.0, planted=True)
    L, U = random_nodes(9101, lb, ub, N, sol, p_narrow=0.2)
    om = orc.OracleModel(V, props)
    ref = om.consistency(L, U, None)
    ctx.set_model(V, props)
    ctx.set_hull(0, 1023)
    for k, v in {**DEFAULTS, "global_dom": 2}.items():
        ctx.set_option(k, v)
    L2, U2 = L.copy(), U.copy()
    U2[3, 11] = 1024
    L2[17, 0] = -1
    dev = torch.device("cuda", 0)
    t_lb, t_ub = torch.from_numpy(L2).to(dev), torch.from_numpy(U2).to(dev)
    t_st = torch.zeros(N, dtype=torch.uint8, device=dev)
    ctx.stats_reset()
    ctx.propagate_device(N, t_lb, t_ub, t_lb, t_ub, None, None, t_st)
    torch.cuda.synchronize()
    assert ctx.last_plan()["path"] == 2
    st = t_st.cpu().numpy()
    assert st[3] == 0xFE and st[17] == 0xFE
    assert np.array_equal(t_ub[3].cpu().numpy(), U2[3]) and np.array_equal(t_lb[17].cpu().numpy(), L2[17])
    others = np.ones(N, bool); others[[3, 17]] = False
    assert np.array_equal(st[others], ref[3][others])
    ok = others & (ref[3] != 0)
    assert np.array_equal(t_lb.cpu().numpy()[ok], ref[0][ok]) and np.array_equal(t_ub.cpu().numpy()[ok], ref[1][ok])
    with pytest.raises(E.PcpError):
        ctx.stats_read()
    ctx.stats_read()
    ctx.set_model(V, props)


def test_hull_wider_than_1024_values_never_takes_bigfix(ctx):
    V, P, N = 300, 2500, 10
    props, lb, ub, sol = random_csp(9200, V, P, dom=(0, 1024), p_const=0.1, p_tern=0.0, planted=True)
    L, U = random_nodes(9201, lb, ub, N, sol, p_narrow=0.2)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(0, 1024)
    for k, v in {**DEFAULTS, "global_dom": 2}.items():
        ctx.set_option(k, v)
    got = ctx.propagate_implicit(L, U)
    assert ctx.last_plan()["path"] != 2
    assert_parity(om.consistency(L, U, None)[:4], got[:4], "hull of 1025 values")
    ctx.set_model(V, props)


def test_host_stepped_dfs_on_a_store_that_takes_bigfix(ctx):
    """pcp_dfs_device one step at a time with the 10-bit-cell kernel running each node (a caller that set force_path = 1 gets it,
    pcp_api.hip): the kernel must propagate the row on TOP of the stack (it used to run row 0).  Whole search == the
    oracle's search in every counter."""
    V = 10
    props = M.nqueens_props(V)
    lb0, ub0 = np.ones(V, np.int32), np.full(V, V, np.int32)
    om = orc.OracleModel(V, props)
    ctx.set_model(V, props)
    ctx.set_hull(1, V)
    for k, v in {**DEFAULTS, "global_dom": 2, "neq_path": 0, "force_path": 1}.items():
        ctx.set_option(k, v)
    ss1, _, _, sol1 = om.search(lb0, ub0, all_solutions=False)
    one = ctx.dfs_device(lb0, ub0, 100000, capacity=256, stop_on_solution=True, chunk=64)
    assert ctx.last_plan()["path"] == 2
    assert (one["nodes"], one["solutions"], one["failed"], one["error"]) == (ss1["num_nodes"], ss1["num_solution"], ss1["num_failed_node"], 0)
    assert np.array_equal(one["first_solution"], sol1)
    ssa, _, _, _ = om.search(lb0, ub0, all_solutions=True, node_limit=600)
    al = ctx.dfs_device(lb0, ub0, 100000, capacity=256, stop_on_solution=False, node_limit=600, chunk=97)
    assert (al["nodes"], al["solutions"], al["failed"]) == (ssa["num_nodes"], ssa["num_solution"], ssa["num_failed_node"])
    for k, v in DEFAULTS.items():
        ctx.set_option(k, v)
    ctx.set_model(V, props)
