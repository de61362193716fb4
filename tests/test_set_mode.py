"""Set mode: IntervalSet<i32> domains (VStoreSet — the reference's default FDSpace, variable/mod.rs:38, search/mod.rs:41-43).

CPU part: the oracle's IntervalSet restatement against hand-derived set algebra and against the answers the reference's
own tests hold on FDSpace (all_solution.rs:70, one_solution.rs:121-128, stop_node.rs:83-104) — the only pins the reference
offers for this mode (no reference test observes an IntervalSet after propagation: SURVEY.md §8c).
GPU part (-m gpu): the HIP set-mode engine (pcp_set.hip, through the C-ABI) bit-exact against that oracle."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
from pcp_amd import search as S

from util import random_active, random_csp, splitmix64

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def bits_of(values, sw=1, base=0):
    w = np.zeros(sw, np.uint64)
    for v in values:
        w[(v - base) >> 6] |= np.uint64(1) << np.uint64((v - base) & 63)
    return w


def values_of(w, base=0):
    return [base + 64 * k + b for k, x in enumerate(w) for b in range(64) if (int(x) >> b) & 1]


# ------------------------------------------------------------------------------------------------------ oracle, CPU
def test_intervalset_algebra():
    """IntervalSet ops as sets of integers (crate intervallum's published behaviour): difference punches holes."""
    s = bits_of([1, 2, 3, 4, 5, 9, 10])
    assert values_of(orc.set_op("difference", s, 0, 3)[0]) == [1, 2, 4, 5, 9, 10]      # interior value: a hole (Inner event)
    assert values_of(orc.set_op("difference", s, 0, 1)[0]) == [2, 3, 4, 5, 9, 10]
    assert values_of(orc.set_op("difference", s, 0, 7)[0]) == [1, 2, 3, 4, 5, 9, 10]   # absent value: unchanged
    assert values_of(orc.set_op("shrink_left", s, 0, 4)[0]) == [4, 5, 9, 10]
    assert values_of(orc.set_op("shrink_left", s, 0, 6)[0]) == [9, 10]                 # the bound snaps to a member
    assert values_of(orc.set_op("shrink_right", s, 0, 8)[0]) == [1, 2, 3, 4, 5]
    assert values_of(orc.set_op("shift", s, 0, 3)[0]) == [4, 5, 6, 7, 8, 12, 13]
    t = bits_of([3, 6, 7, 8, 9])
    assert values_of(orc.set_op("intersection", s, 0, 0, t)[0]) == [3, 9]
    assert orc.set_op("is_disjoint", s, 0, 0, bits_of([6, 7, 8]))[1] is True           # inside the hull, still disjoint
    assert orc.set_op("is_disjoint", s, 0, 0, t)[1] is False
    assert orc.set_op("is_subset", bits_of([2, 9]), 0, 0, s)[1] is True
    assert orc.set_op("is_subset", bits_of([2, 6]), 0, 0, s)[1] is False
    # two words, negative base
    s2 = bits_of([-3, 60, 61, 70], sw=2, base=-10)
    assert values_of(orc.set_op("difference", s2, -10, 61)[0], -10) == [-3, 60, 70]
    assert values_of(orc.set_op("shrink_right", s2, -10, 60)[0], -10) == [-3, 60]


def test_xneqy_punches_interior_holes_and_raises_inner():
    """x_neq_y.rs:82-93 over IntervalSet: x = {5}, y = [0,10] -> y loses 5 although it is no bound; on Interval<i32> it does not
    (x_neq_y.rs:128).  XLessY (Bound subscriber) on the same store is untouched."""
    props = M.lower_units([M.XNeqY(M.Identity(0), M.Identity(1)), M.XLessY(M.Identity(1), M.Identity(2))], 3)
    om = orc.OracleModel(3, props)
    bits = M.interval_bits(np.array([5, 0, 0]), np.array([5, 10, 20]), 1, 0)
    lb, ub, b, act, st, _ = om.consistency_set(bits[None], 0)
    assert values_of(b[0, 1]) == [0, 1, 2, 3, 4, 6, 7, 8, 9, 10] and (lb[0, 1], ub[0, 1]) == (0, 10)
    assert values_of(b[0, 2]) == list(range(1, 21)) and st[0] == M.UNKNOWN
    assert int(act[0, 0]) == 0b10  # x != y is entailed now (the sets are disjoint), y < z is not
    li, ui, _, sti, _ = om.consistency(np.array([[5, 0, 0]], np.int32), np.array([[5, 10, 20]], np.int32))
    assert (li[0, 1], ui[0, 1]) == (0, 10) and sti[0] == M.UNKNOWN


def test_fdspace_answers_of_the_reference_tests():
    """The reference's search tests run on FDSpace = IntervalSet domains: all-solution counts (all_solution.rs:70), first
    solution statuses (one_solution.rs:121-128), StopNode (stop_node.rs:83-104)."""
    g = json.load(open(os.path.join(GOLDEN, "engine_kats.json")))["search"]
    counts = g["all_solutions"]["counts"]
    for n, want in enumerate(counts[:8], start=1):
        props = M.nqueens_props(n) if n > 1 else M.lower_units([], 1)
        om = orc.OracleModel(n, props)
        ss, _, _, _ = om.search_set(np.ones(n, np.int32), np.full(n, n, np.int32), 1, 0, all_solutions=True)
        assert ss["num_solution"] == want, (n, ss)
    for n_s, want_status in g["one_solution"]["status"].items():
        n = int(n_s)
        props = M.nqueens_props(n) if n > 1 else M.lower_units([], 1)
        ss, _, _, sol = orc.OracleModel(n, props).search_set(np.ones(n, np.int32), np.full(n, n, np.int32), 1, 0)
        assert (ss["num_solution"] == 1) == (want_status == "Satisfiable"), n
        if ss["num_solution"]:
            assert len(set(sol)) == n and len({int(sol[i]) + i for i in range(n)}) == n and len({int(sol[i]) - i for i in range(n)}) == n
    sn = g["stop_node"]
    ss, _, _, _ = orc.OracleModel(sn["n"], M.nqueens_props(sn["n"])).search_set(np.ones(sn["n"], np.int32), np.full(sn["n"], sn["n"], np.int32), 1, 0,
                                                                               all_solutions=True, node_limit=sn["limit"])
    assert ss["num_nodes"] == sn["expect_nodes"]


def test_set_and_interval_mode_differ_in_nodes_not_in_solutions():
    """Search is complete in both modes (same solutions); set mode prunes more per node, so the trees differ."""
    n = 8
    om = orc.OracleModel(n, M.nqueens_props(n))
    a, _, _, _ = om.search_set(np.ones(n, np.int32), np.full(n, n, np.int32), 1, 0, all_solutions=True)
    b, _, _, _ = om.search(np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True)
    assert a["num_solution"] == b["num_solution"] == 92 and a["num_nodes"] != b["num_nodes"]


def random_sets(seed, lb, ub, n_nodes, sw, base, sol=None, p_keep=0.7):
    """Random subsets of [lb, ub] per variable (holes included); with `sol` every set contains its planted value."""
    rng = splitmix64(seed)
    V = lb.shape[0]
    full = M.interval_bits(lb, ub, sw, base)
    out = np.zeros((n_nodes, V, sw), np.uint64)
    for n in range(n_nodes):
        r = rng.integers(0, 1 << 62, size=(V, sw), dtype=np.int64).astype(np.uint64)
        r2 = rng.integers(0, 1 << 62, size=(V, sw), dtype=np.int64).astype(np.uint64)
        mask = (r | r2 | (r << np.uint64(2))) if p_keep > 0.5 else (r & r2)
        cand = full & mask
        keep_all = rng.random(V) < 0.4
        cand[keep_all] = full[keep_all]
        if sol is not None:
            cand |= M.interval_bits(sol, sol, sw, base)
        empty = ~cand.any(axis=1)
        cand[empty] = full[empty]
        out[n] = cand
    return out


# ------------------------------------------------------------------------------------------------------ HIP engine, GPU
@pytest.fixture(scope="module")
def ctx():
    import pcp_amd.engine as E
    c = E.Context(0)
    yield c
    c.close()


def assert_set_parity(ref, got, what, check_active=True):
    rlb, rub, rbits, ract, rst = ref
    glb, gub, gbits, gact, gst = got
    assert np.array_equal(rst, gst), f"{what}: status differs at {np.nonzero(rst != gst)[0][:10]} ref={rst[rst != gst][:10]} got={gst[rst != gst][:10]}"
    ok = rst != 0
    assert np.array_equal(rbits[ok], gbits[ok]), f"{what}: sets differ at nodes {np.nonzero((rbits != gbits).any(axis=(1, 2)) & ok)[0][:10]}"
    assert np.array_equal(rlb[ok], glb[ok]) and np.array_equal(rub[ok], gub[ok]), f"{what}: bounds differ"
    if check_active and gact is not None:
        assert np.array_equal(ract[ok], gact[ok]), f"{what}: active differs"


def both_set(ctx, n_vars, props, bits, base, hull, active, what):
    import pcp_amd.engine as E
    import torch
    om = orc.OracleModel(n_vars, props)
    sw = bits.shape[2]
    ctx.set_model(n_vars, props, set_words=sw)
    ctx.set_hull(*hull)
    n = bits.shape[0]
    # explicit `active` rows
    act = active if active is not None else E.full_active(n, om.n_units)
    ref = om.consistency_set(bits, base, act)
    ctx.set_option("implicit_active", 0)
    got = ctx.propagate_set(bits, act)
    ctx.set_option("implicit_active", 1)
    assert_set_parity(ref[:5], got[:5], what + " [explicit]")
    # implicit-active nodes through the device entry, rows materialised on request
    ref_i = ref if active is None else om.consistency_set(bits, base, None)
    dev = torch.device("cuda", 0)
    t_bits = torch.from_numpy(bits.view(np.int64)).to(dev)
    t_lb = torch.zeros((n, n_vars), dtype=torch.int32, device=dev)
    t_ub = torch.zeros_like(t_lb)
    t_act = torch.zeros((n, max(ctx.words, 1)), dtype=torch.int64, device=dev)
    t_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.propagate_device(n, None, None, t_lb, t_ub, None, t_act, t_st, 0, bits_in=t_bits, bits_out=t_bits)
    torch.cuda.synchronize()
    pl = ctx.last_plan()
    assert pl["set_mode"] == 1 and pl["implicit_active"] == 1
    got_i = (t_lb.cpu().numpy(), t_ub.cpu().numpy(), t_bits.cpu().numpy().view(np.uint64), t_act.cpu().numpy().view(np.uint64)[:, : ctx.words], t_st.cpu().numpy())
    assert_set_parity(ref_i[:5], got_i, what + " [implicit]")
    return ref, got


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("planted", [True, False])
def test_set_mode_random_csp(ctx, seed, planted):
    V, P, N = 30 + 9 * seed, 120 + 50 * seed, 64
    kinds = [M.NEQ, M.EQ, M.LT, M.LT3, M.GT3, M.EQ3]
    props, lb, ub, sol = random_csp(700 + seed, V, P, planted=planted, dom=(-5, 90), kinds=kinds)
    base, sw = -5, 2
    bits = random_sets(800 + seed, lb, ub, N, sw, base, sol if planted else None)
    act = random_active(900 + seed, N, P, p_off=0.15)
    ref, _ = both_set(ctx, V, props, bits, base, (-5, 90), act, f"set csp seed={seed} planted={planted}")
    if not planted:
        assert (ref[4] == 0).any()


@pytest.mark.gpu
def test_set_mode_kat_shapes(ctx):
    """The XNeqY / XEqY / XLessY known-answer inputs as one-node set-mode fixpoints, plus interior-removal cases."""
    cases = [
        (M.XNeqY(M.Identity(0), M.Identity(1)), [(5, 5), (0, 10)]),
        (M.XNeqY(M.Identity(0), M.Addition(M.Identity(1), 3)), [(5, 5), (0, 10)]),
        (M.XNeqY(M.Identity(0), M.Constant(4)), [(0, 10)]),
        (M.XEqY(M.Identity(0), M.Addition(M.Identity(1), -2)), [(0, 10), (5, 15)]),
        (M.XEqY(M.Identity(0), M.Constant(7)), [(0, 10)]),
        (M.XLessY(M.Identity(0), M.Identity(1)), [(0, 10), (0, 10)]),
        (M.XNeqY(M.Identity(0), M.Identity(1)), [(1, 1), (1, 1)]),
    ]
    for k, (unit, doms) in enumerate(cases):
        V = len(doms)
        props = M.lower_units([unit], V)
        bits = M.interval_bits(np.array([d[0] for d in doms]), np.array([d[1] for d in doms]), 1, 0)[None]
        both_set(ctx, V, props, bits, 0, (0, 20), None, f"set kat {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 20, 70])
def test_set_mode_nqueens_search_nodes(ctx, n):
    """The first nodes of the reference's DFS over FDSpace (folded inputs as sets) -> fixpoints, one launch."""
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    sw = (n + 63) // 64
    K = 150
    _, _, rec, _ = om.search_set(np.ones(n, np.int32), np.full(n, n, np.int32), sw, 1, all_solutions=True, node_limit=K, max_records=K)
    keep = rec["bits_in"].any(axis=2).all(axis=1)
    ctx.set_model(n, props, set_words=sw)
    ctx.set_hull(1, n)
    got = ctx.propagate_set(rec["bits_in"][keep], rec["active_in"][keep])
    ref = (rec["lb_out"][keep], rec["ub_out"][keep], rec["bits_out"][keep], rec["active_out"][keep], rec["status"][keep])
    assert_set_parity(ref, got[:5], f"set nqueens({n}) nodes")
    assert (got[2] != rec["bits_in"][keep]).any()
    # the same nodes as implicit-active nodes: the all-XNeqY shortcuts (only assigned variables sweep and wake, pcp_set.hip)
    ref_i = om.consistency_set(rec["bits_in"][keep], 1, None)
    got_i = ctx.propagate_set(rec["bits_in"][keep], None)
    assert ctx.last_plan()["implicit_active"] == 1
    assert_set_parity(ref_i[:5], (got_i[0], got_i[1], got_i[2], None, got_i[4]), f"set nqueens({n}) nodes implicit", check_active=False)


@pytest.mark.gpu
@pytest.mark.parametrize("n,batch", [(6, 1), (8, 1), (8, 16)])
def test_set_mode_search_reproduces_the_fdspace_tree(ctx, n, batch):
    """Whole search with every fixpoint on the GPU in set mode == the oracle's DFS over FDSpace (nodes, failures, solutions;
    all_solution.rs:70 for the counts)."""
    props = M.nqueens_props(n)
    ctx.set_model(n, props, set_words=1)
    ctx.set_hull(1, n)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    ss, _, _, sol = orc.OracleModel(n, props).search_set(lb0, ub0, 1, 1, all_solutions=True)
    st = S.dfs_set(ctx, lb0, ub0, 1, all_solutions=True, batch=batch)
    assert (st.num_solution, st.num_nodes, st.num_failed_node) == (ss["num_solution"], ss["num_nodes"], ss["num_failed_node"])
    if batch == 1:
        one = S.dfs_set(ctx, lb0, ub0, 1, all_solutions=False)
        ss1, _, _, sol1 = orc.OracleModel(n, props).search_set(lb0, ub0, 1, 1)
        assert one.num_nodes == ss1["num_nodes"] and np.array_equal(one.solutions[0], sol1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,batch,implicit", [(6, 1, True), (8, 1, False), (8, 64, True), (9, 256, True)])
def test_set_mode_device_resident_search(ctx, n, batch, implicit):
    """Stack, set-mode propagation and set-mode branching (pcp_branch_device_set: FirstSmallestVar by cardinality) on the GPU:
    the oracle's DFS over FDSpace exactly (solutions / nodes / failures; with batch 1 also the order)."""
    from pcp_amd.search_device import DeviceSearch
    props = M.nqueens_props(n)
    ctx.set_model(n, props, set_words=1)
    ctx.set_hull(1, n)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    ss, _, _, sol = orc.OracleModel(n, props).search_set(lb0, ub0, 1, 1, all_solutions=True)
    st = DeviceSearch(ctx, batch=batch, capacity=4096, implicit=implicit).run(lb0, ub0, all_solutions=True, keep_solutions=400, base=1)
    assert (st.num_solution, st.num_nodes, st.num_failed_node) == (ss["num_solution"], ss["num_nodes"], ss["num_failed_node"])
    assert len({tuple(s) for s in st.solutions}) == ss["num_solution"]
    if batch == 1:
        one = DeviceSearch(ctx, batch=1, capacity=4096, implicit=implicit).run(lb0, ub0, all_solutions=False, keep_solutions=1, base=1)
        ss1, _, _, sol1 = orc.OracleModel(n, props).search_set(lb0, ub0, 1, 1)
        assert one.num_nodes == ss1["num_nodes"] and np.array_equal(one.solutions[0], sol1)


@pytest.mark.gpu
def test_set_mode_contract(ctx):
    import pcp_amd.engine as E
    props = M.lower_units([M.XNeqY(M.Identity(0), M.Identity(1))], 2)
    ctx.set_model(2, props, set_words=1)
    bits = M.interval_bits(np.array([0, 0]), np.array([5, 5]), 1, 0)[None]
    with pytest.raises(E.PcpError) as e:  # no hull declared
        ctx.propagate_set(bits)
    assert e.value.code == -2
    ctx.set_hull(0, 100)
    with pytest.raises(E.PcpError) as e:  # hull wider than the universe
        ctx.propagate_set(bits)
    assert e.value.code == -2
    ctx.set_hull(0, 63)
    bad = bits.copy(); bad[0, 1, 0] = 0
    with pytest.raises(E.PcpError) as e:  # empty initial domain (variable/store.rs:136)
        ctx.propagate_set(bad)
    assert e.value.code == -2
    with pytest.raises(E.PcpError) as e:  # XEqYMulZ is interval-mode only
        ctx.set_model(3, M.lower_units([M.XEqYMulZ(M.Identity(0), M.Identity(1), M.Identity(2))], 3), set_words=1)
    assert e.value.code == -5
