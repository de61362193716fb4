"""pcp_dfs_forest_device_set: the reference's search loop over FDSpace (IntervalSet<i32> domains, VStoreTrail) on the device — one
tree per workgroup, the current node in LDS, backtracking by an undo trail.  Against the oracle's DFS over FDSpace
(orc_search_set: OneSolution / AllSolution<Propagation<Brancher<FirstSmallestVar, MiddleVal, BinarySplit>>>, search/mod.rs:45-52):
  * one tree rooted at the root IS the reference's search: nodes, failures, solutions, and in one-solution mode the first solution
    and the node count up to it (exact left-first order);
  * a forest rooted at the open nodes of a breadth-first expansion: expansion + forest = the complete tree, counter for counter;
  * models of mixed propagator kinds (the general sweep and rounds, not the all-XNeqY shortcuts);
  * launches of a few nodes each (the tree is persisted and resumed), the node limit, a trail that is too small."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M

from util import random_csp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import pcp_amd.engine as E
    return E.Context(0)


def root_bits(lb0, ub0, sw, base):
    return M.interval_bits(np.asarray(lb0), np.asarray(ub0), sw, base)[None]


def nqueens(ctx, n):
    props = M.nqueens_props(n)
    sw = (n + 63) // 64
    ctx.set_model(n, props, set_words=sw)
    ctx.set_hull(1, n)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    return props, sw, lb0, ub0


@pytest.mark.parametrize("n,steps", [(6, 1000), (8, 7), (9, 64), (10, 256)])
def test_one_tree_is_the_reference_search(ctx, n, steps):
    props, sw, lb0, ub0 = nqueens(ctx, n)
    ss, _, _, _ = orc.OracleModel(n, props).search_set(lb0, ub0, sw, 1, all_solutions=True)
    r = ctx.dfs_forest_set(root_bits(lb0, ub0, sw, 1), steps_per_launch=steps)
    assert r["error"] == 0 and r["finished_trees"] == 1
    assert (r["solutions"], r["nodes"], r["failed"]) == (ss["num_solution"], ss["num_nodes"], ss["num_failed_node"])
    assert r["total_nodes"] == r["nodes"]
    # one solution: the reference's first solution after the reference's number of nodes
    ss1, _, _, sol1 = orc.OracleModel(n, props).search_set(lb0, ub0, sw, 1)
    one = ctx.dfs_forest_set(root_bits(lb0, ub0, sw, 1), stop_on_solution=True, steps_per_launch=steps)
    assert one["stopped"] and one["solutions"] == 1
    assert one["nodes"] == ss1["num_nodes"] and one["failed"] == ss1["num_failed_node"]
    assert np.array_equal(one["first_solution"], sol1)


def test_a_wide_hull_uses_several_words(ctx):
    """n = 70: two words per set, the branch value crosses the word boundary."""
    n = 70
    props, sw, lb0, ub0 = nqueens(ctx, n)
    assert sw == 2
    ss1, _, _, sol1 = orc.OracleModel(n, props).search_set(lb0, ub0, sw, 1)
    one = ctx.dfs_forest_set(root_bits(lb0, ub0, sw, 1), stop_on_solution=True, steps_per_launch=50)
    assert one["error"] == 0 and one["solutions"] == 1
    assert one["nodes"] == ss1["num_nodes"] and one["failed"] == ss1["num_failed_node"]
    assert np.array_equal(one["first_solution"], sol1)


@pytest.mark.parametrize("n,rounds", [(8, 3), (9, 5), (10, 6)])
def test_forest_below_a_frontier_completes_the_tree(ctx, n, rounds):
    from pcp_amd.search_device import DeviceSearch
    props, sw, lb0, ub0 = nqueens(ctx, n)
    ss, _, _, _ = orc.OracleModel(n, props).search_set(lb0, ub0, sw, 1, all_solutions=True)
    ds = DeviceSearch(ctx, batch=4096, capacity=8192, implicit=True)
    ds.reset(lb0, ub0, 1)
    for _ in range(rounds):  # breadth-first: every open node of a level in one round
        if ds.advance(all_solutions=True, max_rounds=1, keep_solutions=0):
            break
    ds.compact()
    k = ds.size
    assert k > 1
    roots = ds.bits[:k].clone()
    st = ds.stats
    r = ctx.dfs_forest_set(roots, steps_per_launch=40)
    assert r["error"] == 0 and r["finished_trees"] == k
    assert st.num_solution + r["solutions"] == ss["num_solution"]
    assert st.num_nodes + r["nodes"] == ss["num_nodes"]
    assert st.num_failed_node + r["failed"] == ss["num_failed_node"]
    assert (r["per_tree"][:, 0] > 0).all()


@pytest.mark.parametrize("seed", range(8))
def test_mixed_kinds_one_tree(ctx, seed):
    """Random CSPs over every propagator kind the set mode has (XNeqY, XEqY, XLessY, the ternary kinds): the general sweep, the
    FIFO-deduplicated rounds and range removals all go through the trail."""
    rng = np.random.default_rng(7100 + seed)
    V = int(rng.integers(5, 9))
    hi = int(rng.integers(4, 8))
    kinds = [M.NEQ, M.EQ, M.LT, M.LT3, M.GT3, M.EQ3]  # (no XEqYMulZ over sets)
    props, _, _, _ = random_csp(7200 + seed, V, int(rng.integers(6, 14)), planted=bool(seed & 1), dom=(0, hi), kinds=kinds)
    lb0, ub0 = np.zeros(V, np.int32), np.full(V, hi, np.int32)
    om = orc.OracleModel(V, props)
    ss, _, _, _ = om.search_set(lb0, ub0, 1, 0, all_solutions=True)
    ctx.set_model(V, props, set_words=1)
    ctx.set_hull(0, hi)
    r = ctx.dfs_forest_set(root_bits(lb0, ub0, 1, 0), steps_per_launch=int(rng.integers(3, 40)))
    assert r["error"] == 0
    assert (r["solutions"], r["nodes"], r["failed"]) == (ss["num_solution"], ss["num_nodes"], ss["num_failed_node"]), (seed, ss)


def test_node_limit_is_exact_and_the_search_resumes(ctx):
    n = 9
    props, sw, lb0, ub0 = nqueens(ctx, n)
    ss, _, _, _ = orc.OracleModel(n, props).search_set(lb0, ub0, sw, 1, all_solutions=True)
    K = 57
    assert ss["num_nodes"] > K
    ssk, _, _, _ = orc.OracleModel(n, props).search_set(lb0, ub0, sw, 1, all_solutions=True, node_limit=K)
    r = ctx.dfs_forest_set(root_bits(lb0, ub0, sw, 1), node_limit=K, steps_per_launch=16)
    assert r["stopped"] and r["nodes"] == K and r["total_nodes"] == K and r["finished_trees"] == 0
    assert (r["solutions"], r["failed"]) == (ssk["num_solution"], ssk["num_failed_node"])


def test_a_full_trail_is_reported(ctx):
    n = 8
    props, sw, lb0, ub0 = nqueens(ctx, n)
    r = ctx.dfs_forest_set(root_bits(lb0, ub0, sw, 1), trail_capacity=8, steps_per_launch=64)
    assert r["error"] == 4 and r["stopped"]


def test_contract(ctx):
    import pcp_amd.engine as E
    n = 6
    ctx.set_model(n, M.nqueens_props(n))  # an interval-mode model
    ctx.set_hull(1, n)
    ctx.set_words = 1  # (what the binding would size its buffers with)
    try:
        with pytest.raises(E.PcpError):
            ctx.dfs_forest_set(np.zeros((1, n, 1), np.uint64))
    finally:
        ctx.set_words = 0


@pytest.mark.parametrize("n,trees", [(8, 4), (9, 16), (10, 64)])
def test_forest_search_driver_single_and_two_ranks(ctx, n, trees):
    """pcp_amd.search_forest.forest_search_set: expansion + forest = the oracle's complete FDSpace tree; two ranks (run one after the
    other here) take alternate open nodes and their counters add up to the same tree."""
    from pcp_amd.search_forest import forest_search_set
    props, sw, lb0, ub0 = nqueens(ctx, n)
    ss, _, _, _ = orc.OracleModel(n, props).search_set(lb0, ub0, sw, 1, all_solutions=True)
    want = (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
    one = forest_search_set(ctx, lb0, ub0, 1, n_trees=trees, steps_per_launch=32)
    assert one["error"] == 0 and (one["nodes"], one["solutions"], one["failed"]) == want
    parts = [forest_search_set(ctx, lb0, ub0, 1, n_trees=trees, steps_per_launch=32, rank=r, world=2) for r in range(2)]
    assert all(p["error"] == 0 for p in parts) and all(p["trees"] > 0 for p in parts)
    assert tuple(sum(p[k] for p in parts) for k in ("nodes", "solutions", "failed")) == want


@pytest.mark.parametrize("n,rounds,steps", [(9, 2, 6), (10, 3, 10), (11, 2, 25)])
def test_finished_trees_take_work_from_the_others(ctx, n, rounds, steps):
    """pcp_dfs_forest_split_set: with few trees of very different sizes and short launches, finished trees take over the oldest open right
    branch of the others; the union is still exactly the oracle's tree, and it takes fewer launches than without re-balancing."""
    from pcp_amd.search_device import DeviceSearch
    props, sw, lb0, ub0 = nqueens(ctx, n)
    ss, _, _, _ = orc.OracleModel(n, props).search_set(lb0, ub0, sw, 1, all_solutions=True)
    ds = DeviceSearch(ctx, batch=4096, capacity=8192, implicit=True)
    ds.reset(lb0, ub0, 1)
    for _ in range(rounds):
        if ds.advance(all_solutions=True, max_rounds=1, keep_solutions=0):
            break
    ds.compact()
    k = ds.size
    roots, st = ds.bits[:k].clone(), ds.stats
    want = (ss["num_nodes"] - st.num_nodes, ss["num_solution"] - st.num_solution, ss["num_failed_node"] - st.num_failed_node)
    info = {}
    r = ctx.dfs_forest_set(roots, steps_per_launch=steps, info=info)
    assert r["error"] == 0 and (r["nodes"], r["solutions"], r["failed"]) == want
    assert info["splits"] > 0
    plain = ctx.dfs_forest_set(roots, steps_per_launch=steps, rebalance=False)
    assert (plain["nodes"], plain["solutions"], plain["failed"]) == want and plain["splits"] == 0
    assert r["launches"] < plain["launches"]
