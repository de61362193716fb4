"""pcp_dfs_forest_device: the in-kernel search loop of all-XNeqY models (pcp_neq.hip, DFS = true) on many subtrees at once, one workgroup per
tree, each exactly a pcp_dfs_device instance.  Against the oracle's DFS over Interval<i32> stores (orc_search: the reference's default
engine, search/mod.rs:45-52): one tree = the reference's search; the breadth-first expansion plus a forest below its open nodes = the
complete tree, counter for counter; two ranks taking alternate open nodes add up to the same tree."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import pcp_amd.engine as E
    return E.Context(0)


def nqueens(ctx, n):
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    ctx.set_hull(1, n)
    return props, np.ones(n, np.int32), np.full(n, n, np.int32)


def oracle_tree(n, props, lb0, ub0):
    ss, _, _, _ = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=True)
    return ss["num_nodes"], ss["num_solution"], ss["num_failed_node"]


@pytest.mark.parametrize("n,steps", [(6, 1000), (8, 5), (10, 64)])
def test_one_tree_through_the_forest_entry(ctx, n, steps):
    props, lb0, ub0 = nqueens(ctx, n)
    want = oracle_tree(n, props, lb0, ub0)
    r = ctx.dfs_forest(lb0[None], ub0[None], steps_per_launch=steps, capacity=256)
    assert ctx.last_plan()["path"] == 1 and ctx.last_plan()["grid"] == 1
    assert r["error"] == 0 and r["open"] == 0
    assert (r["nodes"], r["solutions"], r["failed"]) == want
    # one solution: the reference's first solution after the reference's number of nodes
    ss1, _, _, sol1 = orc.OracleModel(n, props).search(lb0, ub0)
    one = ctx.dfs_forest(lb0[None], ub0[None], stop_on_solution=True, steps_per_launch=steps, capacity=256, want_solution=True)
    assert one["solutions"] == 1 and one["nodes"] == ss1["num_nodes"] and np.array_equal(one["first_solutions"][0], sol1)


@pytest.mark.parametrize("n,trees", [(8, 8), (10, 64), (11, 300)])
def test_expansion_plus_forest_is_the_complete_tree(ctx, n, trees):
    from pcp_amd.search_forest import forest_search
    props, lb0, ub0 = nqueens(ctx, n)
    want = oracle_tree(n, props, lb0, ub0)
    one = forest_search(ctx, lb0, ub0, n_trees=trees, steps_per_launch=16, capacity=256)
    assert one["error"] == 0 and one["trees"] >= min(trees, 2)
    assert (one["nodes"], one["solutions"], one["failed"]) == want
    parts = [forest_search(ctx, lb0, ub0, n_trees=trees, steps_per_launch=16, capacity=256, rank=r, world=2) for r in range(2)]
    assert all(p["error"] == 0 and p["trees"] > 0 for p in parts)
    assert tuple(sum(p[k] for p in parts) for k in ("nodes", "solutions", "failed")) == want


@pytest.mark.parametrize("block,wgs", [(128, 8), (256, 4), (512, 2)])
def test_tree_shapes(ctx, block, wgs):
    """Threads per tree and trees per CU (options neq_dfs_block / neq_dfs_wgs; the defaults depend on the number of trees): the complete tree
    of n = 10 from 64 roots in every shape, failures, solutions and jump windows included."""
    from pcp_amd.search_forest import forest_search
    n = 10
    props, lb0, ub0 = nqueens(ctx, n)
    want = oracle_tree(n, props, lb0, ub0)
    ctx.set_option("neq_dfs_block", block); ctx.set_option("neq_dfs_wgs", wgs)
    try:
        one = forest_search(ctx, lb0, ub0, n_trees=64, steps_per_launch=16, capacity=256)
        assert ctx.last_plan()["block"] == block
    finally:
        ctx.set_option("neq_dfs_block", 0); ctx.set_option("neq_dfs_wgs", 0)
    assert one["error"] == 0 and (one["nodes"], one["solutions"], one["failed"]) == want


def test_a_node_budget_stops_the_launches(ctx):
    from pcp_amd.search_forest import forest_search
    n = 12
    props, lb0, ub0 = nqueens(ctx, n)
    r = forest_search(ctx, lb0, ub0, node_limit=2000, n_trees=32, steps_per_launch=8, capacity=256)
    assert r["error"] == 0 and 2000 <= r["nodes"] < 2000 + 32 * 8 + 64


def test_contract(ctx):
    import pcp_amd.engine as E
    # a model with another kind: the forest entry refuses it (pcp_dfs_device runs it, one tree)
    V = 4
    props = M.lower_units([M.XNeqY(M.Identity(0), M.Identity(1)), M.XLessY(M.Identity(2), M.Identity(3))], V)
    ctx.set_model(V, props)
    ctx.set_hull(0, 5)
    with pytest.raises(E.PcpError):
        ctx.dfs_forest(np.zeros((2, V), np.int32), np.full((2, V), 5, np.int32), capacity=16)


@pytest.mark.parametrize("n,trees,steps", [(10, 4, 6), (11, 6, 12), (12, 3, 40)])
def test_finished_trees_take_the_oldest_open_node_of_the_others(ctx, n, trees, steps):
    """dfs_forest's refill between launches (the bottom row of a donor's stack moves to a finished tree): with few trees of very
    different sizes and short launches the union is still exactly the oracle's tree, and it takes fewer launches than without."""
    from pcp_amd.search_forest import seed_roots_interval
    props, lb0, ub0 = nqueens(ctx, n)
    want = oracle_tree(n, props, lb0, ub0)
    rl, ru, st = seed_roots_interval(ctx, lb0, ub0, trees)
    rest = (want[0] - st.num_nodes, want[1] - st.num_solution, want[2] - st.num_failed_node)
    r = ctx.dfs_forest(rl, ru, steps_per_launch=steps, capacity=256)
    assert r["error"] == 0 and r["open"] == 0 and (r["nodes"], r["solutions"], r["failed"]) == rest
    assert r["steals"] > 0
    plain = ctx.dfs_forest(rl, ru, steps_per_launch=steps, capacity=256, rebalance=False)
    assert (plain["nodes"], plain["solutions"], plain["failed"]) == rest and plain["steals"] == 0
    assert r["launches"] < plain["launches"]


@pytest.mark.parametrize("n,trees,steps", [(10, 4, 50), (12, 16, 200)])
def test_stacks_grow_on_demand(ctx, n, trees, steps):
    """Stacks that start far too small (2 rows per tree) are doubled whenever a tree fills its rows (ForestStacks.grow): the forest
    still visits exactly the oracle's tree; a ceiling that is too low is reported as error 1, never hidden."""
    from pcp_amd.search_forest import seed_roots_interval
    props, lb0, ub0 = nqueens(ctx, n)
    want = oracle_tree(n, props, lb0, ub0)
    rl, ru, st = seed_roots_interval(ctx, lb0, ub0, trees)
    rest = (want[0] - st.num_nodes, want[1] - st.num_solution, want[2] - st.num_failed_node)
    info = {}
    r = ctx.dfs_forest(rl, ru, steps_per_launch=steps, capacity=2, max_capacity=256, info=info)
    assert r["error"] == 0 and r["open"] == 0 and (r["nodes"], r["solutions"], r["failed"]) == rest
    assert info["grown"] >= 1 and 2 < info["capacity"] <= 256
    low = ctx.dfs_forest(rl, ru, steps_per_launch=steps, capacity=2, max_capacity=2, rebalance=False)
    assert low["error"] == 1 and low["open"] > 0
