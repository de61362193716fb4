"""BASELINE config 5 at its own size: the device-side search forests on N-queens n = 1000, NODE FOR NODE against
the oracle's DFS from each subtree root — the launch shapes bench.py's `forest_nps` / `setforest_nps` legs and the config-5 engines run
(interval: neqfix_kernel<false, true, true, 1>, 256-thread trees, 32-bit cells, stack rows in HBM; sets: setdfs_kernel, 16 words per
variable, an undo trail), asserted through pcp_last_plan so that the tests cannot drift off the measured shape.  The roots are the open
nodes of the breadth-first expansion the search drivers start from (search_forest.seed_roots_interval / seed_roots).
oracle/forest_check.py holds the comparison (bench.py runs the same check inside its forest legs)."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import forest_check as FC
from pcp_amd import model as M

pytestmark = pytest.mark.gpu
N = 1000


@pytest.fixture(scope="module")
def env():
    import pcp_amd.engine as E
    ctx = E.Context(0)
    props = M.nqueens_props(N)
    om = orc.OracleModel(N, props)
    yield ctx, om, props
    ctx.close()


def test_interval_forest_node_for_node(env):
    """8 subtrees x 12 nodes, one node per launch and twelve in one launch (the resumed left children never leave LDS)."""
    from pcp_amd.search_forest import seed_roots_interval
    ctx, om, props = env
    ctx.set_model(N, props)
    ctx.set_hull(1, N)
    lb0, ub0 = np.ones(N, np.int32), np.full(N, N, np.int32)
    rl, ru, st = seed_roots_interval(ctx, lb0, ub0, 8)
    rl, ru = rl.cpu().numpy(), ru.cpu().numpy()
    assert rl.shape[0] >= 8
    checked = FC.check_interval_forest(ctx, om, rl[:8], ru[:8], K=12)
    assert checked == 2 * 8 * 12
    # ... and without the per-row hint words (pcp_dfs_state.dirty == NULL: every popped row is propagated from all its assigned queens)
    assert FC.check_interval_forest(ctx, om, rl[:2], ru[:2], K=6, hints=False) == 2 * 2 * 6


def test_interval_forest_narrow_trees(env):
    """The shape a forest of 4096 or more trees takes (128-thread trees, eight to a CU: pcp_api.hip launch_neq_dfs), forced here on 8 trees by
    option, node for node as above."""
    from pcp_amd.search_forest import seed_roots_interval
    ctx, om, props = env
    ctx.set_model(N, props)
    ctx.set_hull(1, N)
    lb0, ub0 = np.ones(N, np.int32), np.full(N, N, np.int32)
    rl, ru, st = seed_roots_interval(ctx, lb0, ub0, 8)
    ctx.set_option("neq_dfs_block", 128); ctx.set_option("neq_dfs_wgs", 8)
    try:
        checked = FC.check_interval_forest(ctx, om, rl[:8].cpu().numpy(), ru[:8].cpu().numpy(), K=10, expect_block=128)
    finally:
        ctx.set_option("neq_dfs_block", 0); ctx.set_option("neq_dfs_wgs", 0)
    assert checked == 2 * 8 * 10


def test_interval_forest_deep_roots(env):
    """Roots further down: the top rows of the stack after a 300-node dive (about 30 queens assigned; propagation cascades and failures
    are common there), 4 subtrees x 10 nodes."""
    ctx, om, props = env
    ctx.set_model(N, props)
    ctx.set_hull(1, N)
    import ctypes as C
    import torch
    import pcp_amd.engine as E
    lb0, ub0 = np.ones(N, np.int32), np.full(N, N, np.int32)
    dev = torch.device("cuda", 0)
    cap = 512
    lb = torch.zeros((1, cap, N), dtype=torch.int32, device=dev); ub = torch.zeros_like(lb)
    lb[0, 0] = torch.from_numpy(lb0).to(dev); ub[0, 0] = torch.from_numpy(ub0).to(dev)
    sp = torch.ones(1, dtype=torch.int32, device=dev); stop = torch.zeros(1, dtype=torch.int32, device=dev)
    status = torch.zeros((1, cap), dtype=torch.uint8, device=dev); counters = torch.zeros((1, 5), dtype=torch.int64, device=dev)
    st = E.DfsState(lb.data_ptr(), ub.data_ptr(), cap, sp.data_ptr(), stop.data_ptr(), status.data_ptr(), counters.data_ptr(), None)
    ctx._check(ctx._L.pcp_dfs_forest_device(ctx._h, C.byref(st), 1, 300, 0, 0, None))
    torch.cuda.synchronize()
    assert int(counters[0, 0].item()) == 300 and int(counters[0, 3].item()) == 0
    k = int(sp.item())
    assert k >= 8
    pick = [k - 1, k - 2, k // 2, 1]  # the node about to be visited, its sibling's right branch, older open nodes
    rl, ru = lb[0, pick].cpu().numpy(), ub[0, pick].cpu().numpy()
    checked = FC.check_interval_forest(ctx, om, rl, ru, K=10)
    assert checked == 2 * 4 * 10


def test_set_forest_node_for_node(env):
    """FDSpace (IntervalSet<i32>, the reference's default): 6 subtrees x 8 nodes, 16 words per variable, the trail undone on failures."""
    from pcp_amd.search_forest import seed_roots
    ctx, om, props = env
    sw = (N + 63) // 64
    ctx.set_model(N, props, set_words=sw)
    ctx.set_hull(1, N)
    lb0, ub0 = np.ones(N, np.int32), np.full(N, N, np.int32)
    roots, st = seed_roots(ctx, lb0, ub0, 1, 8)
    rb = roots.cpu().numpy().view(np.uint64).reshape(-1, N, sw)
    assert rb.shape[0] >= 6
    checked = FC.check_set_forest(ctx, om, rb[:6], 1, K=8)
    assert checked == 2 * 6 * 8
    ctx.set_model(N, props)
    ctx.set_hull(1, N)


def test_set_forest_deep_root(env):
    """A root 400 nodes down the FDSpace dive (its sets are full of holes left by ~40 assigned queens): the node a one-tree forest is
    about to visit after 400 steps, handed to a fresh forest as a root, 2 x 8 nodes against the oracle's DFS from the same sets."""
    import ctypes as C
    import torch
    import pcp_amd.engine as E
    ctx, om, props = env
    sw = (N + 63) // 64
    ctx.set_model(N, props, set_words=sw)
    ctx.set_hull(1, N)
    lb0, ub0 = np.ones(N, np.int32), np.full(N, N, np.int32)
    dev = torch.device("cuda", 0)
    bound = N * sw * 64 + 16
    bits = torch.from_numpy(M.interval_bits(lb0, ub0, sw, 1)[None].view(np.int64)).to(dev).clone()
    tree = torch.zeros((1, 4), dtype=torch.int32, device=dev); tree[:, 2] = -1
    levels = torch.zeros((1, 1 << 12, 4), dtype=torch.int32, device=dev); trail = torch.zeros((1, bound, 4), dtype=torch.int32, device=dev)
    counters = torch.zeros((1, 4), dtype=torch.int64, device=dev); glob = torch.zeros(4, dtype=torch.int64, device=dev)
    st = E.ForestState(1, 1 << 12, bound, 0, bits.data_ptr(), tree.data_ptr(), levels.data_ptr(), trail.data_ptr(), counters.data_ptr(),
                       glob.data_ptr(), glob.data_ptr() + 8, None, None)
    roots = []
    for steps in (400, 37):
        ctx._check(ctx._L.pcp_dfs_forest_device_set(ctx._h, C.byref(st), steps, 0, 0, None))
        torch.cuda.synchronize()
        roots.append(bits.cpu().numpy().view(np.uint64).reshape(N, sw).copy())
    assert int(counters[0, 0].item()) == 437 and int(counters[0, 3].item()) == 0
    checked = FC.check_set_forest(ctx, om, np.stack(roots), 1, K=8)
    assert checked == 2 * 2 * 8
    ctx.set_model(N, props)
    ctx.set_hull(1, N)
