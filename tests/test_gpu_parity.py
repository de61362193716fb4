"""Parity tests proper: the HIP engine, called through the C-ABI (pcp_amd.engine -> libpcp_hip.so), against
the CPU oracle on the same seeded inputs.  Bar: status equal on every node; on nodes that are
not False, (lb,ub) and `active` bit-exact.  Integer work: no tolerance anywhere."""
import json
import os
import sys

import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
import pcp_amd.engine as E

from util import assert_parity, planted_binary_csp, random_active, random_csp, random_nodes, unit_narrowing_prefix
from test_oracle_golden import build_unit

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = E.Context(0)
    yield c
    c.close()


def both(ctx, n_vars, props, lb, ub, active, what, **opts):
    om = orc.OracleModel(n_vars, props)
    ref = om.consistency(lb, ub, active)
    ctx.set_model(n_vars, props)
    for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1, "group_level": 1, **opts}.items():
        ctx.set_option(k, v)
    ctx.set_option("small_path", 1)
    act_in = active if active is not None else E.full_active(np.asarray(lb).reshape(-1, n_vars).shape[0], om.n_units)
    got = ctx.propagate(lb, ub, act_in)
    assert_parity(ref[:4], got[:4], what)
    if ctx.last_plan()["path"] == 4:
        # a small store took the one-wavefront-per-node kernel (pcp_small.hip): the generic kernels under the same options too
        ctx.set_option("small_path", 0)
        got_g = ctx.propagate(lb, ub, act_in)
        assert ctx.last_plan()["path"] == 0
        assert_parity(ref[:4], got_g[:4], what + " [generic kernels]")
        ctx.set_option("small_path", 1)
    # the same nodes as IMPLICIT-active nodes (domains only, every unit active on entry, liveness derived from the domains;
    # `active` rows materialised on request) under the same launch options
    ref_i = ref if active is None else om.consistency(lb, ub, None)
    got_i = ctx.propagate_implicit(lb, ub)
    assert ctx.last_plan()["implicit_active"] == 1
    assert_parity(ref_i[:4], got_i[:4], what + " [implicit]")
    if ctx.last_plan()["path"] in (1, 4):
        # an all-XNeqY model took the assignment-driven kernel (pcp_neq.hip), or a small store the one-wavefront-per-node kernel
        # (pcp_small.hip): the generic implicit kernels under the same options too
        ctx.set_option("neq_path", 0); ctx.set_option("small_path", 0)
        got_g = ctx.propagate_implicit(lb, ub)
        ctx.set_option("neq_path", 1); ctx.set_option("small_path", 1)
        assert ctx.last_plan()["path"] == 0
        assert_parity(ref_i[:4], got_g[:4], what + " [implicit, generic kernels]")
    return ref, got


def test_kat_inputs_as_one_node_fixpoints(ctx):
    """Every transcribed reference KAT input (tests/golden/propagator_kats.json) as a one-unit store."""
    data = json.load(open(os.path.join(GOLDEN, "propagator_kats.json")))
    n_cases = 0
    for s in data["suites"]:
        for c in s["cases"]:
            unit = build_unit(s, c)
            n = len(c["doms"])
            props = M.lower_units([unit], n)
            lb = np.array([[d[0] for d in c["doms"]]], np.int32)
            ub = np.array([[d[1] for d in c["doms"]]], np.int32)
            ref, got = both(ctx, n, props, lb, ub, None, f"{s['name']}#{c['n']}")
            # one propagate() that fails <=> the one-unit fixpoint is False
            assert (ref[3][0] == M.FALSE) == (not c["ok"])
            n_cases += 1
    assert n_cases >= 80


def test_engine_answers(ctx):
    """The reference's (commented-out) engine tests: chained X1<...<Xn on [1,10] (propagation/store.rs:362-385)."""
    g = json.load(open(os.path.join(GOLDEN, "engine_kats.json")))["engine"]
    K = {"F": 0, "T": 1, "U": 2}
    for n, exp in g["chained_lt"]["cases"]:
        vs, cs = M.chained_lt(n)
        lb, ub = vs.bounds()
        props = cs.lower(n)
        ref, got = both(ctx, n, props, lb[None], ub[None], None, f"chained_lt({n})")
        assert got[3][0] == K[exp]


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("planted", [True, False])
def test_random_csp_batch(ctx, seed, planted):
    V, P, N = 40 + 13 * seed, 150 + 60 * seed, 96
    props, lb, ub, sol = random_csp(1000 + seed, V, P, planted=planted)
    L, U = random_nodes(2000 + seed, lb, ub, N, sol if planted else None)
    act = random_active(3000 + seed, N, P, p_off=0.15)
    both(ctx, V, props, L, U, act, f"csp seed={seed} planted={planted}")


@pytest.mark.parametrize("seed", range(4))
def test_random_csp_mixed_statuses(ctx, seed):
    """Nodes that fail, stay Unknown and become True in the same launch."""
    V, P, N = 40 + 13 * seed, 100 + 40 * seed, 128
    props, lb, ub, sol = random_csp(1000 + seed, V, P, planted=True, dom=(0, 12))
    L, U = random_nodes(2000 + seed, lb, ub, N, None, p_narrow=0.05)
    act = random_active(3000 + seed, N, P, p_off=0.15)
    ref, _ = both(ctx, V, props, L, U, act, f"mixed seed={seed}")
    assert (ref[3] == 0).any() and (ref[3] == 2).any()
    props, lb, ub, sol = random_csp(5 + seed, 12, 10, planted=True, dom=(0, 3))
    L, U = random_nodes(6 + seed, lb, ub, 200, sol, p_narrow=0.9)
    ref, got = both(ctx, 12, props, L, U, None, f"tiny seed={seed}")
    assert (ref[3] == 1).any() and (ref[3] == 2).any()
    assert got[4]["nodes"] == 200 and got[4]["steps"] + got[4]["steps3"] >= 200 * 10


@pytest.mark.parametrize("opts", [
    {"nodes_per_block": 1}, {"nodes_per_block": 3}, {"nodes_per_block": 32}, {"block_threads": 256}, {"block_threads": 512},
    {"list_cap": 64}, {"force_path": 2, "team": 2}, {"force_path": 2, "team": 7}, {"force_path": 2, "team": 64},
    {"global_dom": 1}, {"global_dom": 1, "list_cap": 64}, {"global_dom": 1, "force_path": 2, "team": 5},
])
def test_random_csp_launch_shapes(ctx, opts):
    """Same answers whatever the launch geometry: tile size, block size, dense-round fallback, team size."""
    V, P, N = 120, 900, 70
    props, lb, ub, sol = random_csp(77, V, P, planted=True, dom=(0, 60))
    L, U = random_nodes(78, lb, ub, N, sol)
    act = random_active(79, N, P, p_off=0.05)
    both(ctx, V, props, L, U, act, f"shapes {opts}", **opts)


@pytest.mark.parametrize("npb", [8, 16, 32])
@pytest.mark.parametrize("seed", range(3))
def test_packed_tiles_binary_csp(ctx, npb, seed):
    """16-bit packed tiles (binary models, every bound within +-16383) against the oracle, and against the 32-bit
    tiles of the same size; odd seeds are unplanted, so nodes fail and the crossing bounds get clamped."""
    V, P, N = 90 + 20 * seed, 700 + 150 * seed, 100 + 7 * seed
    props, lb, ub, sol = random_csp(400 + seed, V, P, planted=seed % 2 == 0, p_tern=0.0, dom=(-40, 60))
    L, U = random_nodes(500 + seed, lb, ub, N, sol if seed % 2 == 0 else None, p_narrow=0.3 if seed % 2 == 0 else 0.05)
    act = random_active(600 + seed, N, P, p_off=0.1)
    both(ctx, V, props, L, U, act, f"packed npb={npb} seed={seed}", nodes_per_block=npb, packed=1)
    both(ctx, V, props, L, U, act, f"unpacked npb={npb} seed={seed}", nodes_per_block=npb, packed=0)


def test_packed_tiles_hand_back_wide_bounds(ctx):
    """A tile with a bound beyond +-16383 cannot use the packed cells: it is re-run with 32-bit cells inside the same
    call, the other tiles of the batch stay packed."""
    V, P, N = 150, 1200, 203
    props, lb, ub, sol = random_csp(910, V, P, planted=True, p_tern=0.0, dom=(0, 80))
    L, U = random_nodes(911, lb, ub, N, sol, p_narrow=0.2)
    rng = np.random.default_rng(912)
    wide = rng.choice(N, size=9, replace=False)
    for r in wide:  # supersets of the model's domains are legal inputs
        v = rng.choice(V, size=3, replace=False)
        L[r, v[0]] = -20000
        U[r, v[1]] = 30000
        L[r, v[2]], U[r, v[2]] = -(1 << 20), (1 << 24)
    act = random_active(913, N, P, p_off=0.05)
    for npb in (8, 16, 32):
        both(ctx, V, props, L, U, act, f"hand-back npb={npb}", nodes_per_block=npb, packed=1)
    # every tile out of range, in place (lb_in == lb_out inside pcp_propagate)
    L[:, 0] = -17000
    both(ctx, V, props, L, U, act, "hand-back all", nodes_per_block=16, packed=1)


def test_packed_tiles_nqueens_frontier(ctx):
    """N-queens-200 breadth-first frontier: packed 32-node tiles, parity with the oracle on every node."""
    n = 200
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1}.items():
        ctx.set_option(k, v)
    from pcp_amd.search import bfs_frontier
    L, U, A, _ = bfs_frontier(ctx, np.ones(n, np.int32), np.full(n, n, np.int32), 300)
    both(ctx, n, props, L, U, A, "nqueens frontier packed", nodes_per_block=32, packed=1)


def test_declared_hull(ctx):
    """pcp_model_set_hull: same answers without the retry launch; a bound outside the hull is a contract violation."""
    import torch
    V, P, N = 150, 1200, 203
    props, lb, ub, sol = random_csp(910, V, P, planted=True, p_tern=0.0, dom=(0, 80))
    L, U = random_nodes(911, lb, ub, N, sol, p_narrow=0.2)
    act = random_active(913, N, P, p_off=0.05)
    om = orc.OracleModel(V, props)
    ref = om.consistency(L, U, act)
    ctx.set_model(V, props)
    ctx.set_hull(int(lb.min()), int(ub.max()))
    for k, v in {"force_path": 0, "nodes_per_block": 16, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1}.items():
        ctx.set_option(k, v)
    got = ctx.propagate(L, U, act)
    assert_parity(ref[:4], got[:4], "declared hull")
    # host-buffer path: validated before the launch
    L2 = L.copy(); L2[5, 7] = int(lb.min()) - 1
    with pytest.raises(E.PcpError) as ei:
        ctx.propagate(L2, U, act)
    assert ei.value.code == -2
    # device path: the offending tile keeps its inputs, its nodes get PCP_STATUS_HULL, stats_read reports the violation
    ctx.set_hull(-20000, 20000)  # a hull too wide for the packed cells: plain 32-bit tiles, no error possible
    got = ctx.propagate(L, U, act)
    assert_parity(ref[:4], got[:4], "wide hull")
    ctx.set_hull(int(lb.min()), int(ub.max()))
    dev = torch.device("cuda:0")
    Ld = L.copy(); Ld[40, 3] = -17000
    t_lb, t_ub = torch.from_numpy(Ld).to(dev), torch.from_numpy(U).to(dev)
    t_act = torch.from_numpy(act.view(np.int64)).to(dev)
    t_st = torch.zeros(N, dtype=torch.uint8, device=dev)
    ctx.stats_reset()
    ctx.propagate_device(N, t_lb, t_ub, t_lb, t_ub, t_act, t_act, t_st)
    st = t_st.cpu().numpy().copy()
    # the violation is STICKY: a clean launch in between must not make the engine forget it, and pcp_branch_device counts
    # the refused nodes in counts[4]
    c_lb = torch.empty((2 * N, V), dtype=torch.int32, device=dev); c_ub = torch.empty_like(c_lb)
    c_act = torch.empty((2 * N, t_act.shape[1]), dtype=torch.int64, device=dev)
    counts = torch.zeros(5, dtype=torch.int32, device=dev)
    ctx.branch_device(N, t_lb, t_ub, t_act, t_st, c_lb, c_ub, c_act, counts)
    torch.cuda.synchronize()
    assert counts.cpu().tolist()[4] == 16
    t2_lb, t2_ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
    t2_st = torch.zeros(N, dtype=torch.uint8, device=dev)
    ctx.propagate_device(N, t2_lb, t2_ub, t2_lb, t2_ub, None, None, t2_st)
    with pytest.raises(E.PcpError) as ei:
        ctx.stats_read()
    assert ei.value.code == -2
    ctx.stats_read()  # reported once, then cleared
    assert (st[32:48] == 0xFE).all() and (st[:32] != 0xFE).all() and (st[48:] != 0xFE).all()
    assert np.array_equal(t_lb.cpu().numpy()[32:48], Ld[32:48])
    ok = np.ones(N, bool); ok[32:48] = False
    assert np.array_equal(st[ok], ref[3][ok])
    ctx.set_model(V, props)  # forgets the hull


def _all_pairs_model(n, kinds, seed, dom=(0, 60)):
    """x_i (kind) x_j + d for all i < j, sorted by i: the words of the table have short slot ranges, so the packed tiles
    sweep it by word groups with the level -1 range test (XLessY words: minimum and maximum tables)."""
    rng = np.random.default_rng(seed)
    sol = rng.integers(dom[0], dom[1] + 1, size=n)
    ii, jj = np.triu_indices(n, 1)
    P = len(ii)
    props = np.zeros(P, dtype=M.PROP_DTYPE)
    props["var"][:] = M.PCP_NOVAR
    props["group"] = np.arange(P)
    kind = rng.choice(kinds, size=P) if len(kinds) > 1 else np.full(P, kinds[0])
    # keep runs of one kind long (whole words): switch kind per block of 64 records
    kind = np.repeat(kind[::64], 64)[:P]
    props["kind"] = kind
    props["var"][:, 0] = ii
    props["var"][:, 1] = jj
    d = np.where(kind == M.LT, sol[ii] - sol[jj] + 1 + rng.integers(0, 8, size=P), rng.integers(-5, 6, size=P))
    clash = (kind == M.NEQ) & (sol[ii] == sol[jj] + d)
    d[clash] += 1
    props["off"][:, 1] = d
    lb = np.full(n, dom[0], np.int32); ub = np.full(n, dom[1], np.int32)
    return props, lb, ub, sol


@pytest.mark.parametrize("kinds", [[M.NEQ], [M.LT], [M.NEQ, M.LT]])
@pytest.mark.parametrize("npb", [8, 16])
def test_word_group_sweep(ctx, kinds, npb):
    """Packed tiles with word descriptors (level -1 on range tables, phase B at the record level) against the oracle and
    against the chunked sweep (word_level = 0); consistent nodes (long cascades), random sub-boxes (failures), ragged batch."""
    n = 90
    props, lb, ub, sol = _all_pairs_model(n, kinds, seed=31 + len(kinds) + npb)
    for planted in (True, False):
        L, U = random_nodes(77 + npb, lb, ub, 75, sol if planted else None, p_narrow=0.25 if planted else 0.08)
        act = random_active(78, 75, len(props), p_off=0.1)
        for wl in (1, 0):
            both(ctx, n, props, L, U, act, f"word sweep kinds={kinds} npb={npb} planted={planted} word_level={wl}",
                 nodes_per_block=npb, packed=1, word_level=wl)
        # implicit nodes with and without the group-level test ahead of the word level
        both(ctx, n, props, L, U, None, f"word sweep kinds={kinds} npb={npb} planted={planted} group_level=0",
             nodes_per_block=npb, packed=1, word_level=1, group_level=0)
    ctx.set_option("word_level", 1)
    ctx.set_option("group_level", 1)


@pytest.mark.parametrize("kinds", [[M.NEQ], [M.LT], [M.NEQ, M.LT]])
def test_group_level_sweep(ctx, kinds):
    """Implicit nodes on a table with several groups of 64 words (all pairs of 200 variables: 5 groups): the group-level range
    test clears whole groups on nodes where few variables are narrowed, and hands the others to the word level."""
    n = 200
    props, lb, ub, sol = _all_pairs_model(n, kinds, seed=77 + len(kinds), dom=(0, 400))
    for p_narrow, planted in ((0.02, True), (0.3, True), (0.03, False)):
        L, U = random_nodes(91, lb, ub, 70, sol if planted else None, p_narrow=p_narrow)
        for gl in (1, 0):
            both(ctx, n, props, L, U, None, f"group level kinds={kinds} p={p_narrow} planted={planted} gl={gl}", force_path=1, nodes_per_block=16, packed=1, word_level=1, group_level=gl)
        pl = ctx.last_plan()
        assert (pl["word_level"], pl["packed"], pl["nodes_per_block"], pl["implicit_active"]) == (2, 1, 16, 1), pl
    ctx.set_option("group_level", 1)


def test_device_entry_refuses_bounds_beyond_bound_max(ctx):
    """pcp_propagate_device checks the bounds it stages: a node with a bound beyond +-PCP_BOUND_MAX is refused (status
    PCP_STATUS_HULL, outputs untouched, sticky flag reported by pcp_stats_read) instead of wrapping i32; its neighbours in the
    tile are propagated as usual.  32-bit tiles, the team geometry and the HBM-resident variant."""
    import torch
    V, P, N = 60, 400, 37
    props, lb, ub, sol = random_csp(77, V, P, planted=True, p_tern=0.2, dom=(-300000, 300000))
    L, U = random_nodes(78, lb, ub, N, sol, p_narrow=0.3)
    om = orc.OracleModel(V, props)
    ref = om.consistency(L, U, None)
    dev = torch.device("cuda:0")
    Lb, Ub = L.copy(), U.copy()
    Lb[5, 3] = -M.PCP_BOUND_MAX - 1
    Ub[20, 11] = M.PCP_BOUND_MAX + 7
    ok = np.ones(N, bool); ok[[5, 20]] = False
    ctx.set_model(V, props)
    for opts in ({"force_path": 1, "nodes_per_block": 8}, {"force_path": 1, "nodes_per_block": 1}, {"force_path": 1, "global_dom": 1}):
        for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1, **opts}.items():
            ctx.set_option(k, v)
        t_lb, t_ub = torch.from_numpy(Lb).to(dev), torch.from_numpy(Ub).to(dev)
        t_st = torch.zeros(N, dtype=torch.uint8, device=dev)
        ctx.stats_reset()
        ctx.propagate_device(N, t_lb, t_ub, t_lb, t_ub, None, None, t_st)
        with pytest.raises(E.PcpError) as ei:
            ctx.stats_read()
        assert ei.value.code == -2, opts
        ctx.stats_read()
        st, gl, gu = t_st.cpu().numpy(), t_lb.cpu().numpy(), t_ub.cpu().numpy()
        assert st[5] == 0xFE and st[20] == 0xFE, (opts, st)
        assert np.array_equal(st[ok], ref[3][ok]), opts
        live = ok & (ref[3] != M.FALSE)
        assert np.array_equal(gl[live], ref[0][live]) and np.array_equal(gu[live], ref[1][live]), opts
        if not opts.get("global_dom"):
            assert np.array_equal(gl[[5, 20]], Lb[[5, 20]]) and np.array_equal(gu[[5, 20]], Ub[[5, 20]]), opts
    for k, v in {"force_path": 0, "nodes_per_block": 0, "global_dom": 0}.items():
        ctx.set_option(k, v)


def test_solo_cascade_forbidden_value_jump(ctx):
    """The tail of a cascade (one changed variable per round) is re-run in place, and a bound that walks through values
    forbidden by assigned neighbours is moved in one jump (pcp_kernels.hip, rounds c0) — same fixpoint as the reference's
    one-value-per-wake-up chain, with the shortcut on and off, packed and 32-bit cells, explicit and implicit nodes."""
    # (1) partial N-queens assignments: every unassigned queen's bounds climb through the columns / diagonals taken
    n = 72
    props = M.nqueens_props(n)
    rng = np.random.default_rng(11)
    N = 48
    L, U = np.ones((N, n), np.int32), np.full((N, n), n, np.int32)
    for i in range(N):
        k = int(rng.integers(n // 4, n - 1))
        placed = []
        for r_ in rng.permutation(n)[:k]:  # k queens placed without a clash; propagation then fails about a third of the nodes
            free = [c for c in range(1, n + 1) if all(c != pc and abs(c - pc) != abs(int(r_) - pr) for pr, pc in placed)]
            if not free:
                break
            c = int(rng.choice(free))
            placed.append((int(r_), c))
            L[i, r_] = c; U[i, r_] = c
    try:
        for solo in (1, 0):
            for opts in ({"nodes_per_block": 16}, {"nodes_per_block": 16, "packed": 0}, {"nodes_per_block": 1}, {"force_path": 2}):
                ref, got = both(ctx, n, props, L, U, None, f"solo cascade nqueens solo={solo} {opts}", solo_cascade=solo, **({"force_path": 1} | opts))
        assert (ref[3] == M.FALSE).any() and (ref[3] != M.FALSE).any()
        act = random_active(5, N, orc.OracleModel(n, props).n_units, p_off=0.1)
        for solo in (1, 0):
            both(ctx, n, props, L, U, act, f"solo cascade nqueens explicit rows solo={solo}", solo_cascade=solo, force_path=1, nodes_per_block=16)
        # (2) one variable against 3900 constants: x != c for c in [0, 2000) and (2099, 4000] leaves [2000, 2099] — two jumps
        units = [M.XNeqY(M.Identity(0), M.Constant(c)) for c in list(range(0, 2000)) + list(range(2100, 4001))]
        units += [M.XLessY(M.Identity(1), M.Identity(0))]  # x1 < x0: follows the jump in the next round
        props2 = M.lower_units(units, 2)
        L2 = np.array([[0, 0], [5, 0], [0, 0], [2000, 1990], [1000, 0]], np.int32)
        U2 = np.array([[4000, 4000], [3990, 4000], [1999, 10], [2099, 2050], [3000, 999]], np.int32)
        for solo in (1, 0):
            ref, got = both(ctx, 2, props2, L2, U2, None, f"solo cascade constants solo={solo}", solo_cascade=solo, force_path=1, nodes_per_block=1)
        assert ref[3][2] == M.FALSE and (ref[0][0] == [2000, 0]).all() and (ref[1][0] == [2099, 2098]).all()
    finally:
        ctx.set_option("solo_cascade", 1)


@pytest.mark.parametrize("n,dive", [(60, 25), (90, 60)])
def test_packed_tiles_deep_search_nodes(ctx, n, dive):
    """Open nodes taken from deep in a device-resident search (many assigned queens: the range tests clear little, the
    tile switches to the chunked record-level sweep after phase A) — every variant of the engine against the oracle."""
    from pcp_amd.search_device import DeviceSearch
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1}.items():
        ctx.set_option(k, v)
    ds = DeviceSearch(ctx, batch=64, capacity=4096)
    ds.reset(np.ones(n, np.int32), np.full(n, n, np.int32))
    ds.advance(max_rounds=dive, batch=1)
    ds.advance(max_rounds=5, batch=64)
    lb, ub, act = (t.cpu().numpy() for t in ds.top(200))
    assert lb.shape[0] >= 64
    A = act.view(np.uint64)
    om = orc.OracleModel(n, props)
    ref = om.consistency(lb, ub, A)
    for opts in ({"nodes_per_block": 16}, {"nodes_per_block": 8}, {"nodes_per_block": 16, "word_level": 0}, {"nodes_per_block": 16, "packed": 0},
                 {"nodes_per_block": 32}):
        for k, v in {"packed": 1, "word_level": 1, **opts}.items():
            ctx.set_option(k, v)
        got = ctx.propagate(lb, ub, A)
        assert_parity(ref[:4], got[:4], f"deep nodes n={n} {opts}")
    ctx.set_option("nodes_per_block", 0)


@pytest.mark.parametrize("n_nodes", [1, 17, 255, 2051, 9000])
def test_batch_sizes_auto_tiling(ctx, n_nodes):
    """Default launch policy over batch sizes from one node to many tiles per CU (team path, small tiles, packed 16- and
    32-node tiles with ragged last tile): all against the oracle."""
    n = 24
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    L, U = random_nodes(100 + n_nodes, lb0, ub0, n_nodes, None, p_narrow=0.12)
    act = random_active(200 + n_nodes, n_nodes, len(props), p_off=0.1)
    ref, got = both(ctx, n, props, L, U, act, f"auto tiling n_nodes={n_nodes}")
    assert got[4]["nodes"] == n_nodes


def test_soak_short():
    """Eight seconds of tools/soak.py: random all-pairs models (kinds, ragged x-blocks, hulls, batch and tile sizes) on the
    packed / word-group paths against the oracle; the long version checked 136 000 launches."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "8", "20260928"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_long_cascade(ctx):
    """x0 < x1 < ... < x299 on [0,299]: a 300-wave cascade ending in a full assignment (status True)."""
    n = 300
    vs, cs = M.VStore(), M.CStore()
    xs = [vs.alloc((0, n - 1)) for _ in range(n)]
    for i in range(n - 1):
        cs.alloc(M.XLessY(xs[i], xs[i + 1]))
    lb, ub = vs.bounds()
    for opts in ({}, {"force_path": 2, "team": 4}):
        ref, got = both(ctx, n, cs.lower(n), lb[None], ub[None], None, "cascade", **opts)
        assert got[3][0] == M.TRUE and np.array_equal(got[0][0], np.arange(n))


@pytest.mark.parametrize("n", [8, 20, 50])
def test_nqueens_dfs_nodes(ctx, n):
    """The first K nodes of the reference's default DFS (FirstSmallestVar/MiddleVal/BinarySplit), each node's
    folded input domains + active mask -> fixpoint, one launch for all of them, batch and team paths."""
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    K = 200
    _, _, rec, _ = om.search(np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True, node_limit=K, max_records=K)
    keep = (rec["lb_in"] <= rec["ub_in"]).all(axis=1)
    ref = (rec["lb_out"][keep], rec["ub_out"][keep], rec["active_out"][keep], rec["status"][keep])
    ctx.set_model(n, props)
    for opts in ({"force_path": 1}, {"force_path": 1, "nodes_per_block": 5}, {"force_path": 2, "team": 3}):
        for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1, **opts}.items():
            ctx.set_option(k, v)
        got = ctx.propagate(rec["lb_in"][keep], rec["ub_in"][keep], rec["active_in"][keep])
        assert_parity(ref, got[:4], f"nqueens({n}) {opts}")
    assert (ref[3] == 0).any() or n == 50


def test_nqueens_1000_root_and_dive(ctx):
    """BASELINE config 2 shape: V=1000, P=1 498 500.  Root node plus nodes with q0..q_{k-1} assigned along a
    diagonal-free prefix; oracle run without the duplicate-subscription assert to stay within seconds."""
    n = 1000
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    N = 6
    L = np.ones((N, n), np.int32)
    U = np.full((N, n), n, np.int32)
    prefix = [1, 3, 5, 7, 9, 11, 13, 15]  # q_i = 2i+1: mutually non-attacking
    for k in range(1, N):
        for i in range(k + 1):
            L[k, i] = U[k, i] = prefix[i]
    ref = om.consistency(L, U, None, check_dup=False)
    ctx.set_model(n, props)
    for opts in ({"force_path": 1}, {"force_path": 2}):
        for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1, **opts}.items():
            ctx.set_option(k, v)
        got = ctx.propagate(L, U, E.full_active(N, om.n_units))
        assert_parity(ref[:4], got[:4], f"nqueens1000 {opts}")
        assert got[4]["steps"] >= N * om.n_units


def test_all_different_units_on_the_small_kernel(ctx):
    """Distinct::new units (distinct.rs:63-83) filtered as a group by pcp_small.hip — value mask, duplicate detection, entailment — against
    the oracle's pairwise XNeqY: random boxes with many assigned variables (chains of forced values, duplicates), domains wider than the
    128-value mask (that round falls back to the pairs), two units sharing variables, a unit that ends up entailed."""
    rng = np.random.default_rng(77)
    for case in range(6):
        V = [9, 14, 20, 33, 12, 40][case]
        dom = [(0, 12), (-5, 20), (0, 40), (1, 60), (0, 300), (0, 70)][case]
        vs, cs = M.VStore(), M.CStore()
        xs = [vs.alloc(dom) for _ in range(V)]
        cs.alloc(M.Distinct(xs[: V - 2]))
        if case in (1, 3):
            cs.alloc(M.Distinct(xs[2:]))          # a second unit overlapping the first
        for i in range(V - 1):
            if case != 5 and rng.random() < 0.4:
                cs.alloc(M.XLessY(xs[i], M.Addition(xs[i + 1], int(rng.integers(0, 4)))))
        props = cs.lower(V)
        lb0, ub0 = vs.bounds()
        N = 300
        L = np.tile(lb0, (N, 1)); U = np.tile(ub0, (N, 1))
        for k in range(N):
            p_assign = rng.uniform(0.1, 0.95)
            for v in range(V):
                r = rng.random()
                if r < p_assign:
                    L[k, v] = U[k, v] = int(rng.integers(dom[0], min(dom[1], dom[0] + V + 3) + 1))   # values collide often
                elif r < p_assign + 0.2:
                    a_ = int(rng.integers(dom[0], dom[1] + 1)); b_ = int(rng.integers(a_, dom[1] + 1))
                    L[k, v], U[k, v] = a_, b_
        if case == 5:  # pairwise disjoint singletons: the unit is entailed
            for k in range(0, N, 3):
                perm = rng.permutation(dom[1] - dom[0] + 1)[:V] + dom[0]
                L[k] = U[k] = perm
        L, U = L.astype(np.int32), U.astype(np.int32)
        act = random_active(500 + case, N, len(cs), p_off=0.1)
        ref, got = both(ctx, V, props, L, U, act, f"all-different case {case}")
        assert ctx.n_units == len(cs)
        assert (ref[3] == 0).any()
        if case == 5:
            assert (ref[3] == 1).any()  # (every third node is a permutation: the unit, the only one, is entailed — True)


def test_all_different_group_counts_the_units_pairs(ctx):
    """a Distinct unit filtered as a group counts its cnt (cnt - 1) / 2 pair filters per round — what the pairwise path
    (small_alldiff = 0) and the reference run — not twice that.  Nodes on which nothing narrows take one round either way: equal steps."""
    V = 12
    vs, cs = M.VStore(), M.CStore()
    xs = [vs.alloc((0, 40)) for _ in range(V)]
    cs.alloc(M.Distinct(xs))
    props = cs.lower(V)
    lb0, ub0 = vs.bounds()
    N = 64
    L = np.tile(lb0, (N, 1)).astype(np.int32); U = np.tile(ub0, (N, 1)).astype(np.int32)
    L[1::2, 0] = U[1::2, 0] = 20  # an interior value: still nothing to narrow (a value is removed only at a bound, x_neq_y.rs:82-93)
    ctx.set_model(V, props)
    ctx.set_hull(0, 40)
    ctx.set_option("neq_path", 0)  # (a store of XNeqY alone would take the assignment-driven kernel: this test is about the small kernel's counters)
    steps = {}
    for ad in (1, 0):
        ctx.set_option("small_alldiff", ad)
        lb, ub, act, st, stats = ctx.propagate_implicit(L, U)
        assert ctx.last_plan()["path"] == 4 and np.array_equal(lb, L) and np.array_equal(ub, U) and (st == 2).all()
        steps[ad] = (stats["steps"], stats["evaluated"])
    ctx.set_option("small_alldiff", 1)
    ctx.set_option("neq_path", 1)
    assert steps[1] == steps[0] == (N * V * (V - 1) // 2, N * V * (V - 1) // 2), steps


def test_golomb_distinct_sum_network(ctx):
    """BASELINE config 4: EQ3 sum network + ONE Distinct unit (a Conjunction group of 990 XNeqY) + LT chain, V=55;
    the search frontier after 12 BinarySplit levels, propagated in one launch (batch and team paths)."""
    vs, cs = M.golomb(10, 80)
    V = len(vs)
    props = cs.lower(V)
    om = orc.OracleModel(V, props)
    assert om.n_units == len(cs) and len(props) == 1 + 9 + 45 + 990 + 1
    lb0, ub0 = vs.bounds()
    from oracle_ctx import OracleCtx
    from pcp_amd import search as S
    L, U, A, _ = S.bfs_frontier(OracleCtx(V, props), lb0, ub0, 512, max_rounds=14)
    assert L.shape[0] >= 256
    ref = om.consistency(L, U, A)
    assert (ref[3] == 0).any() and (ref[3] == 2).any()
    ctx.set_model(V, props)
    for opts in ({"force_path": 1}, {"force_path": 1, "nodes_per_block": 2}, {"force_path": 2, "team": 3}):
        for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1, **opts}.items():
            ctx.set_option(k, v)
        got = ctx.propagate(L, U, A)
        assert_parity(ref[:4], got[:4], f"golomb {opts}")
        assert got[4]["steps3"] > 0
    # the bench leg's own size: a 4096-node frontier (grown with the engine), every node against the oracle
    for k, v in {"force_path": 0, "nodes_per_block": 0, "team": 0}.items():
        ctx.set_option(k, v)
    L4, U4, A4, _ = S.bfs_frontier(ctx, lb0, ub0, 4096, max_rounds=24)
    assert L4.shape[0] >= 2048
    ref4 = om.consistency(L4, U4, A4)
    got4 = ctx.propagate(L4, U4, A4)
    assert ctx.last_plan()["path"] == 4  # the bench leg's kernel: one wavefront per node, the Distinct of 990 pairs as ONE all-different step
    assert_parity(ref4[:4], got4[:4], f"golomb frontier of {L4.shape[0]} nodes")
    ctx.set_option("small_alldiff", 0)   # the same kernel with the Distinct pair by pair
    got4p = ctx.propagate(L4, U4, A4)
    ctx.set_option("small_alldiff", 1)
    assert ctx.last_plan()["path"] == 4
    assert_parity(ref4[:4], got4p[:4], f"golomb frontier of {L4.shape[0]} nodes, pairwise Distinct")
    got4i = ctx.propagate_implicit(L4, U4)
    assert_parity(om.consistency(L4, U4, None)[:4], got4i[:4], f"golomb frontier of {L4.shape[0]} nodes [implicit]")


@pytest.mark.parametrize("n", [6, 9])
def test_nqueens_global_distinct_search(ctx, n):
    """The doc-comment N-queens (lib.rs:56) with ONE Distinct unit: whole search on the GPU engine through the host
    driver == the oracle's search (node, failure and solution counts; the tree is schedule-independent)."""
    from pcp_amd import search as S
    vs, cs = M.nqueens(n, "global")
    props = cs.lower(n)
    lb0, ub0 = vs.bounds()
    ss, _, _, _ = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=True)
    ctx.set_model(n, props)
    for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1}.items():
        ctx.set_option(k, v)
    st = S.dfs(ctx, lb0, ub0, all_solutions=True, batch=32)
    assert (st.num_solution, st.num_nodes, st.num_failed_node) == (ss["num_solution"], ss["num_nodes"], ss["num_failed_node"])


@pytest.mark.parametrize("style", ["global", "join"])
def test_grouped_units_on_packed_tiles(ctx, style):
    """Distinct / join_distinct units (several records per `active` bit) on packed tiles with the word-group sweep: the
    unit-level rows are expanded to record-level live rows, swept in place and contracted back."""
    n = 40
    vs, cs = M.nqueens(n, style if style == "global" else "join")
    props = cs.lower(n)
    lb0, ub0 = vs.bounds()
    om = orc.OracleModel(n, props)
    L, U = random_nodes(4242, lb0, ub0, 90, None, p_narrow=0.06)
    act = random_active(4243, 90, om.n_units, p_off=0.1)
    for opts in ({"nodes_per_block": 16}, {"nodes_per_block": 8, "word_level": 0}, {"nodes_per_block": 16, "packed": 0}):
        both(ctx, n, props, L, U, act, f"grouped {style} {opts}", **opts)


def test_config3_random_binary_csp_full_size(ctx):
    """BASELINE config 3 at full size: 50 000 Interval<i32> variables (400 KB of bounds per node: more than LDS, so
    the HBM-resident-domain variant runs), 500 000 `x ◇ y + c` constraints, planted solution, long cascades."""
    V, P = 50_000, 500_000
    props, lb, ub, sol = planted_binary_csp(0xC3, V, P)
    L, U = unit_narrowing_prefix(0xC3 + 1, lb, ub, sol, 32)
    om = orc.OracleModel(V, props)
    ref = om.consistency(L, U, None)
    assert (ref[3] == 2).all() and ((ref[0] != L) | (ref[1] != U)).sum() > V
    ctx.set_model(V, props)
    for opts in ({}, {"force_path": 2, "team": 16}, {"hull": 1, "force_path": 1}):
        if opts.pop("hull", 0):
            ctx.set_hull(0, 999)  # the declared hull lets the 50 000-variable store sit in LDS as 10-bit cells
        for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1, **opts}.items():
            ctx.set_option(k, v)
        got = ctx.propagate(L, U, E.full_active(L.shape[0], P))
        assert_parity(ref[:4], got[:4], f"config3 {opts}")
    assert ctx.last_plan()["global_dom"] == 2


@pytest.mark.parametrize("seed", range(3))
def test_ten_bit_lds_cells(ctx, seed):
    """The HBM-resident variant with a declared hull of at most 1024 values keeps the domains in LDS as 10-bit cells (three per
    u64, CAS narrowing): random binary/ternary CSPs with offsets and constants, planted and failing, against the oracle."""
    V, P, N = 200 + 50 * seed, 1500 + 400 * seed, 60
    props, lb, ub, sol = random_csp(880 + seed, V, P, planted=seed != 1, dom=(-100, 900))
    L, U = random_nodes(890 + seed, lb, ub, N, sol if seed != 1 else None, p_narrow=0.2 if seed != 1 else 0.04)
    act = random_active(895 + seed, N, P, p_off=0.1)
    om = orc.OracleModel(V, props)
    ref = om.consistency(L, U, act)
    ctx.set_model(V, props)
    ctx.set_hull(-100, 900)
    for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 2, "packed": 1, "word_level": 1}.items():
        ctx.set_option(k, v)
    got = ctx.propagate(L, U, act)
    assert ctx.last_plan()["global_dom"] == 2
    assert_parity(ref[:4], got[:4], f"10-bit cells seed={seed}")
    gi = ctx.propagate_implicit(L, U)
    assert_parity(om.consistency(L, U, None)[:4], gi[:4], f"10-bit cells implicit seed={seed}")
    ctx.set_option("global_dom", 0)
    ctx.set_model(V, props)


def test_mul3_with_addition_views(ctx):
    """XEqYMulZ takes arbitrary views (x_eq_y_mul_z.rs:99-105): Addition offsets on all three operands, constants, negative
    factors staying out of the unpinned corner (non-negative operand ranges)."""
    rng = np.random.default_rng(2026)
    units, V = [], 12
    val = np.full(V, -1, np.int64)  # a planted assignment: variables 0..5 are factors, 6..11 products
    val[:6] = rng.integers(0, 5, size=6)
    while len(units) < 40:
        x = int(rng.integers(6, 12))
        y, z = (int(t) for t in rng.choice(6, size=2, replace=False))
        b, c = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        const = rng.random() < 0.2
        zc = int(rng.integers(0, 5)) if const else int(val[z]) + c
        prod = (int(val[y]) + b) * zc
        if val[x] < 0:
            a = int(rng.integers(-4, 5))
            if not 0 <= prod - a <= 30:
                continue
            val[x] = prod - a
        else:
            a = prod - int(val[x])
            if not -4 <= a <= 4:
                continue
        units.append(M.XEqYMulZ(M.Addition(M.Identity(x), a), M.Addition(M.Identity(y), b), M.Constant(zc) if const else M.Addition(M.Identity(z), c)))
    for i in range(V - 1):
        units.append(M.XLessY(M.Identity(i), M.Addition(M.Identity(i + 1), 31)))
    props = M.lower_units(units, V)
    lb0, ub0 = np.zeros(V, np.int32), np.full(V, 30, np.int32)
    L1, U1 = random_nodes(4711, lb0, ub0, 40, val, p_narrow=0.3)    # boxes around the planted assignment
    L2, U2 = random_nodes(4713, lb0, ub0, 40, None, p_narrow=0.15)  # arbitrary boxes (mostly inconsistent)
    L, U = np.concatenate([L1, L2]), np.concatenate([U1, U2])
    act = random_active(4712, 80, len(units), p_off=0.1)
    ref, got = both(ctx, V, props, L, U, act, "mul3 with views")
    assert got[4]["steps3"] > 0 and (ref[3] == 0).any() and (ref[3] != 0).any()


def test_device_branching_matches_host_branching(ctx):
    """pcp_branch_device == the host driver's (numpy) FirstSmallestVar / MiddleVal / BinarySplit on a propagated batch,
    including the reference's selector tables (first_smallest_var.rs:63-72, binary_split.rs:108-133)."""
    import torch
    from pcp_amd import search as S
    n = 12
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    _, _, rec, _ = om.search(np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True, node_limit=300, max_records=300)
    lb, ub, act, status = rec["lb_out"], rec["ub_out"], rec["active_out"], rec["status"]
    unk = status == 2
    hl, hu, ha = S.branch(lb[unk], ub[unk], act[unk])
    ctx.set_model(n, props)
    dev = torch.device("cuda", 0)
    N, V, W = lb.shape[0], n, act.shape[1]
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
    d_lb, d_ub, d_act, d_st = t(lb, np.int32), t(ub, np.int32), t(act, np.int64), t(status, np.uint8)
    c_lb = torch.empty((2 * N, V), dtype=torch.int32, device=dev)
    c_ub = torch.empty_like(c_lb)
    c_act = torch.empty((2 * N, W), dtype=torch.int64, device=dev)
    counts = torch.zeros(5, dtype=torch.int32, device=dev)
    ctx.branch_device(N, d_lb, d_ub, d_act, d_st, c_lb, c_ub, c_act, counts)
    torch.cuda.synchronize()
    nc, nt, nf, nu, n_other = counts.cpu().tolist()
    assert n_other == 0
    assert (nc, nt, nf, nu) == (2 * int(unk.sum()), int((status == 1).sum()), int((status == 0).sum()), int(unk.sum()))
    assert np.array_equal(c_lb[:nc].cpu().numpy(), hl) and np.array_equal(c_ub[:nc].cpu().numpy(), hu)
    assert np.array_equal(c_act[:nc].cpu().numpy().view(np.uint64), ha)
    g = json.load(open(os.path.join(GOLDEN, "engine_kats.json")))["search"]
    root = g["binary_split"]["root"]
    for c in g["binary_split"]["cases"]:  # make var c the smallest non-assigned one
        doms = [[5, 5]] * 3
        doms[c["var"]] = root[c["var"]]
        ctx.set_model(3, M.lower_units([], 3))
        dl = torch.tensor([[d[0] for d in doms]], dtype=torch.int32, device=dev)
        du = torch.tensor([[d[1] for d in doms]], dtype=torch.int32, device=dev)
        ds = torch.tensor([2], dtype=torch.uint8, device=dev)
        ol, ou = torch.empty((2, 3), dtype=torch.int32, device=dev), torch.empty((2, 3), dtype=torch.int32, device=dev)
        ctx.branch_device(1, dl, du, None, ds, ol, ou, None, counts)
        torch.cuda.synchronize()
        got = [[int(ol[0, c["var"]]), int(ou[0, c["var"]])], [int(ol[1, c["var"]]), int(ou[1, c["var"]])]]
        assert got == c["children"]


@pytest.mark.parametrize("n,batch", [(6, 1), (8, 1), (8, 64), (9, 256)])
def test_device_resident_search(ctx, n, batch):
    """Whole search with stack, propagation and branching on the GPU: the reference's tree exactly
    (solutions / nodes / failures of the oracle's DFS; all_solution.rs:70 for the counts)."""
    from pcp_amd.search_device import DeviceSearch
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1}.items():
        ctx.set_option(k, v)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    ss, _, _, sol = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=True)
    st = DeviceSearch(ctx, batch=batch, capacity=4096).run(lb0, ub0, all_solutions=True, keep_solutions=400)
    assert (st.num_solution, st.num_nodes, st.num_failed_node) == (ss["num_solution"], ss["num_nodes"], ss["num_failed_node"])
    assert len({tuple(s) for s in st.solutions}) == ss["num_solution"]
    if batch == 1:  # exact reference order: the first solution found is the reference's
        one = DeviceSearch(ctx, batch=1, capacity=4096).run(lb0, ub0, all_solutions=False, keep_solutions=1)
        ss1, _, _, sol1 = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=False)
        assert one.num_nodes == ss1["num_nodes"] and np.array_equal(one.solutions[0], sol1)


@pytest.mark.parametrize("n", [1, 4, 6, 8, 10])
def test_device_side_dfs(ctx, n):
    """pcp_dfs_device: the reference's one-node-per-step search with no host in the loop == the oracle's DFS, node for node:
    first solution (one_solution.rs:121-128 statuses), all solutions (all_solution.rs:70), StopNode (stop_node.rs:83-104)."""
    props = M.nqueens_props(n) if n > 1 else M.lower_units([], 1)
    ctx.set_model(n, props)
    for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1}.items():
        ctx.set_option(k, v)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    om = orc.OracleModel(n, props)
    ss1, _, _, sol1 = om.search(lb0, ub0, all_solutions=False)
    one = ctx.dfs_device(lb0, ub0, 100000, capacity=256, stop_on_solution=True)
    assert (one["nodes"], one["solutions"], one["failed"], one["error"]) == (ss1["num_nodes"], ss1["num_solution"], ss1["num_failed_node"], 0)
    if ss1["num_solution"]:
        assert np.array_equal(one["first_solution"], sol1)
    ssa, _, _, _ = om.search(lb0, ub0, all_solutions=True)
    al = ctx.dfs_device(lb0, ub0, 100000, capacity=256, stop_on_solution=False, chunk=97)
    assert (al["nodes"], al["solutions"], al["failed"], al["open"]) == (ssa["num_nodes"], ssa["num_solution"], ssa["num_failed_node"], 0)
    if n == 6:
        lim = ctx.dfs_device(lb0, ub0, 100000, capacity=256, stop_on_solution=False, node_limit=10)
        assert lim["nodes"] == 10 and lim["stopped"]
    assert ctx.propagate(lb0[None], ub0[None])[3][0] in (0, 1, 2)  # the context still serves ordinary calls


@pytest.mark.parametrize("opts", [{}, {"neq_dfs": 0}, {"neq_dfs": 0, "small_path": 0}])
def test_node_limit_lands_on_leaves_and_inner_nodes(ctx, opts):
    """StopNode under Monitor (stop_node.rs:57-62, 90-97): the node that reaches the limit is a node and nothing else.  n-queens 6, EVERY limit
    up to the size of the tree (the limit falls on failures, solutions and inner nodes), for the in-kernel search loop of pcp_neq.hip, the
    stepping kernel of the generic path, the batched device search and the set forest: all four agree with the oracle."""
    from pcp_amd.search_device import DeviceSearch
    n = 6
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    for k, v in {"force_path": 0, "nodes_per_block": 0, "block_threads": 1024, "team": 0, "list_cap": 2048, "global_dom": 0, "packed": 1, "word_level": 1,
                 "neq_dfs": 1, "small_path": 1, **opts}.items():
        ctx.set_option(k, v)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    om = orc.OracleModel(n, props)
    total = om.search(lb0, ub0, all_solutions=True)[0]["num_nodes"]
    try:
        for limit in range(1, total + 1):
            ss = om.search(lb0, ub0, all_solutions=True, node_limit=limit)[0]
            want = (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
            r = ctx.dfs_device(lb0, ub0, 100000, capacity=64, stop_on_solution=False, node_limit=limit, chunk=7)
            assert (r["nodes"], r["solutions"], r["failed"]) == want and r["error"] == 0, (limit, r, want)
            if not opts and limit % 5 == 0:
                st = DeviceSearch(ctx, batch=1, capacity=256).run(lb0, ub0, all_solutions=True, node_limit=limit)
                assert (st.num_nodes, st.num_solution, st.num_failed_node) == want, limit
    finally:
        ctx.set_option("neq_dfs", 1); ctx.set_option("small_path", 1)


def test_parallel_search_device_single_rank(ctx):
    """parallel_search_device with world_size 1 (the driver's GPU box has one GPU): the exchange steps run (all_gather,
    all_reduce) and the totals are the reference's tree."""
    import torch
    import torch.distributed as dist
    from pcp_amd import distributed as D
    from pcp_amd.search_device import DeviceSearch
    n = 8
    props = M.nqueens_props(n)
    ctx.set_model(n, props)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ds = DeviceSearch(ctx, batch=64, capacity=4096)
        nodes, sols, fails, steps, moved = D.parallel_search_device(ds, np.ones(n, np.int32), np.full(n, n, np.int32), dist)
    finally:
        dist.destroy_process_group()
    assert (nodes, sols, fails, moved) == (779, 92, 298, 0) and steps > 0


def test_cpp_host_mirror_nqueens():
    """The C++ host side (pcp_amd/host/pcp_host.hpp) running the reference's n-queens example code
    (example/src/nqueens.rs:28-74) with every node's fixpoint on the GPU: same first solution, same node and failure
    counts as the oracle's DFS, same all-solution counts as the reference's test table (all_solution.rs:70)."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    exe = os.path.join(g.ROOT, "pcp_amd", "host", "examples", "nqueens")
    for n in (1, 2, 3, 4, 6, 8, 10):
        props = M.nqueens_props(n) if n > 1 else M.lower_units([], 1)
        lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
        out = json.loads(subprocess.run([exe, str(n)], check=True, capture_output=True, text=True).stdout)
        ss, _, _, sol = orc.OracleModel(n, props).search(lb0, ub0, all_solutions=False)
        assert out["nodes"] == ss["num_nodes"] and out["failed"] == ss["num_failed_node"] and out["solutions"] == ss["num_solution"], (n, out, ss)
        assert out["status"] == ("Satisfiable" if ss["num_solution"] else "Unsatisfiable")
        if ss["num_solution"]:
            assert out["first"] == [int(v) for v in sol]
    counts = [1, 0, 0, 2, 10, 4, 40, 92]
    for n in (4, 6, 8):
        out = json.loads(subprocess.run([exe, str(n), "all"], check=True, capture_output=True, text=True).stdout)
        ss, _, _, _ = orc.OracleModel(n, M.nqueens_props(n)).search(np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True)
        assert out["solutions"] == counts[n - 1] and out["nodes"] == ss["num_nodes"] and out["failed"] == ss["num_failed_node"]
    out = json.loads(subprocess.run([exe, "6", "all", "10"], check=True, capture_output=True, text=True).stdout)
    assert out["nodes"] == 10 and out["status"] == "EndOfSearch"  # search/stop_node.rs:82-104
    # label / alloc(x < y) / consistency / restore / alloc(x > y) / consistency must propagate x > y, not the stale x < y
    out = json.loads(subprocess.run([exe, "restore-test"], check=True, capture_output=True, text=True).stdout)
    assert out == {"first": [[0, 8], [1, 9]], "second": [[1, 9], [0, 8]]}


def test_cpp_host_resident_store():
    """ResidentGpuCStore (pcp_amd/host/pcp_host_resident.hpp, the compiled twin of integration/pcp-gpu-cstore/src/resident.rs): the node's rows
    stay in HBM between consistency() calls (pcp_propagate_device), only the changed index range goes up.  Node for node the same search as the
    host-buffer store and the oracle — and far fewer bytes over PCIe than whole nodes."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    exe = os.path.join(g.ROOT, "pcp_amd", "host", "examples", "nqueens")
    for n, args in [(8, []), (10, []), (8, ["all"]), (6, ["all", "10"]), (1, []), (3, [])]:
        plain = json.loads(subprocess.run([exe, str(n), *args], check=True, capture_output=True, text=True).stdout)
        res = json.loads(subprocess.run([exe, "resident", str(n), *args], check=True, capture_output=True, text=True).stdout)
        for k in ("status", "solutions", "nodes", "failed", "first"):
            assert plain[k] == res[k], (n, args, k, plain, res)
        assert res["pcie_whole_node"] > 0 and plain["pcie_whole_node"] == 0
        if res["nodes"] > 4:
            assert res["pcie_in"] + res["pcie_out"] < res["pcie_whole_node"]
            assert res["pcie_in"] * 4 < res["pcie_whole_node"]  # the way in is a small part of what whole nodes would have cost
    n = 8
    ss, _, _, _ = orc.OracleModel(n, M.nqueens_props(n)).search(np.ones(n, np.int32), np.full(n, n, np.int32), all_solutions=True)
    res = json.loads(subprocess.run([exe, "resident", str(n), "all"], check=True, capture_output=True, text=True).stdout)
    assert (res["nodes"], res["solutions"], res["failed"]) == (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])


def test_contract_errors(ctx):
    ctx.set_model(2, M.lower_units([M.XLessY(M.Identity(0), M.Identity(1))], 2))
    with pytest.raises(E.PcpError) as e:
        ctx.propagate(np.array([[3, 0]], np.int32), np.array([[2, 5]], np.int32))  # empty initial domain
    assert e.value.code == -2
    bad = M.lower_units([M.XLessY(M.Identity(0), M.Identity(1))], 2)
    bad["var"][0][1] = 9
    with pytest.raises(E.PcpError) as e:
        ctx.set_model(2, bad)
    assert e.value.code == -2
    same = M.lower_units([M.XLessY(M.Identity(0), M.Identity(1))], 2)
    same["var"][0][1] = 0
    with pytest.raises(E.PcpError) as e:
        ctx.set_model(2, same)
    assert e.value.code == -2


def test_units_never_join_across_pushes_and_active_tail_bits(ctx):
    """two Conjunction units pushed by two calls with the same `group` value stay two units; `active` rows with
    bits at or above n_units are rejected."""
    V = 4
    u1 = M.Conjunction((M.XLessY(M.Identity(0), M.Identity(1)), M.XLessY(M.Identity(1), M.Identity(2))))
    u2 = M.Conjunction((M.XNeqY(M.Identity(2), M.Identity(3)), M.XLessY(M.Identity(0), M.Identity(3))))
    p1, p2 = M.lower_units([u1], V), M.lower_units([u2], V)  # both carry group 0, group_kind 1
    ctx.set_model(V, p1)
    ctx.push_props(p2)
    assert ctx.n_units == 2
    both = np.concatenate([p1, M.lower_units([u2], V, gid_base=1)])
    om = orc.OracleModel(V, both)
    assert om.n_units == 2
    lb, ub = np.zeros((1, V), np.int32), np.full((1, V), 3, np.int32)
    ref = om.consistency(lb, ub, None)
    got = ctx.propagate(lb, ub, E.full_active(1, 2))
    assert_parity(ref[:4], got[:4], "two pushes, same gid")
    with pytest.raises(E.PcpError) as e:
        ctx.propagate(lb, ub, np.array([[0b111]], np.uint64))
    assert e.value.code == -1


def test_truncate_mirrors_restore(ctx):
    """pcp_model_truncate ≡ FrozenStore::restore's truncate (propagation/store.rs:319-323)."""
    vs, cs = M.chained_lt(10)
    props = cs.lower(10)
    lb, ub = vs.bounds()
    ctx.set_model(10, props)
    full = ctx.propagate(lb[None], ub[None])
    assert full[3][0] == M.TRUE
    ctx.truncate(5)
    om = orc.OracleModel(10, props[:5])
    ref = om.consistency(lb[None], ub[None])
    got = ctx.propagate(lb[None], ub[None], E.full_active(1, 5))
    assert_parity(ref[:4], got[:4], "truncate")
    ctx.push_props(props[5:])
    again = ctx.propagate(lb[None], ub[None])
    assert np.array_equal(again[0], full[0]) and again[3][0] == M.TRUE
