"""Tile tickets of the assignment-driven kernel (pcp_neq.hip, option neq_dynamic): a persistent workgroup takes its first `neq_dynamic` tiles by the
fixed stride and draws the later ones from one of eight device-side tickets; the last workgroup to finish zeroes them for the next launch.  The
schedule must not show in the result: every node's status, domains and counters equal the oracle's (Store::consistency, propagation/store.rs:125-164),
whatever the number of static tiles, launch after launch on the same context (the tickets must come back to zero), with ragged last tiles and with a
launch too small to draw at all in between.  Bit-exact (integer work)."""
import numpy as np
import pytest

from oracle import oracle as orc
import pcp_amd.engine as E

from test_neq_path import neq_model, nodes_with_assignments
from util import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = E.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def batch():
    V, P, dom = 40, 300, (0, 12)
    props = neq_model(4242, V, P, dom)
    L0, U0 = nodes_with_assignments(4343, V, 1500, dom)
    rng = np.random.default_rng(7)
    pick = rng.integers(0, L0.shape[0], size=13001)  # (13001: the last tile is ragged for every tile size; more than 3 x 1024 tiles of four nodes)
    L, U = np.ascontiguousarray(L0[pick]), np.ascontiguousarray(U0[pick])
    om = orc.OracleModel(V, props)
    ref0 = om.consistency(L0, U0, None)
    ref = tuple(np.ascontiguousarray(r[pick]) for r in ref0[:4])
    return V, props, dom, L, U, ref


@pytest.mark.parametrize("static", [0, 1, 2, 3])
@pytest.mark.parametrize("tile", [2, 4])
def test_tickets_do_not_show_in_the_result(ctx, batch, static, tile):
    V, props, dom, L, U, ref = batch
    ctx.set_model(V, props)
    ctx.set_hull(*dom)
    for k, v in {"neq_path": 1, "small_path": 0, "nodes_per_block": tile, "neq_dynamic": static, "neq_persist": 1}.items():
        ctx.set_option(k, v)
    try:
        N = L.shape[0]
        tiles = (N + tile - 1) // tile
        for rep in range(3):
            ctx.stats_reset()
            got = ctx.propagate_implicit(L, U, want_active=True)
            pl = ctx.last_plan()
            assert pl["path"] == 1 and pl["nodes_per_block"] == tile, pl
            assert pl["grid"] * max(static, 1) < tiles, (pl, tiles)  # (persistent, and with static > 0 the launch does draw tickets)
            assert_parity(ref, got[:4], f"tickets static={static} tile={tile} launch {rep}")
            assert got[4]["nodes"] == N, got[4]
            if rep == 1:  # a launch that is too small to draw: the words stay zero for the next one
                small = ctx.propagate_implicit(L[:33], U[:33], want_active=True)
                assert_parity(tuple(r[:33] for r in ref), small[:4], "small launch between two ticket launches")
    finally:
        for k, v in {"nodes_per_block": 0, "neq_dynamic": 2, "small_path": 1}.items():
            ctx.set_option(k, v)


def test_tickets_on_16_node_tiles_at_n1000():
    """The shape the tickets were built for: 16-node tiles of N-queens-1000, 512 persistent workgroups, two static tiles each
    and the rest drawn — 32 768 frontier nodes = 2048 tiles, 1024 of them dealt by ticket, 128 per residue of blockIdx.x & 7.  64 nodes against
    the oracle: from static tiles and from ticketed tiles of EVERY residue (so a residue that is never drawn, or drawn twice, shows), the rest of
    the batch against the same launch with the fixed stride (neq_dynamic 0), and a second launch on the same context (the words came back to zero)."""
    import torch
    from pcp_amd import model as M
    from pcp_amd import workloads as W
    n = 1000
    c = E.Context(0)
    try:
        props = M.nqueens_props(n)
        c.set_model(n, props)
        c.set_hull(1, n)
        nodes = 32768
        L, U, _ = W.nqueens_frontier(c, n, nodes, share=0, shares=8, implicit=True)
        L, U = np.ascontiguousarray(L[:nodes]), np.ascontiguousarray(U[:nodes])
        dev = torch.device("cuda", 0)

        def launch(dynamic):
            c.set_option("neq_dynamic", dynamic)
            lb, ub = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
            st = torch.full((nodes,), 255, dtype=torch.uint8, device=dev)
            c.propagate_device(nodes, lb, ub, lb, ub, None, None, st)
            pl = c.last_plan()
            torch.cuda.synchronize()
            return lb.cpu().numpy(), ub.cpu().numpy(), st.cpu().numpy(), pl

        c.stats_reset()
        l2, u2, s2, pl = launch(2)
        assert pl["path"] == 1 and pl["nodes_per_block"] == 16 and pl["packed"] == 1, pl
        tiles, grid = nodes // 16, pl["grid"]
        assert tiles >= 3 * grid, pl  # persistent, >= 3 tiles per workgroup (MI355X: grid 512, four each): the third and later ones are drawn
        assert c.stats_read()["nodes"] == nodes  # every tile ran exactly once (a tile drawn twice would count its nodes twice)
        l0, u0, s0, _ = launch(0)
        assert np.array_equal(s2, s0) and np.array_equal(l2, l0) and np.array_equal(u2, u0)
        assert (s2 != 255).all()
        l2b, u2b, s2b, _ = launch(2)  # the tickets were left zero
        assert np.array_equal(s2b, s0) and np.array_equal(l2b, l0) and np.array_equal(u2b, u0)
        # 64 nodes against the oracle: 16 from static tiles, 6 from ticketed tiles of each residue (tile = 2 grid + 8 ticket + residue)
        rng = np.random.default_rng(11)
        pick = list(rng.integers(0, 2 * grid * 16, size=16))
        for r in range(8):
            cand = np.arange(2 * grid + r, tiles, 8)
            for t in rng.choice(cand, size=6, replace=False):
                pick.append(int(t) * 16 + int(rng.integers(0, 16)))
        pick = np.array(pick[:64])
        om = orc.OracleModel(n, props)
        ref = om.consistency(L[pick], U[pick], None, check_dup=False)
        assert_parity((ref[0], ref[1], None, ref[3]), (l2[pick], u2[pick], None, s2[pick]), "ticketed 16-node tiles at n = 1000")
    finally:
        c.set_option("neq_dynamic", 2)
        c.close()
