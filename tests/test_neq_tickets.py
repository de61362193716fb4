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
