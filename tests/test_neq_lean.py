"""The lean round 0 of the all-XNeqY kernel (pcp_neq.hip: neq_fast_load / neq_fast_test, round 6): full 16-node tiles in 16-bit cells whose nodes
have at most four assigned variables between them.  The benchmarked frontier takes its common branch (nothing narrows in 97 % of the tiles); this
file drives the others on small dense models where they are the rule: tiles in which nodes narrow (quiet re-passes), in which a narrowing ASSIGNS a
variable (the tile is handed to the general rounds), in which nodes fail, tiles with one to four listed variables with partial node masks, chains of
bounds that run into several forbidden values in a row (one re-pass per value), and tiles with five assigned variables (the general path, next to
lean tiles in the same launch).  Every launch is compared with the oracle (Store::consistency, propagation/store.rs:125-164, 247-257; XNeqY
x_neq_y.rs:66-104) AND with the same launch with the lean form switched off (`neq_debug` 131072): status and domains, bit-exact."""
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


def dense_neq(seed, V, dom, per_pair=3, p_pair=0.9, max_off=3):
    """x != y + c over many pairs, several offsets per pair (the shape of N-queens: a bound that loses a value often meets the next forbidden one)."""
    rng = splitmix64(seed)
    rows = []
    for x in range(V):
        for y in range(x + 1, V):
            if rng.random() < p_pair:
                for c in rng.choice(np.arange(-max_off, max_off + 1), size=per_pair, replace=False):
                    rows.append((x, y, int(c)))
    props = np.zeros(len(rows), dtype=M.PROP_DTYPE)
    props["var"][:] = M.PCP_NOVAR
    props["group"] = np.arange(len(rows))
    props["kind"] = M.NEQ
    for r, (x, y, c) in enumerate(rows):
        props[r]["var"][0], props[r]["var"][1], props[r]["off"][1] = x, y, c
    return props


def tiles(seed, V, dom, n_tiles, max_assigned, p_short=0.25):
    """16-node tiles; in each, up to `max_assigned` variables are assigned in random subsets of the tile's nodes, values near the others' bounds so
    that filters act; some other variables are narrowed to short intervals (so that a narrowing can assign them) or to bounds next to forbidden values."""
    rng = splitmix64(seed)
    N = 16 * n_tiles
    L = np.full((N, V), dom[0], np.int32)
    U = np.full((N, V), dom[1], np.int32)
    for t in range(n_tiles):
        k = int(rng.integers(0, max_assigned + 1))
        vs = rng.choice(V, size=k, replace=False)
        for v in vs:
            mask = rng.random(16) < (1.0 if rng.random() < 0.5 else 0.5)
            val = int(rng.integers(dom[0], dom[1] + 1))
            for b in np.nonzero(mask)[0]:
                n = 16 * t + b
                L[n, v] = U[n, v] = val if rng.random() < 0.8 else int(rng.integers(dom[0], dom[1] + 1))
        for b in range(16):
            n = 16 * t + b
            for v in range(V):
                if v in vs:
                    continue
                u = rng.random()
                if u < p_short:  # a short interval: two or three values (a narrowing may assign it: the general rounds take over)
                    a = int(rng.integers(dom[0], dom[1])); L[n, v], U[n, v] = a, min(dom[1], a + int(rng.integers(1, 3)))
                elif u < p_short + 0.25:  # a wide interval with a bound inside the band of forbidden values
                    a = int(rng.integers(dom[0], dom[0] + 6)); L[n, v] = a
                    U[n, v] = int(rng.integers(max(a + 3, dom[1] - 6), dom[1] + 1))
    return L, U


SEEN = {"neq_lean": 0, "neq_lean_passes": 0, "neq_lean_handover": 0, "general": 0}


@pytest.mark.parametrize("seed,V,max_assigned,p_short", [(1, 24, 1, 0.02), (2, 24, 2, 0.25), (3, 40, 4, 0.02), (4, 40, 6, 0.25), (5, 64, 3, 0.1)])
def test_lean_round_zero_branches(ctx, seed, V, max_assigned, p_short):
    dom = (0, 20)
    props = dense_neq(100 + seed, V, dom)
    om = orc.OracleModel(V, props)
    L, U = tiles(200 + seed, V, dom, 40, max_assigned, p_short)
    ref = om.consistency(L, U, None)
    ctx.set_model(V, props)
    ctx.set_hull(*dom)
    got = {}
    try:
        for dbg in (0, 131072):
            for k, v in {"neq_path": 1, "small_path": 0, "nodes_per_block": 16, "neq_debug": dbg}.items():
                ctx.set_option(k, v)
            ctx.stats_reset()
            g = ctx.propagate_implicit(L, U, want_active=True)
            pl = ctx.last_plan()
            assert pl["path"] == 1 and pl["nodes_per_block"] == 16 and pl["packed"] == 1, pl
            assert_parity(ref[:4], g[:4], f"lean round 0 seed {seed} neq_debug {dbg}")
            got[dbg] = g
            dc = ctx.debug_counters()
            if dbg == 0:  # which branches ran (summed over the cases: test_every_branch_ran)
                assert dc["neq_tiles"] == 40 and dc["neq_lean"] > 10 and dc["neq_lean_passes"] + dc["neq_lean_handover"] > 0, dc
                assert (dc["neq_lean"] < 40) == (max_assigned > 4), dc
                for k in ("neq_lean", "neq_lean_passes", "neq_lean_handover"):
                    SEEN[k] += dc[k]
                SEEN["general"] += 40 - dc["neq_lean"]
            else:
                assert dc["neq_lean"] == 0, dc
        assert np.array_equal(got[0][3], got[131072][3])
        # the branches the file is about did occur: nodes that narrowed, nodes that failed, nodes left open
        st = ref[3]
        changed = ((ref[0] != L) | (ref[1] != U)).any(axis=1) & (st != 0)
        assert changed.sum() > 20 and ((st == 0).sum() > 5 or max_assigned < 2) and (st == 2).sum() > 20, (int(changed.sum()), np.bincount(st, minlength=3))
        # ... including narrowings that assigned a variable (the hand-over to the general rounds)
        newly = (((ref[0] == ref[1]) & (L != U)).any(axis=1) & (st != 0)).sum()
        assert newly > (5 if p_short >= 0.1 else 0), int(newly)
    finally:
        for k, v in {"nodes_per_block": 0, "neq_debug": 0, "small_path": 1}.items():
            ctx.set_option(k, v)


def test_every_branch_ran():
    """Over the cases above: lean tiles, quiet re-passes, hand-overs to the general rounds, and general tiles next to lean ones (pcp_debug_counters)."""
    assert SEEN["neq_lean"] > 100 and SEEN["neq_lean_passes"] > 10 and SEEN["neq_lean_handover"] > 10 and SEEN["general"] > 0, SEEN
