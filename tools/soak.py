#!/usr/bin/env python
"""Randomised soak of the packed / word-group paths and the wake-up rounds against the oracle, explicit rows and implicit nodes
(not a pytest: run it when the kernels change).
usage: python tools/soak.py [seconds] [seed0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pcp_amd.engine as E
from pcp_amd import model as M
from oracle import oracle as orc
from util import random_nodes, random_active, assert_parity

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = E.Context(0)
t0, it, checked = time.time(), 0, 0
paths = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}
cells_checked = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + it)
    n = int(rng.integers(20, 110))
    lo = int(rng.integers(-50, 50)); hi = lo + int(rng.integers(3, 120))
    ii, jj = np.triu_indices(n, 1)
    if rng.random() < 0.3:  # thin the pairs out: ragged x-blocks, words that straddle several blocks
        keep = rng.random(len(ii)) < rng.uniform(0.2, 0.9); ii, jj = ii[keep], jj[keep]
    P = len(ii)
    if P < 70:
        it += 1; continue
    sol = rng.integers(lo, hi + 1, size=n)
    kinds = [M.NEQ, M.LT, M.EQ][: int(rng.integers(1, 4))]
    kind = np.repeat(rng.choice(kinds, size=P // 32 + 1), 32)[:P]
    planted = rng.random() < 0.6
    d = np.where(kind == M.LT, sol[ii] - sol[jj] + 1 + rng.integers(0, 8, size=P), np.where(kind == M.EQ, sol[ii] - sol[jj], rng.integers(-9, 10, size=P)))
    if not planted:
        d = d + rng.integers(-3, 4, size=P)
    clash = (kind == M.NEQ) & (sol[ii] == sol[jj] + d)
    d[clash] += 1
    props = np.zeros(P, dtype=M.PROP_DTYPE); props["var"][:] = M.PCP_NOVAR; props["group"] = np.arange(P)
    props["kind"] = kind; props["var"][:, 0] = ii; props["var"][:, 1] = jj; props["off"][:, 1] = d
    lb = np.full(n, lo, np.int32); ub = np.full(n, hi, np.int32)
    N = int(rng.integers(1, 140))
    L, U = random_nodes(seed0 + 7 * it, lb, ub, N, sol if planted else None, p_narrow=float(rng.uniform(0.02, 0.5)))
    if rng.random() < 0.5:
        # many assigned variables (the deep-tile regime: long wake-up cascades, single-variable tails, forbidden-value walks)
        for i in range(N):
            k = int(rng.integers(1, max(2, n - 2)))
            vs = rng.choice(n, size=k, replace=False)
            vals = sol[vs] if (planted and rng.random() < 0.7) else rng.integers(lo, hi + 1, size=k)
            L[i, vs] = vals; U[i, vs] = vals
    act = random_active(seed0 + 11 * it, N, P, p_off=float(rng.uniform(0.0, 0.4)))
    om = orc.OracleModel(n, props)
    ref = om.consistency(L, U, act)
    ref_i = om.consistency(L, U, None)
    ctx.set_model(n, props)
    if rng.random() < 0.5:
        ctx.set_hull(lo, hi)
    for opts in ({"nodes_per_block": 16}, {"nodes_per_block": 8}, {"nodes_per_block": 16, "word_level": 0}, {"nodes_per_block": 32}, {}):
        for k, v in {"nodes_per_block": 0, "packed": 1, "word_level": 1, "small_path": 0, "block_threads": int(rng.choice([256, 512, 1024])), **opts}.items():
            ctx.set_option(k, v)
        ctx.set_option("solo_cascade", int(rng.integers(0, 2)))
        got = ctx.propagate(L, U, act)
        assert_parity(ref[:4], got[:4], f"soak it={it} n={n} P={P} N={N} kinds={kinds} planted={planted} {opts}")
        got = ctx.propagate_implicit(L, U)
        assert_parity(ref_i[:4], got[:4], f"soak it={it} n={n} P={P} N={N} kinds={kinds} planted={planted} {opts} [implicit]")
        checked += 2
    if N <= 8:  # the team geometry (one node per team of workgroups) and one-node blocks
        for opts in ({"force_path": 2}, {"force_path": 1, "nodes_per_block": 1}):
            for k, v in {"nodes_per_block": 0, "force_path": 0, "block_threads": 1024, **opts}.items():
                ctx.set_option(k, v)
            got = ctx.propagate_implicit(L, U)
            assert_parity(ref_i[:4], got[:4], f"soak it={it} n={n} P={P} N={N} kinds={kinds} planted={planted} {opts} [implicit]")
            checked += 1
        ctx.set_option("force_path", 0)
    # ---- the specialised implicit-node kernels, path drawn and ASSERTED: 1 = assignment-driven (all-XNeqY), 2 = 10-bit cells (any
    # binary model under a declared hull of <= 1024 values), 0 = the generic kernels under the same options
    want = int(rng.choice([0, 1, 2, 4]))
    for k, v in {"nodes_per_block": 0, "force_path": 0, "block_threads": 1024, "global_dom": 0, "neq_path": 1, "big_path": 1, "big_round": 0, "small_path": 1}.items():
        ctx.set_option(k, v)
    ctx.set_model(n, props)
    ctx.set_hull(lo, hi)
    small_ok = n <= 128 and P <= 2048
    if want == 2:
        ctx.set_option("global_dom", 2); ctx.set_option("big_round", int(rng.integers(0, 3)))
    elif want == 0:
        ctx.set_option("neq_path", 0); ctx.set_option("big_path", 0); ctx.set_option("small_path", 0)
    elif want == 4:
        ctx.set_option("neq_path", 0)
    got = ctx.propagate_implicit(L, U)
    path = ctx.last_plan()["path"]
    if want == 2:
        expect = 2
    elif want == 1:
        expect = 1 if (kinds == [M.NEQ] and (N >= 64 or not small_ok)) else (4 if small_ok else 0)
    elif want == 4:
        expect = 4 if small_ok else 0
    else:
        expect = 0
    assert path == expect, (path, expect, want, kinds, n, P, N)
    assert_parity(ref_i[:4], got[:4], f"soak it={it} n={n} P={P} N={N} kinds={kinds} planted={planted} path={path} [implicit]")
    if path == 4:  # the small-store kernel on the explicit rows too
        got = ctx.propagate(L, U, act)
        assert ctx.last_plan()["path"] == 4
        assert_parity(ref[:4], got[:4], f"soak it={it} n={n} P={P} N={N} kinds={kinds} planted={planted} path=4 [explicit]")
        checked += 1
    paths[path] += 1
    checked += 1
    if kinds == [M.NEQ]:
        # the same nodes resident as packed cells (pcp_device_batch.cell_format PCP_CELLS_PACKED16): any tile size, in or out of place
        import torch
        for k, v in {"neq_path": 1, "small_path": 1, "global_dom": 0, "nodes_per_block": int(rng.choice([0, 1, 2, 4, 8, 16]))}.items():
            ctx.set_option(k, v)
        dev = torch.device("cuda", ctx.device)
        tl, tu = torch.from_numpy(L).to(dev), torch.from_numpy(U).to(dev)
        cells = ctx.pack_rows(tl, tu)
        out = cells if rng.random() < 0.5 else torch.zeros_like(cells)
        st = torch.zeros(N, dtype=torch.uint8, device=dev)
        ctx.propagate_device(N, cells, None, out, None, None, None, st, cells=True)
        assert ctx.last_plan()["path"] == 1
        gl, gu = ctx.unpack_rows(out)
        torch.cuda.synchronize()
        st = st.cpu().numpy(); ok = st != 0
        assert np.array_equal(st, ref_i[3]) and np.array_equal(gl.cpu().numpy()[ok], ref_i[0][ok]) and np.array_equal(gu.cpu().numpy()[ok], ref_i[1][ok]), \
            f"soak it={it} n={n} P={P} N={N} [cells]"
        ctx.set_option("nodes_per_block", 0)
        cells_checked += 1
        checked += 1
    for k, v in {"global_dom": 0, "neq_path": 1, "big_path": 1, "big_round": 0, "small_path": 1}.items():
        ctx.set_option(k, v)
    if rng.random() < 0.3:  # path 3: a random store of formula units (the reified layer), explicit rows and implicit nodes
        from test_reified import random_formula_store, random_boxes
        vs, cs = random_formula_store(seed0 + 13 * it, n_vars=int(rng.integers(5, 14)), n_units=int(rng.integers(4, 24)), dom=(0, int(rng.integers(3, 9))))
        Lf, Uf = random_boxes(seed0 + 17 * it, vs, int(rng.integers(1, 200)))
        omf = orc.OracleModel(len(vs)); M.push_model(omf, cs, len(vs)); M.push_model(ctx, cs, len(vs))
        reff = omf.consistency(Lf, Uf, None)
        got = ctx.propagate(Lf, Uf, E.full_active(Lf.shape[0], omf.n_units))
        has_formula = any(M.is_formula_unit(u) or (isinstance(u, M.Elementary) and u.kind >= M.BOOL) for u in cs.units)
        assert (ctx.last_plan()["path"] == 3) == has_formula
        assert_parity(reff[:4], got[:4], f"soak it={it} formula store [explicit]")
        got = ctx.propagate_implicit(Lf, Uf)
        assert_parity(reff[:4], got[:4], f"soak it={it} formula store [implicit]")
        paths[3] += has_formula; checked += 2
    it += 1
print(f"soak ok: {it} models, {checked} launches checked in {time.time() - t0:.0f} s; implicit launches by asserted path {paths}; {cells_checked} launches on packed cells")
