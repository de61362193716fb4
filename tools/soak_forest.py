#!/usr/bin/env python
"""Randomised soak of the device-side search loops against the oracle's DFS (complete trees, so node / solution / failure counts are
order-independent and must match exactly):
  set mode  — pcp_dfs_forest_device_set (setdfs_kernel: node in LDS, undo trail, refill between launches): random mixed-kind CSPs over
              IntervalSet domains; one tree from the root, and forests below a breadth-first frontier of random depth, launches of a
              random number of nodes, refill on and off;
  intervals — pcp_dfs_forest_device (neqfix_kernel<.., DFS>): random all-XNeqY models, expansion + forest of random width.
usage: python tools/soak_forest.py [seconds] [seed0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pcp_amd.engine as E
from pcp_amd import model as M
from pcp_amd.search_device import DeviceSearch
from pcp_amd.search_forest import forest_search
from oracle import oracle as orc
from util import random_csp
from test_neq_path import neq_model

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = E.Context(0)
t0, it, n_set, n_int, big = time.time(), 0, 0, 0, 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + it)
    it += 1
    if rng.random() < 0.6:
        # ---- set mode ----
        # domains within [0, ..]: for a domain like [-3, -2] the reference's MiddleVal, (lower + upper) / 2 with Rust's truncating division,
        # is its UPPER bound, the left branch x <= value changes nothing and the reference's search (hence the oracle's) never ends
        V = int(rng.integers(4, 11)); hi = int(rng.integers(3, 9)); lo = int(rng.integers(0, 4)); hi += lo
        sw = 1
        kinds = [M.NEQ] if rng.random() < 0.3 else [M.NEQ, M.EQ, M.LT, M.LT3, M.GT3, M.EQ3][: int(rng.integers(2, 7))]
        props, _, _, _ = random_csp(seed0 + 3 * it, V, int(rng.integers(V, 3 * V)), planted=bool(rng.random() < 0.6), dom=(lo, hi), kinds=kinds)
        lb0, ub0 = np.full(V, lo, np.int32), np.full(V, hi, np.int32)
        ss, _, _, _ = orc.OracleModel(V, props).search_set(lb0, ub0, sw, lo, all_solutions=True, node_limit=200_001)
        if ss["num_nodes"] > 200_000:
            big += 1
            continue
        want = (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
        ctx.set_model(V, props, set_words=sw); ctx.set_hull(lo, hi)
        steps = int(rng.integers(1, 60))
        root = M.interval_bits(lb0, ub0, sw, lo)[None]
        r = ctx.dfs_forest_set(root, steps_per_launch=steps, trail_capacity=1 << 16, level_capacity=256)
        assert r["error"] == 0 and (r["nodes"], r["solutions"], r["failed"]) == want, ("set one tree", it, want, r)
        ds = DeviceSearch(ctx, batch=4096, capacity=16384, implicit=True)
        ds.reset(lb0, ub0, lo)
        done = False
        for _ in range(int(rng.integers(1, 6))):
            if ds.advance(all_solutions=True, max_rounds=1, keep_solutions=0):
                done = True
                break
        ds.compact()
        if not done and ds.size:
            st = ds.stats
            rest = (want[0] - st.num_nodes, want[1] - st.num_solution, want[2] - st.num_failed_node)
            for reb in (True, False):
                f = ctx.dfs_forest_set(ds.bits[:ds.size].clone(), steps_per_launch=steps, trail_capacity=1 << 16, level_capacity=256, rebalance=reb)
                assert f["error"] == 0 and (f["nodes"], f["solutions"], f["failed"]) == rest, ("set forest", it, reb, rest, f)
        n_set += 1
    else:
        # ---- intervals, all-XNeqY ----
        V = int(rng.integers(4, 10)); dom = (0, int(rng.integers(3, 7)))
        props = neq_model(seed0 + 5 * it, V, int(rng.integers(V, 3 * V)), dom)
        lb0, ub0 = np.full(V, dom[0], np.int32), np.full(V, dom[1], np.int32)
        ss, _, _, _ = orc.OracleModel(V, props).search(lb0, ub0, all_solutions=True, node_limit=200_001)
        if ss["num_nodes"] > 200_000:
            big += 1
            continue
        want = (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
        ctx.set_model(V, props)
        if rng.random() < 0.5:
            ctx.set_hull(dom[0], dom[1])
        ctx.set_option("neq_dfs_block", int(rng.choice([0, 256, 512])))
        r = forest_search(ctx, lb0, ub0, n_trees=int(rng.integers(1, 200)), steps_per_launch=int(rng.integers(1, 50)), capacity=512)
        assert r["error"] == 0 and (r["nodes"], r["solutions"], r["failed"]) == want, ("interval forest", it, want, r)
        n_int += 1
ctx.set_option("neq_dfs_block", 0)
print(f"forest soak ok: {n_set} set-mode models (one tree + forest with and without refill), {n_int} all-XNeqY interval models, {big} skipped as too large, in {time.time() - t0:.0f} s")
