#!/usr/bin/env python
"""Randomised soak of set mode (IntervalSet<i32> domains, pcp_set.hip) against the oracle: mixed-kind CSPs over sets with holes and
all-XNeqY models with assigned variables (the singleton shortcuts of implicit nodes), explicit rows and implicit nodes.
usage: python tools/soak_set.py [seconds] [seed0]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pcp_amd.engine as E
from pcp_amd import model as M
from util import random_active, random_csp
from test_set_mode import both_set, random_sets

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = E.Context(0)
t0, it = time.time(), 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 + it)
    V = int(rng.integers(5, 90)); P = int(rng.integers(V, 6 * V)); N = int(rng.integers(1, 70))
    lo = int(rng.integers(-40, 10)); hi = lo + int(rng.integers(4, 200))
    sw = (hi - lo) // 64 + 1
    planted = rng.random() < 0.6
    neq_only = rng.random() < 0.4
    kinds = [M.NEQ] if neq_only else [M.NEQ, M.EQ, M.LT, M.LT3, M.GT3, M.EQ3][: int(rng.integers(2, 7))]
    props, lb, ub, sol = random_csp(seed0 + 3 * it, V, P, planted=planted, dom=(lo, hi), kinds=kinds)
    bits = random_sets(seed0 + 5 * it, lb, ub, N, sw, lo, sol if planted else None, p_keep=float(rng.choice([0.7, 0.3])))
    if rng.random() < 0.6:  # assigned variables
        for i in range(N):
            k = int(rng.integers(1, V))
            vs = rng.choice(V, size=k, replace=False)
            vals = sol[vs] if (planted and rng.random() < 0.7) else rng.integers(lo, hi + 1, size=k)
            bits[i, vs] = M.interval_bits(vals, vals, sw, lo)
    act = random_active(seed0 + 7 * it, N, P, p_off=float(rng.uniform(0.0, 0.3))) if rng.random() < 0.7 else None
    both_set(ctx, V, props, bits, lo, (lo, hi), act, f"set soak it={it} V={V} P={P} N={N} kinds={kinds} planted={planted}")
    it += 1
print(f"set-mode soak ok: {it} models (explicit + implicit each) in {time.time() - t0:.0f} s")
