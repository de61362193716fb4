"""150 000 nodes of a small model in one call (many tiles per CU, ragged last tile), 3000 of them checked against the oracle."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pcp_amd.engine as E
from pcp_amd import model as M
from oracle import oracle as orc
from util import random_nodes, random_active, assert_parity
n = 16
props = M.nqueens_props(n)
ctx = E.Context(0); ctx.set_model(n, props)
N = 150000
lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
rng = np.random.default_rng(5)
L = np.tile(lb0, (N, 1)); U = np.tile(ub0, (N, 1))
idx = rng.integers(0, n, size=N); val = rng.integers(1, n + 1, size=N)
L[np.arange(N), idx] = val; U[np.arange(N), idx] = val
act = E.full_active(N, len(props))
got = ctx.propagate(L, U, act)
om = orc.OracleModel(n, props)
sel = rng.choice(N, size=3000, replace=False)
ref = om.consistency(L[sel], U[sel], act[sel])
assert_parity(ref[:4], tuple(g[sel] if g is not None else None for g in got[:4]), "150k nodes")
print("ok", N, "nodes; statuses", np.bincount(got[3], minlength=3), got[4]["nodes"])
