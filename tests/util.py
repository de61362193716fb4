"""Shared helpers for the parity tests: seeded synthetic instances and the A.4 comparison."""
import numpy as np

from pcp_amd import model as M


def splitmix64(seed):
    """splitmix64 stream as a numpy Generator seed helper (fixed seeds)."""
    return np.random.Generator(np.random.PCG64(seed))


def random_csp(seed, n_vars, n_props, dom=(0, 30), p_const=0.1, p_tern=0.2, planted=True, kinds=None):
    """Random binary/ternary CSP `x ◇ y + c` / `x ◇ y + z + c`.  With planted=True every constraint is satisfied
    by a hidden solution, so root propagation never fails (long cascades); otherwise failures are common."""
    rng = splitmix64(seed)
    lo, hi = dom
    sol = rng.integers(lo, hi + 1, size=n_vars)
    props = np.zeros(n_props, dtype=M.PROP_DTYPE)
    props["var"][:] = M.PCP_NOVAR
    props["group"] = np.arange(n_props)
    for r in range(n_props):
        tern = rng.random() < p_tern and n_vars >= 3
        if kinds is not None:
            kind = int(rng.choice(kinds))
            tern = kind >= M.LT3
        elif tern:
            kind = int(rng.choice([M.LT3, M.GT3, M.EQ3]))
        else:
            kind = int(rng.choice([M.NEQ, M.EQ, M.LT], p=[0.35, 0.15, 0.5]))
        n = 3 if tern else 2
        vs = rng.choice(n_vars, size=n, replace=False)
        val = [int(sol[v]) for v in vs]
        ops = [(int(v), 0) for v in vs]
        # turn one non-first operand into a constant sometimes
        if rng.random() < p_const:
            k = int(rng.integers(1, n))
            ops[k] = (M.PCP_CONST, val[k] if planted else int(rng.integers(lo, hi + 1)))
        slack = int(rng.integers(0, 6))
        # choose the offset on operand 1 so that the planted solution satisfies the constraint
        rhs_rest = sum(val[1:])
        if kind in (M.NEQ,):
            off1 = int(rng.integers(-3, 4))
            if planted and val[0] == rhs_rest + off1:
                off1 += 1
        elif kind in (M.EQ, M.EQ3):
            off1 = val[0] - rhs_rest if planted else int(rng.integers(-3, 4))
        elif kind in (M.LT, M.LT3):
            off1 = val[0] - rhs_rest + 1 + slack if planted else int(rng.integers(-3, 4))
        else:  # GT3: x > y + z + off
            off1 = val[0] - rhs_rest - 1 - slack if planted else int(rng.integers(-3, 4))
        if ops[1][0] == M.PCP_CONST:
            ops[1] = (M.PCP_CONST, ops[1][1] + off1)  # Addition(Constant(c), off) == Constant(c + off)
        else:
            ops[1] = (ops[1][0], off1)
        props[r]["kind"] = kind
        for k, (v, o) in enumerate(ops):
            props[r]["var"][k] = v
            props[r]["off"][k] = o
    lb = np.full(n_vars, lo, np.int32)
    ub = np.full(n_vars, hi, np.int32)
    return props, lb, ub, sol


def random_nodes(seed, lb, ub, n_nodes, sol=None, p_narrow=0.3):
    """n_nodes random sub-boxes of (lb,ub).  If `sol` is given every box contains it (consistent nodes)."""
    rng = splitmix64(seed)
    V = lb.shape[0]
    L = np.tile(lb, (n_nodes, 1)).astype(np.int32)
    U = np.tile(ub, (n_nodes, 1)).astype(np.int32)
    for n in range(n_nodes):
        mask = rng.random(V) < p_narrow
        for v in np.nonzero(mask)[0]:
            a, b = int(L[n, v]), int(U[n, v])
            if sol is not None:
                na = int(rng.integers(a, int(sol[v]) + 1))
                nb = int(rng.integers(int(sol[v]), b + 1))
            else:
                na = int(rng.integers(a, b + 1))
                nb = int(rng.integers(na, b + 1))
            L[n, v], U[n, v] = na, nb
    return L, U


def random_active(seed, n_nodes, n_units, p_off=0.1):
    rng = splitmix64(seed)
    words = (n_units + 63) // 64
    bits = rng.random((n_nodes, words * 64)) >= p_off
    bits[:, n_units:] = False
    a = np.zeros((n_nodes, words), dtype=np.uint64)
    for w in range(words):
        chunk = bits[:, w * 64 : (w + 1) * 64]
        a[:, w] = (chunk.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
    return a


def assert_parity(ref, got, what=""):
    """status equal; for status != False also (lb,ub) and `active` bit-exact."""
    rlb, rub, ract, rst = ref
    glb, gub, gact, gst = got
    assert np.array_equal(rst, gst), f"{what}: status differs at nodes {np.nonzero(rst != gst)[0][:10]} ref={rst[rst != gst][:10]} got={gst[rst != gst][:10]}"
    ok = rst != 0
    assert np.array_equal(rlb[ok], glb[ok]), f"{what}: lb differs"
    assert np.array_equal(rub[ok], gub[ok]), f"{what}: ub differs"
    if ract is not None and gact is not None:
        assert np.array_equal(ract[ok], gact[ok]), f"{what}: active differs"


# BASELINE config-3 generators live with the other workloads (bench.py measures what the tests check)
from pcp_amd.workloads import planted_binary_csp, unit_narrowing_prefix  # noqa: E402,F401
