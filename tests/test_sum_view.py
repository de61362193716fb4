"""term::Sum (term/sum.rs:56-92) as a propagator operand: read = the interval sum of the members, an update through a Sum of
several variables never narrows (it only has to overlap), a Sum of one variable forwards, the propagator depends on every
member.  CPU: the oracle's Sum (restated in oracle/pcp_oracle_engine.inc) on hand-derived cases; GPU: the HIP engine against it."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M

from util import assert_parity, random_active, random_nodes, splitmix64

I = M.Identity


def lower(units, V):
    sums = []
    return M.lower_units(units, V, sums_out=sums), sums


def run_oracle(units, doms):
    V = len(doms)
    props, sums = lower(units, V)
    om = orc.OracleModel(V, props, sums)
    lb = np.array([[d[0] for d in doms]], np.int32)
    ub = np.array([[d[1] for d in doms]], np.int32)
    rl, ru, act, st, _ = om.consistency(lb, ub, None)
    return list(zip(rl[0].tolist(), ru[0].tolist())), int(st[0]), act


def test_sum_semantics_in_the_oracle():
    # sum.rs:66-69: with several members the update only has to overlap: nothing is pruned although a + b < 10 could prune
    doms, st, _ = run_oracle([M.XLessY(M.Sum((I(0), I(1))), M.Constant(10))], [(0, 8), (0, 8)])
    assert doms == [(0, 8), (0, 8)] and st == M.UNKNOWN
    # ... and fails when the shrunk sum is empty: a + b in [4, 10] < 3
    doms, st, _ = run_oracle([M.XLessY(M.Sum((I(0), I(1))), M.Constant(3))], [(2, 5), (2, 5)])
    assert st == M.FALSE
    # entailed through the sum's bounds: [4, 10] < 11
    doms, st, act = run_oracle([M.XLessY(M.Sum((I(0), I(1))), M.Constant(11))], [(2, 5), (2, 5)])
    assert st == M.TRUE and int(act[0, 0]) == 0
    # sum.rs:63-64: one member forwards the update: 5 < a
    doms, st, _ = run_oracle([M.XLessY(M.Constant(5), M.Sum((I(0),)))], [(0, 9)])
    assert doms == [(6, 9)] and st == M.TRUE
    # the capacity constraint of cumulative.rs:104-108: c >= r + sum(a, b): c and r are narrowed from the sum's bounds
    doms, st, _ = run_oracle([M.x_geq_y_plus_z(I(0), I(1), M.Sum((I(2), I(3))))], [(0, 10), (2, 9), (1, 4), (3, 5)])
    assert doms == [(6, 10), (2, 6), (1, 4), (3, 5)] and st == M.UNKNOWN
    # a member's change wakes the propagator (sum.rs:85-91): 2 < a raises the sum's lower bound, which raises c's
    doms, st, _ = run_oracle([M.x_geq_y_plus_z(I(0), I(1), M.Sum((I(2), I(3)))), M.XLessY(M.Constant(2), I(2))], [(0, 10), (2, 9), (1, 4), (3, 5)])
    assert doms == [(8, 10), (2, 4), (3, 4), (3, 5)]
    # constants and offsets of the members fold into the operand: sum(a + 1, 2, b) = a + b + 3
    doms, st, _ = run_oracle([M.XLessY(M.Constant(12), M.Sum((M.Addition(I(0), 1), M.Constant(2), I(1))))], [(0, 5), (0, 5)])
    assert st == M.UNKNOWN and doms == [(0, 5), (0, 5)]
    doms, st, _ = run_oracle([M.XLessY(M.Constant(13), M.Sum((M.Addition(I(0), 1), M.Constant(2), I(1))))], [(0, 5), (0, 5)])
    assert st == M.FALSE


def test_sum_contract():
    with pytest.raises(M.ContractViolation):  # the same variable twice in one propagator (indexed_deps.rs:69-77)
        lower([M.XLessY(I(0), M.Sum((I(0), I(1))))], 2)
    with pytest.raises(M.ContractViolation):
        M.lower_units([M.XLessY(I(0), M.Sum((I(1), I(2))))], 3)  # no sums_out


def random_sum_csp(seed, V, P):
    """Random inequalities / equalities in which one operand is a Sum of 2..4 other variables; a planted solution satisfies
    every constraint, so consistent nodes exist and cascades run through the sums."""
    rng = splitmix64(seed)
    sol = rng.integers(0, 21, size=V)
    units = []
    for _ in range(P):
        k = int(rng.integers(2, 5))
        vs = rng.choice(V, size=k + 2, replace=False)
        members = [int(v) for v in vs[:k]]
        sm_val = int(sol[members].sum())
        sm = M.Sum(tuple(I(v) for v in members))
        a, b = int(vs[k]), int(vs[k + 1])
        slack = int(rng.integers(0, 4))
        kind = int(rng.integers(0, 6))
        if kind == 0:    # a + c < sum
            units.append(M.XLessY(M.Addition(I(a), sm_val - int(sol[a]) - 1 - slack), sm))
        elif kind == 1:  # sum < a + c
            units.append(M.XLessY(sm, M.Addition(I(a), sm_val - int(sol[a]) + 1 + slack)))
        elif kind == 2:  # a + c >= b + sum
            units.append(M.x_geq_y_plus_z(M.Addition(I(a), int(sol[b]) + sm_val - int(sol[a]) + slack), I(b), sm))
        elif kind == 3:  # sum + c < a + b
            units.append(M.XLessYPlusZ(M.Addition(sm, int(sol[a]) + int(sol[b]) - sm_val - 1 - slack), I(a), I(b)))
        elif kind == 4:  # a + c = sum
            units.append(M.XEqY(M.Addition(I(a), sm_val - int(sol[a])), sm))
        else:            # sum != a + c
            units.append(M.XNeqY(sm, M.Addition(I(a), sm_val - int(sol[a]) + 1)))
    for _ in range(2 * P):  # plain binary filler so that cascades reach the members
        x, y = (int(t) for t in rng.choice(V, size=2, replace=False))
        units.append(M.XLessY(I(x), M.Addition(I(y), int(sol[x]) - int(sol[y]) + 1 + int(rng.integers(0, 3)))))
    return units, sol


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(5))
def test_sum_views_on_the_gpu(seed):
    import pcp_amd.engine as E
    V, P, N = 30 + 5 * seed, 25 + 10 * seed, 70
    units, sol = random_sum_csp(40 + seed, V, P)
    props, sums = lower(units, V)
    assert (props["var"] >= M.PCP_SUM).any() and len(sums) >= P
    om = orc.OracleModel(V, props, sums)
    lb0, ub0 = np.zeros(V, np.int32), np.full(V, 20, np.int32)
    L, U = random_nodes(50 + seed, lb0, ub0, N // 2, sol, p_narrow=0.25)       # consistent nodes: long cascades through the sums
    L2, U2 = random_nodes(55 + seed, lb0, ub0, N - N // 2, None, p_narrow=0.05)  # arbitrary boxes: failures
    L, U = np.concatenate([L, L2]), np.concatenate([U, U2])
    act = random_active(60 + seed, N, om.n_units, p_off=0.1)
    ref = om.consistency(L, U, act)
    assert (ref[3] == 0).any() and (ref[3] == 2).any() and ((ref[0] != L) | (ref[1] != U)).sum() > N
    ctx = E.Context(0)
    ctx.set_model(V, props, sums=sums)
    for opts in ({}, {"nodes_per_block": 8}, {"force_path": 2, "team": 3}, {"global_dom": 1}):
        for k, v in {"force_path": 0, "nodes_per_block": 0, "team": 0, "global_dom": 0, **opts}.items():
            ctx.set_option(k, v)
        got = ctx.propagate(L, U, act)
        assert_parity(ref[:4], got[:4], f"sum views seed={seed} {opts}")
        gi = ctx.propagate_implicit(L, U)
        assert_parity(om.consistency(L, U, None)[:4], gi[:4], f"sum views implicit seed={seed} {opts}")
    ctx.close()
