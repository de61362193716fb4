"""Pins the CPU oracle (oracle/) against the known-answer vectors transcribed from the reference's own
#[test] tables (tests/golden/*.json; provenance in each suite's "source")."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M

K = {"F": M.FALSE, "T": M.TRUE, "U": M.UNKNOWN}
EV = {"A": 0, "B": 1, "I": 2}

CTORS = {
    "XNeqY": M.XNeqY, "XEqY": M.XEqY, "XLessY": M.XLessY, "XLessYPlusZ": M.XLessYPlusZ,
    "XGreaterYPlusZ": M.XGreaterYPlusZ, "XEqYPlusZ": M.XEqYPlusZ, "XEqYMulZ": M.XEqYMulZ,
    "x_greater_y": M.x_greater_y, "x_geq_y": M.x_geq_y, "x_leq_y": M.x_leq_y,
}


def _view(op):
    if "c" in op:
        return M.Constant(op["c"])
    v = M.Identity(op["v"])
    return M.Addition(v, op["o"]) if "o" in op else v


def build_unit(suite, case):
    ctor = case.get("ctor", suite["ctor"])
    n = len(case["doms"])
    ops = [_view(o) for o in case["ops"]] if "ops" in case else [M.Identity(i) for i in range(n)]
    if ctor == "Distinct":
        return M.Distinct(ops)
    if ctor == "AllEqual":
        return M.AllEqual(ops)
    return CTORS[ctor](*ops)


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def kat_cases(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    data = load(golden_dir, "propagator_kats.json")
    for s in data["suites"]:
        for c in s["cases"]:
            yield pytest.param(s, c, id=f"{s['name'].split(' ')[0]}-{c['n']}")


@pytest.mark.parametrize("suite,case", list(kat_cases()))
def test_propagator_kat(suite, case):
    unit = build_unit(suite, case)
    n = len(case["doms"])
    props = M.lower_units([unit], n)
    lb = [d[0] for d in case["doms"]]
    ub = [d[1] for d in case["doms"]]
    r = orc.kat(n, lb, ub, props)
    assert r["before"] == K[case["before"]], "is_subsumed before"
    assert r["ok"] == case["ok"], "propagate() result"
    if case["ok"]:
        assert r["delta"] == [(v, EV[e]) for v, e in case["delta"]], "drained delta"
    assert r["after"] == K[case["after"]], "is_subsumed after"
    if "final" in case:
        assert [[int(a), int(b)] for a, b in zip(r["lb"], r["ub"])] == case["final"]


def _dom(x):
    return (1, 0) if x == "empty" else tuple(x)


def test_store_update_events(golden_dir):
    g = load(golden_dir, "engine_kats.json")
    for c in g["store_update"]["cases"]:
        ok, ev = orc.vstore_update(_dom(c["from"]), _dom(c["to"]))
        assert ok == c["ok"], c
        assert ev == (EV[c["events"][0]] if c["events"] else -1), c


def test_store_panics(golden_dir):
    g = load(golden_dir, "engine_kats.json")
    for c in g["store_panics"]["cases"]:
        if "alloc" in c:
            with pytest.raises(M.ContractViolation):
                M.VStore().alloc((1, 0))
            with pytest.raises(orc.OraclePanic):
                orc.OracleModel(1, M.lower_units([], 1)).consistency([[1]], [[0]])
        elif c["panic"]:
            with pytest.raises(orc.OraclePanic):
                orc.vstore_update(_dom(c["from"]), _dom(c["to"]))
        else:
            ok, _ = orc.vstore_update(_dom(c["from"]), _dom(c["to"]))
            assert ok == c["ok"]


def test_shrink_and_intersection(golden_dir):
    g = load(golden_dir, "engine_kats.json")
    for c in g["shrink"]["cases"]:
        new = orc.interval_op(c["op"], c["from"], c["arg"])
        if c["expect"] == "empty":
            assert new[0] > new[1]
        else:
            assert list(new) == c["expect"]
        ok, ev = orc.vstore_update(tuple(c["from"]), new)
        assert ok == c["ok"], c
        assert ev == (EV[c["events"][0]] if c["events"] else -1), c
    for c in g["intersection"]["cases"]:
        new = orc.interval_op("intersection", c["a"], c["b"][0], c["b"][1])
        if c["expect"] == "empty":
            assert new[0] > new[1]
        else:
            assert list(new) == c["expect"]
        got = []
        for i, d in enumerate((c["a"], c["b"])):
            ok, ev = orc.vstore_update(tuple(d), new)
            assert ok == c["ok"]
            if ev >= 0:
                got.append([i, "ABI"[ev]])
        if c["ok"]:
            assert got == c["delta"]


def _run_reactor(ops, r):
    for op in ops:
        if op[0] == "is_empty":
            assert r.is_empty() == op[1]
        elif op[0] == "subscribe":
            r.subscribe(op[1], EV[op[2]], op[3])
        elif op[0] == "unsubscribe":
            r.unsubscribe(op[1], EV[op[2]], op[3])
        elif op[0] == "react":
            assert r.react(op[1], EV[op[2]]) == op[3], op


def test_reactor_tables(golden_dir):
    g = load(golden_dir, "engine_kats.json")["reactor"]
    _run_reactor(g["subscribe_test"], orc.Reactor(3))
    _run_reactor(g["unsubscribe_test"], orc.Reactor(3))
    for p in g["panics"]:
        with pytest.raises(orc.OraclePanic):
            _run_reactor(p["ops"], orc.Reactor(3))


def _run_fifo(ops, f):
    for op in ops:
        if op[0] == "schedule":
            f.schedule(op[1])
        elif op[0] == "unschedule":
            f.unschedule(op[1])
        elif op[0] == "pop":
            assert f.pop() == op[1], op
        elif op[0] == "is_empty":
            assert f.is_empty() == op[1]


def test_scheduler_tables(golden_dir):
    g = load(golden_dir, "engine_kats.json")["scheduler"]
    _run_fifo(g["schedule_test"], orc.Fifo(3))
    _run_fifo(g["unschedule_test"], orc.Fifo(3))
    for p in g["panics"]:
        with pytest.raises(orc.OraclePanic):
            _run_fifo(p["ops"], orc.Fifo(3))


def _status(vs, cs, check_dup=True):
    lb, ub = vs.bounds()
    m = orc.OracleModel(len(vs), cs.lower(len(vs)))
    _, _, _, st, _ = m.consistency(lb[None, :], ub[None, :], check_dup=check_dup)
    return int(st[0])


def test_engine_basic(golden_dir):
    g = load(golden_dir, "engine_kats.json")["engine"]
    vs, cs = M.VStore(), M.CStore()
    exp = [K[s["expect"]] for s in g["basic_test"]["steps"]]
    assert _status(vs, cs) == exp[0]
    v1, v2, v3 = vs.alloc((1, 4)), vs.alloc((1, 4)), vs.alloc((1, 1))
    assert _status(vs, cs) == exp[1]
    cs.alloc(M.XLessY(v1, v2))
    assert _status(vs, cs) == exp[2]
    cs.alloc(M.XEqY(v1, v3))
    assert _status(vs, cs) == exp[3]


def test_engine_chained_lt(golden_dir):
    g = load(golden_dir, "engine_kats.json")["engine"]
    for n, exp in g["chained_lt"]["cases"]:
        vs, cs = M.chained_lt(n)
        assert _status(vs, cs) == K[exp], n


def test_engine_nqueens_root(golden_dir):
    g = load(golden_dir, "engine_kats.json")["engine"]
    for n, exp in g["nqueens_root"]["cases"]:
        vs, cs = M.VStore(), M.CStore()
        q = [vs.alloc((1, n)) for _ in range(n)]
        for i in range(n - 1):
            for j in range(i + 1, n):
                q1, q2 = i + 1, j + 1
                cs.alloc(M.XNeqY(M.Addition(q[i], q1), M.Addition(q[j], q2)))
                cs.alloc(M.XNeqY(q[i], M.Addition(q[j], -q2 + q1)))
        cs.alloc(M.Distinct(q))
        assert _status(vs, cs) == K[exp], n


@pytest.mark.parametrize("distinct", ["join", "global"])
def test_search_all_solutions(golden_dir, distinct):
    g = load(golden_dir, "engine_kats.json")["search"]
    for n, count in enumerate(g["all_solutions"]["counts"], start=1):
        if n > 8 and distinct == "global":
            continue
        vs, cs = M.nqueens(n, distinct)
        lb, ub = vs.bounds()
        m = orc.OracleModel(n, cs.lower(n))
        ss, _, _, _ = m.search(lb, ub, all_solutions=True)
        assert ss["num_solution"] == count, n
        assert ss["end_of_search"] == 1


def test_search_one_solution(golden_dir):
    g = load(golden_dir, "engine_kats.json")["search"]
    for n_s, status in g["one_solution"]["status"].items():
        n = int(n_s)
        vs, cs = M.nqueens(n, "global")
        lb, ub = vs.bounds()
        m = orc.OracleModel(n, cs.lower(n))
        ss, _, _, sol = m.search(lb, ub, all_solutions=False)
        assert (ss["num_solution"] == 1) == (status == "Satisfiable"), n
        if status == "Satisfiable":
            # a real placement: all columns distinct, no shared diagonal
            assert len(set(sol)) == n
            assert len({int(sol[i]) + i for i in range(n)}) == n and len({int(sol[i]) - i for i in range(n)}) == n


def test_search_stop_node(golden_dir):
    g = load(golden_dir, "engine_kats.json")["search"]["stop_node"]
    vs, cs = M.nqueens(g["n"], "global")
    lb, ub = vs.bounds()
    m = orc.OracleModel(g["n"], cs.lower(g["n"]))
    ss, _, _, _ = m.search(lb, ub, all_solutions=True, node_limit=g["limit"])
    assert ss["num_nodes"] == g["expect_nodes"] and ss["end_of_search"] == 1


def test_branching_tables(golden_dir):
    g = load(golden_dir, "engine_kats.json")["search"]
    root = g["binary_split"]["root"]
    for c in g["binary_split"]["cases"]:
        lb, ub = root[c["var"]]
        v = orc.middle_val(lb, ub)
        assert [[lb, v], [v + 1, ub]] == c["children"]
    for c in g["first_smallest_var"]["cases"]:
        assert orc.first_smallest_var(c["vars"]) == c["expect"]
    with pytest.raises(orc.OraclePanic):
        orc.first_smallest_var(g["first_smallest_var"]["panic"]["vars"])


def test_lowering_matches_vectorised_nqueens():
    for n in (2, 5, 9):
        vs, cs = M.nqueens(n, "join")
        a = cs.lower(n)
        b = M.nqueens_props(n)
        assert a.tobytes() == b.tobytes()


def test_mirror_contract_checks():
    vs, cs = M.VStore(), M.CStore()
    x = vs.alloc((0, 3))
    cs.alloc(M.XLessY(x, M.Addition(x, 1)))
    with pytest.raises(M.ContractViolation):
        cs.lower(1)  # same variable twice -> reactor panic in the reference
    cs2 = M.CStore()
    cs2.alloc(M.XNeqY(x, M.Identity(7)))
    with pytest.raises(M.ContractViolation):
        cs2.lower(1)
