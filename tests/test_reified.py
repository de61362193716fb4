"""The reified layer (logic/: Boolean, BooleanNeg, Disjunction, Conjunction, implication, equivalence) and
propagators::Cumulative — SURVEY.md §8 f4.

Golden vectors: tests/golden/cumulative_kats.json = the reference's only LIVE full-`consistency()` tests
(propagators/cumulative.rs:254-319), transcribed: `Store::is_subsumed` before, `consistency()` after, `is_subsumed` after.
CPU part: the oracle (structure-faithful Disjunction / Boolean / BooleanNeg classes, pcp_oracle_engine.inc) replays them.
GPU part (-m gpu): the HIP formula kernel (pcp_formula.hip, through pcp_model_push_formula) against the golden statuses and,
bit-exact, against the oracle on the same stores and on random formula stores."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M

from util import splitmix64

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = json.load(open(os.path.join(GOLDEN, "cumulative_kats.json")))["cases"]


def cumulative_store(case):
    """CumulativeTest::instantiate (cumulative.rs:201-233): starts, durations, resources (Constant views for singletons when
    `constant`), the capacity variable, then Cumulative::join."""
    vs, cs = M.VStore(), M.CStore()
    def mk(dom):
        return M.Constant(dom[0]) if (case["constant"] and dom[0] == dom[1]) else vs.alloc(tuple(dom))
    starts = [mk(d) for d in case["starts"]]
    durations = [mk(d) for d in case["durations"]]
    resources = [mk(d) for d in case["resources"]]
    capacity = vs.alloc(tuple(case["capacity"]))
    M.Cumulative(starts, durations, resources, capacity).join(vs, cs)
    return vs, cs


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_replays_the_cumulative_tests(case):
    vs, cs = cumulative_store(case)
    om = orc.OracleModel(len(vs))
    M.push_model(om, cs, len(vs))
    assert om.n_units == len(cs)
    lb, ub = vs.bounds()
    assert om.is_subsumed(lb, ub) == case["before"]
    r = om.consistency(lb[None], ub[None], None)
    assert int(r[3][0]) == case["after"]
    if case["after"] != M.FALSE:
        assert om.is_subsumed(r[0][0], r[1][0]) == case["after"]


def test_formula_lowering_shapes():
    """equivalence(b, s_i <= s_j /\\ s_j < s_i + d_i) is Conjunction[Disjunction[b, Disjunction[not, not]], Disjunction[Conjunction[..], not b]]
    (logic/mod.rs:30-45 with the De Morgan negation of conjunction.rs:70-74): 10 nodes, 6 leaves, 4 levels."""
    vs = M.VStore()
    si, sj, di, b = (vs.alloc((0, 9)) for _ in range(4))
    conj = M.And((M.x_leq_y(si, sj), M.XLessYPlusZ(sj, si, di)))
    f = M.equivalence(M.Boolean(b), conj)
    nodes, leaves = M.lower_formula(f, len(vs))
    assert len(nodes) == 11 and len(leaves) == 6
    assert nodes[0]["type"] == M.F_AND and nodes[0]["n_children"] == 2
    assert sorted(int(k) for k in leaves["kind"]) == sorted([M.BOOL, M.LT, M.GT3, M.LT, M.LT3, M.NBOOL])
    with pytest.raises(M.ContractViolation):
        M.not_(M.XEqYPlusZ(si, sj, di))  # unimplemented!() in the reference (x_eq_y_plus_z.rs:74-76)


# --------------------------------------------------------------------------------------------------------------- GPU
def random_formula_store(seed, n_vars=9, n_units=10, dom=(0, 6)):
    """Random stores of formula units over all leaf kinds: implications, equivalences, nested And / Or, plus plain units."""
    rng = splitmix64(seed)
    vs, cs = M.VStore(), M.CStore()
    xs = [vs.alloc(dom) for _ in range(n_vars)]
    bs = [vs.alloc((0, 1)) for _ in range(3)]

    def leaf():
        # (no Boolean here: under a top-level Conjunction, Boolean::propagate on a variable already 0 is a non-monotonic update —
        # the reference panics, boolean.rs:134-137 + variable/store.rs:153-156; Booleans enter as Disjunction children below)
        k = int(rng.integers(0, 6))
        if k == 5:
            k = 6
        a, b, c = (xs[i] for i in rng.choice(n_vars, size=3, replace=False))
        off = int(rng.integers(-2, 3))
        if k == 0:
            return M.XNeqY(a, M.Addition(b, off))
        if k == 1:
            return M.XEqY(a, M.Addition(b, off))
        if k == 2:
            return M.XLessY(a, M.Addition(b, off))
        if k == 3:
            return M.XLessYPlusZ(a, b, M.Addition(c, off))
        if k == 4:
            return M.XGreaterYPlusZ(a, b, c)
        if k == 5:
            return M.Boolean(bs[int(rng.integers(0, 3))])
        return M.XLessY(a, M.Constant(int(rng.integers(dom[0], dom[1] + 1))))

    def formula(depth):
        if depth == 0 or rng.random() < 0.3:
            return leaf()
        n = int(rng.integers(2, 4))
        kids = tuple(formula(depth - 1) for _ in range(n))
        if rng.random() < 0.5:
            return M.And(kids)
        if rng.random() < 0.4:
            b = bs[int(rng.integers(0, 3))]
            kids = kids + ((M.Boolean(b) if rng.random() < 0.5 else M.BooleanNeg(b)),)
        return M.Or(kids)

    for _ in range(n_units):
        u = rng.random()
        if u < 0.25:
            cs.alloc(leaf())
        elif u < 0.5:
            cs.alloc(M.implication(formula(1), leaf()))
        elif u < 0.7:
            cs.alloc(M.equivalence(M.Boolean(bs[int(rng.integers(0, 3))]), formula(1)))
        else:
            f = formula(2)
            cs.alloc(f if M.is_formula_unit(f) else M.Or((f, leaf())))
    return vs, cs


def random_boxes(seed, vs, n_nodes):
    rng = splitmix64(seed)
    lb0, ub0 = vs.bounds()
    L = np.tile(lb0, (n_nodes, 1)); U = np.tile(ub0, (n_nodes, 1))
    for k in range(n_nodes):
        for v in range(len(lb0)):
            if rng.random() < 0.5:
                a = int(rng.integers(lb0[v], ub0[v] + 1)); b = int(rng.integers(a, ub0[v] + 1))
                L[k, v], U[k, v] = a, b
    return L.astype(np.int32), U.astype(np.int32)


@pytest.fixture(scope="module")
def ctx():
    import pcp_amd.engine as E
    c = E.Context(0)
    yield c
    c.close()


def gpu_vs_oracle(ctx, vs, cs, L, U, what):
    import pcp_amd.engine as E
    from util import assert_parity
    om = orc.OracleModel(len(vs))
    M.push_model(om, cs, len(vs))
    M.push_model(ctx, cs, len(vs))
    assert ctx.n_units == om.n_units == len(cs)
    ref = om.consistency(L, U, None)
    got = ctx.propagate(L, U, E.full_active(L.shape[0], om.n_units))   # explicit unit-level rows
    has_formula = any(M.is_formula_unit(u) or (isinstance(u, M.Elementary) and u.kind >= M.BOOL) for u in cs.units)
    assert (ctx.last_plan()["path"] == 3) == has_formula
    assert_parity(ref[:4], got[:4], what + " [explicit]")
    got_i = ctx.propagate_implicit(L, U)                                # implicit nodes, rows on request
    assert_parity(ref[:4], got_i[:4], what + " [implicit]")
    return ref, got


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hip_replays_the_cumulative_tests(ctx, case):
    vs, cs = cumulative_store(case)
    lb, ub = vs.bounds()
    ref, got = gpu_vs_oracle(ctx, vs, cs, lb[None], ub[None], case["name"])
    assert int(got[3][0]) == case["after"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_random_formula_stores(ctx, seed):
    vs, cs = random_formula_store(9000 + seed)
    L, U = random_boxes(9100 + seed, vs, 200)
    ref, _ = gpu_vs_oracle(ctx, vs, cs, L, U, f"formula store seed={seed}")
    assert len(ref[3]) == 200


@pytest.mark.gpu
def test_cumulative_with_open_starts(ctx):
    """Four tasks with open start windows: 12 equivalences, 12 XEqYMulZ, 4 sums — many nodes, narrowing through the disjunctions."""
    vs, cs = M.VStore(), M.CStore()
    starts = [vs.alloc((0, 6)) for _ in range(4)]
    durations = [M.Constant(d) for d in (3, 2, 4, 2)]
    resources = [M.Constant(r) for r in (2, 1, 2, 1)]
    cap = vs.alloc((3, 3))
    M.Cumulative(starts, durations, resources, cap).join(vs, cs)
    L, U = random_boxes(77, vs, 300)
    ref, _ = gpu_vs_oracle(ctx, vs, cs, L, U, "cumulative 4 tasks")
    assert (ref[3] == 0).any() and (ref[3] != 0).any()


@pytest.mark.gpu
def test_formula_contract(ctx):
    import pcp_amd.engine as E
    ctx.reset_model(3)
    nodes = np.zeros(2, dtype=M.FNODE_DTYPE)
    nodes[0] = (M.F_OR, 0, 1, 1); nodes[1] = (M.F_LEAF, 0, 0, 0)
    leaves = M.lower_units([M.XNeqY(M.Identity(0), M.Identity(1))], 3)
    ctx.push_formula(nodes, leaves)
    bad = nodes.copy(); bad[0]["first"] = 0  # a child in front of its parent
    with pytest.raises(E.PcpError):
        ctx.push_formula(bad, leaves)
    two = M.lower_units([M.XNeqY(M.Identity(0), M.Identity(1)), M.XNeqY(M.Identity(1), M.Identity(2))], 3)
    with pytest.raises(E.PcpError):
        ctx.push_formula(nodes, two)      # a leaf nobody uses
    ctx.truncate(0)
    assert ctx.n_units == 0


@pytest.mark.gpu
def test_device_dfs_on_a_cumulative_model(ctx):
    """pcp_dfs_device on a store with formula units: every step must propagate the node on TOP of the stack — the
    formula kernel used to run row 0 whatever the stack pointer said, so left subtrees were silently skipped.  Whole search (first
    solution, then all solutions) == the oracle's DFS over the same model, counter for counter."""
    vs, cs = M.VStore(), M.CStore()
    starts = [vs.alloc((0, 5)) for _ in range(4)]
    durations = [M.Constant(d) for d in (3, 2, 2, 1)]
    resources = [M.Constant(r) for r in (2, 1, 2, 1)]
    cap = vs.alloc((3, 3))
    M.Cumulative(starts, durations, resources, cap).join(vs, cs)
    V = len(vs)
    om = orc.OracleModel(V)
    M.push_model(om, cs, V)
    M.push_model(ctx, cs, V)
    lb0, ub0 = vs.bounds()
    ss1, _, _, sol1 = om.search(lb0, ub0, all_solutions=False)
    one = ctx.dfs_device(lb0, ub0, 100000, capacity=256, stop_on_solution=True, chunk=32)
    assert ctx.last_plan()["path"] == 3
    assert ss1["num_nodes"] > 3
    assert (one["nodes"], one["solutions"], one["failed"], one["error"]) == (ss1["num_nodes"], ss1["num_solution"], ss1["num_failed_node"], 0)
    if ss1["num_solution"]:
        assert np.array_equal(one["first_solution"], sol1)
    ssa, _, _, _ = om.search(lb0, ub0, all_solutions=True, node_limit=3000)
    al = ctx.dfs_device(lb0, ub0, 100000, capacity=256, stop_on_solution=False, node_limit=3000, chunk=97)
    assert (al["nodes"], al["solutions"], al["failed"]) == (ssa["num_nodes"], ssa["num_solution"], ssa["num_failed_node"])
    assert ssa["num_failed_node"] > 0 and ssa["num_solution"] > 0


@pytest.mark.gpu
def test_formula_at_the_depth_limit(ctx):
    """A unit nested to the eight levels lower_formula accepts (alternating Or / And down to a leaf at level 8), next to plain
    units: the kernel's tree walk at its deepest."""
    rng = splitmix64(4242)
    vs, cs = M.VStore(), M.CStore()
    xs = [vs.alloc((0, 7)) for _ in range(10)]
    bs = [vs.alloc((0, 1)) for _ in range(4)]

    def leaf():
        a, b = (xs[i] for i in rng.choice(10, size=2, replace=False))
        k = int(rng.integers(0, 3))
        off = int(rng.integers(-2, 3))
        return (M.XNeqY, M.XEqY, M.XLessY)[k](a, M.Addition(b, off))

    def nest(depth):  # a tree of exactly `depth` levels
        if depth == 1:
            return leaf()
        side = M.Boolean(bs[int(rng.integers(0, 4))]) if depth % 2 == 0 else leaf()
        kids = (nest(depth - 1), side) if rng.random() < 0.5 else (side, nest(depth - 1))
        return M.Or(kids) if depth % 2 == 0 else M.And(kids)

    for _ in range(6):
        f = nest(8)
        nodes, _ = M.lower_formula(f, len(vs))
        assert len(nodes) >= 15
        cs.alloc(f)
    for _ in range(4):
        cs.alloc(leaf())
    with pytest.raises(M.ContractViolation):
        M.lower_formula(M.Or((nest(8), leaf())), len(vs))  # nine levels: refused by the host mirror
    L, U = random_boxes(4243, vs, 300)
    ref, _ = gpu_vs_oracle(ctx, vs, cs, L, U, "depth-8 formulas")
    assert (ref[3] == 0).any() and (ref[3] != 0).any()


def _run_host_example(name, args):
    import subprocess
    import __graft_entry__ as g
    g.build()
    exe = os.path.join(g.ROOT, "pcp_amd", "host", "examples", name)
    return json.loads(subprocess.run([exe, *[str(a) for a in args]], check=True, capture_output=True, text=True).stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_cpp_host_twin_replays_the_cumulative_tests(case):
    """The 14 cumulative vectors through the C++ host twin (pcp_amd/host/pcp_host.hpp: Cumulative::join over Boolean / Conjunction /
    Disjunction / implication / equivalence / XEqYMulZ / Sum, lowered by the header itself to pcp_model_push_formula): the golden status,
    and the oracle's fixpoint bit for bit on the same store built by the Python mirror (same allocation order, so the same variable indices)."""
    args = [int(case["constant"]), len(case["starts"])]
    for key in ("starts", "durations", "resources"):
        for d in case[key]:
            args += list(d)
    args += list(case["capacity"])
    out = _run_host_example("cumulative", args)
    vs, cs = cumulative_store(case)
    assert out["vars"] == len(vs) and out["units"] == len(cs)
    assert out["status"] == case["after"]
    om = orc.OracleModel(len(vs))
    M.push_model(om, cs, len(vs))
    lb, ub = vs.bounds()
    r = om.consistency(lb[None], ub[None], None)
    assert int(r[3][0]) == out["status"]
    if out["status"] != M.FALSE:
        assert out["lb"] == [int(v) for v in r[0][0]] and out["ub"] == [int(v) for v in r[1][0]]


@pytest.mark.gpu
def test_cpp_host_twin_logic_layer():
    """implication / equivalence / not_ of the C++ host twin against the same formulas built by the Python mirror and run by the oracle;
    the second step of each pair allocs one more propagator on the narrowed store, as a search would."""
    out = _run_host_example("cumulative", ["logic-test"])

    def oracle_steps(doms, units_steps):
        vs, cs = M.VStore(), M.CStore()
        vars_ = [vs.alloc(d) for d in doms]
        lb, ub = vs.bounds()
        res = []
        for mk in units_steps:
            cs.alloc(mk(*vars_))
            om = orc.OracleModel(len(vs))
            M.push_model(om, cs, len(vs))
            r = om.consistency(lb[None], ub[None], None)
            res.append((int(r[3][0]), [int(v) for v in r[0][0]], [int(v) for v in r[1][0]]))
            lb, ub = r[0][0].copy(), r[1][0].copy()
        return res
    want = oracle_steps([(0, 9), (0, 9), (0, 1)], [lambda x, y, b: M.equivalence(M.Boolean(b), M.XLessY(x, y)),
                                                   lambda x, y, b: M.XEqY(b, M.Constant(1))])
    want += oracle_steps([(5, 9), (0, 7)], [lambda x, y: M.implication(M.XLessY(x, y), M.XLessY(M.Addition(x, 3), y)),
                                            lambda x, y: M.not_(M.x_geq_y(x, y))])
    assert len(out) == 4
    for got, (st, lb, ub) in zip(out, want):
        assert got["status"] == st, (got, st)
        if st != M.FALSE:
            assert got["lb"] == lb and got["ub"] == ub, (got, lb, ub)
    assert out[1]["status"] != M.FALSE and out[1]["ub"][0] == 8 and out[1]["lb"][1] == 1  # b = 1 propagated x < y
