"""Host-side mirror of libpcp's model-building surface, lowered to the C-ABI's ``pcp_prop`` records.

The names, argument meaning and error behaviour follow the reference so that tests read like the
reference's own tests (paths relative to /root/reference/src/libpcp):

* views        ``Identity`` (term/identity.rs:47-70), ``Addition`` (term/addition.rs:80-110),
               ``Constant`` (term/constant.rs:43-68)
* propagators  ``XNeqY XEqY XLessY XLessYPlusZ XGreaterYPlusZ XEqYPlusZ XEqYMulZ`` (propagators/cmp/*.rs)
               and the constructor sugar ``x_greater_y x_geq_y x_leq_y x_geq_y_plus_z x_leq_y_plus_z``
               (propagators/cmp/mod.rs:34-86)
* globals      ``Distinct`` / ``join_distinct`` (propagators/distinct.rs:26-126), ``Conjunction``
               (logic/conjunction.rs:77-119)
* stores       ``VStore.alloc`` (variable/store.rs:129-141), ``CStore.alloc`` (propagation/store.rs:223-230)

Lowering: a view flattens to ``(var | PCP_CONST, offset)`` — ``Addition(Addition(x,a),b)`` = ``(x, a+b)``,
``Addition(Constant(c), a)`` = ``Constant(c+a)`` (both are exact: Interval ``+``/``-`` of a constant is a
translation, term/addition.rs:87,98).  This module is pure host logic: it never touches the oracle.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple, Union

import numpy as np

PCP_CONST = 0xFFFFFFFF
PCP_NOVAR = 0xFFFFFFFE
PCP_SUM = 0xC0000000
PCP_BOUND_MAX = 0x1FFFFFFF

NEQ, EQ, LT, LT3, GT3, EQ3, MUL3, BOOL, NBOOL = range(9)
KIND_NAMES = ["NEQ", "EQ", "LT", "LT3", "GT3", "EQ3", "MUL3", "BOOL", "NBOOL"]
F_LEAF, F_AND, F_OR = 0, 1, 2  # pcp_fnode_type
FNODE_DTYPE = np.dtype([("type", np.uint8), ("reserved", np.uint8), ("n_children", np.uint16), ("first", np.uint32)])
assert FNODE_DTYPE.itemsize == 8
FALSE, TRUE, UNKNOWN = 0, 1, 2

# Same layout as `pcp_prop` in include/pcp_hip.h (32 bytes).
PROP_DTYPE = np.dtype(
    [("kind", "u1"), ("group_kind", "u1"), ("reserved", "u2"), ("group", "u4"), ("var", "u4", (3,)), ("off", "i4", (3,))],
    align=True,
)
assert PROP_DTYPE.itemsize == 32


class ContractViolation(Exception):
    """Raised where the reference would panic (assert!)."""


# ----------------------------------------------------------------------------------------------- views
@dataclass(frozen=True)
class Identity:
    idx: int

    def flat(self) -> Tuple[int, int]:
        return (self.idx, 0)


@dataclass(frozen=True)
class Addition:
    x: "View"
    v: int

    def flat(self) -> Tuple[int, int]:
        var, off = self.x.flat()
        return (var, off + self.v)


@dataclass(frozen=True)
class Constant:
    value: int

    def flat(self) -> Tuple[int, int]:
        return (PCP_CONST, self.value)


@dataclass(frozen=True)
class Sum:
    """term::Sum (term/sum.rs:56-92): read = the interval sum of the members; an update through a Sum of several variables
    only has to overlap (no pruning).  Members may be Identity / Addition / Constant views: their variables become the term,
    their constants fold into the operand's offset.  Lowered through ``lower_units(..., sums_out=[...])``."""
    vars: Tuple["View", ...]

    def flat(self):
        raise ContractViolation("a Sum view is lowered by lower_units(..., sums_out=list)")


View = Union[Identity, Addition, Constant, Sum]


# ------------------------------------------------------------------------------------------ propagators
@dataclass(frozen=True)
class Elementary:
    kind: int
    ops: Tuple[View, ...]

    def rows(self) -> List[Tuple[int, Tuple[View, ...]]]:
        return [(self.kind, self.ops)]


def XNeqY(x: View, y: View) -> Elementary:
    return Elementary(NEQ, (x, y))


def XEqY(x: View, y: View) -> Elementary:
    return Elementary(EQ, (x, y))


def XLessY(x: View, y: View) -> Elementary:
    return Elementary(LT, (x, y))


def XLessYPlusZ(x: View, y: View, z: View) -> Elementary:
    return Elementary(LT3, (x, y, z))


def XGreaterYPlusZ(x: View, y: View, z: View) -> Elementary:
    return Elementary(GT3, (x, y, z))


def XEqYPlusZ(x: View, y: View, z: View) -> Elementary:
    return Elementary(EQ3, (x, y, z))


def XEqYMulZ(x: View, y: View, z: View) -> Elementary:
    return Elementary(MUL3, (x, y, z))


# propagators/cmp/mod.rs:34-86
def x_greater_y(x: View, y: View) -> Elementary:
    return XLessY(y, x)


def x_geq_y(x: View, y: View) -> Elementary:
    return x_greater_y(Addition(x, 1), y)


def x_leq_y(x: View, y: View) -> Elementary:
    return XLessY(x, Addition(y, 1))


def x_geq_y_plus_z(x: View, y: View, z: View) -> Elementary:
    return XGreaterYPlusZ(Addition(x, 1), y, z)


def x_leq_y_plus_z(x: View, y: View, z: View) -> Elementary:
    return XLessYPlusZ(Addition(x, -1), y, z)


@dataclass(frozen=True)
class Conjunction:
    """logic/conjunction.rs:77-119 over elementary members: ONE unit."""

    fs: Tuple[Elementary, ...]
    group_kind: int = 1

    def rows(self):
        return [r for f in self.fs for r in f.rows()]


def Distinct(vars: Sequence[View]) -> Conjunction:
    """propagators/distinct.rs:63-83: conjunction of all i<j XNeqY, ONE unit."""
    if len(vars) == 0:
        raise ContractViolation("Variable array in `Distinct` must be non-empty.")
    fs = tuple(XNeqY(vars[i], vars[j]) for i in range(len(vars) - 1) for j in range(i + 1, len(vars)))
    return Conjunction(fs, group_kind=2)


def AllEqual(vars: Sequence[View]) -> Conjunction:
    """propagators/all_equal.rs:47-64: conjunction of XEqY(vars[i], vars[i+1]), ONE unit whose dependencies are the
    variables in order (all_equal.rs:95-102) — the same shape as Distinct's."""
    if len(vars) == 0:
        raise ContractViolation("Variable array in `AllEqual` must be non-empty.")
    fs = tuple(XEqY(vars[i], vars[i + 1]) for i in range(len(vars) - 1))
    return Conjunction(fs, group_kind=2)


# ----------------------------------------------------------------------------------------------- the reified layer (logic/)
def Boolean(var: View) -> Elementary:
    """logic/boolean.rs:111-140: the formula "var = 1" over a 0/1 variable (allocate it with vstore.alloc((0, 1)))."""
    return Elementary(BOOL, (var,))


def BooleanNeg(var: View) -> Elementary:
    """logic/boolean_neg.rs:71-96: "var = 0"."""
    return Elementary(NBOOL, (var,))


@dataclass(frozen=True)
class And:
    """logic::Conjunction over arbitrary formulas (logic/conjunction.rs:77-119): ONE unit, lowered as a formula tree."""
    fs: Tuple["FormulaT", ...]


@dataclass(frozen=True)
class Or:
    """logic::Disjunction (logic/disjunction.rs:78-141): ONE unit; a child is propagated only when all others are disentailed."""
    fs: Tuple["FormulaT", ...]


FormulaT = Union[Elementary, And, Or]


def not_(f: FormulaT) -> FormulaT:
    """NotFormula::not (logic/ops.rs:17-19), applied when the formula is built, as the reference does."""
    if isinstance(f, And):
        return Or(tuple(not_(g) for g in f.fs))          # De Morgan, conjunction.rs:70-74
    if isinstance(f, Or):
        return And(tuple(not_(g) for g in f.fs))         # disjunction.rs:68-75
    if isinstance(f, Conjunction):
        return Or(tuple(not_(g) for g in f.fs))
    k, o = f.kind, f.ops
    if k == NEQ:
        return XEqY(o[0], o[1])                          # x_neq_y.rs:61-63
    if k == EQ:
        return XNeqY(o[0], o[1])                         # x_eq_y.rs:62-64
    if k == LT:
        return x_geq_y(o[0], o[1])                       # x_less_y.rs:62-64
    if k == LT3:
        return x_geq_y_plus_z(o[0], o[1], o[2])          # x_less_y_plus_z.rs:66-72
    if k == GT3:
        return x_leq_y_plus_z(o[0], o[1], o[2])          # x_greater_y_plus_z.rs:66-72
    if k == BOOL:
        return BooleanNeg(o[0])                          # boolean.rs:74-76
    if k == NBOOL:
        return Boolean(o[0])                             # boolean_neg.rs:66-68
    raise ContractViolation("not implemented")           # XEqYPlusZ / XEqYMulZ: unimplemented!() (x_eq_y_plus_z.rs:74-76)


def implication(f: FormulaT, g: FormulaT) -> Or:
    """logic/mod.rs:30-36 — exactly the reference's shape, Disjunction[f, g.not()]."""
    return Or((f, not_(g)))


def equivalence(f: FormulaT, g: FormulaT) -> And:
    """logic/mod.rs:38-45."""
    return And((implication(f, g), implication(g, f)))


def lower_formula(f: FormulaT, n_vars: int, sums_out=None, max_depth: int = 8):
    """One formula unit as (nodes, leaves) for pcp_model_push_formula: nodes[0] is the root, the children of an inner node are
    consecutive (breadth-first layout)."""
    nodes, leaves = [], []
    queue = [(f, 0)]
    nodes.append(None)
    qi = 0
    while qi < len(queue):
        g, at = queue[qi]
        qi += 1
        depth = 0 if at == 0 else None
        if isinstance(g, Conjunction):
            g = And(tuple(g.fs))
        if isinstance(g, (And, Or)):
            if len(g.fs) == 0:
                raise ContractViolation("a Conjunction / Disjunction needs at least one child")
            first = len(nodes)
            nodes[at] = (F_AND if isinstance(g, And) else F_OR, len(g.fs), first)
            for c in g.fs:
                nodes.append(None)
                queue.append((c, len(nodes) - 1))
        else:
            nodes[at] = (F_LEAF, 0, len(leaves))
            leaves.append(g)

    def depth_of(i):
        t, n, first = nodes[i]
        return 1 if t == F_LEAF else 1 + max(depth_of(first + c) for c in range(n))
    if depth_of(0) > max_depth:
        raise ContractViolation(f"formula deeper than {max_depth} levels")
    nd = np.zeros(len(nodes), dtype=FNODE_DTYPE)
    for i, (t, n, first) in enumerate(nodes):
        nd[i]["type"], nd[i]["n_children"], nd[i]["first"] = t, n, first
    lv = _lower_rows([(e.kind, e.ops) for e in leaves], n_vars, sums_out, dedup_unit=False)
    return nd, lv


def is_formula_unit(u) -> bool:
    return isinstance(u, (And, Or))


# ----------------------------------------------------------------------------------------------- stores
class VStore:
    """variable::Store over Interval<i32> (VStoreFD, variable/mod.rs:36): just the domains."""

    def __init__(self):
        self.lb: List[int] = []
        self.ub: List[int] = []

    def alloc(self, dom: Tuple[int, int]) -> Identity:  # variable/store.rs:129-141
        lb, ub = int(dom[0]), int(dom[1])
        if lb > ub:
            raise ContractViolation("alloc: empty domain")
        if abs(lb) > PCP_BOUND_MAX or abs(ub) > PCP_BOUND_MAX:
            raise ContractViolation("bound outside +-PCP_BOUND_MAX")
        self.lb.append(lb)
        self.ub.append(ub)
        return Identity(len(self.lb) - 1)

    def __len__(self):
        return len(self.lb)

    def bounds(self) -> Tuple[np.ndarray, np.ndarray]:
        return np.array(self.lb, dtype=np.int32), np.array(self.ub, dtype=np.int32)


def interval_bits(lb, ub, set_words: int, base: int) -> np.ndarray:
    """Bitset words of IntervalSet::new(lb, ub) per entry (the C ABI's set-mode domains, include/pcp_hip.h): value v is bit
    (v - base) of the entry's `set_words` uint64 words.  lb/ub: arrays of any shape; result shape + (set_words,)."""
    lb = np.asarray(lb, np.int64)
    ub = np.asarray(ub, np.int64)
    if ((lb <= ub) & ((lb < base) | (ub >= base + 64 * set_words))).any():
        raise ContractViolation("domain outside the bitset universe [base, base + 64 * set_words)")
    w = np.arange(set_words, dtype=np.int64) * 64 + base  # first value of each word
    lo = np.clip(lb[..., None] - w, 0, 64)                # bits below lo are clear
    hi = np.clip(ub[..., None] - w + 1, 0, 64)            # bits from hi up are clear
    ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    def below(k):  # mask of the bits [0, k)
        k = k.astype(np.uint64)
        return np.where(k >= 64, ones, (np.uint64(1) << np.minimum(k, np.uint64(63))) - np.uint64(1))
    return np.where(hi > lo, below(hi) & ~below(lo), np.uint64(0)).astype(np.uint64)


def bits_bounds(bits: np.ndarray, base: int):
    """(lb, ub) of bitset domains [..., set_words]; an empty set gives (1, 0)."""
    bits = np.asarray(bits, np.uint64)
    sw = bits.shape[-1]
    flat = bits.reshape(-1, sw)
    lb = np.ones(flat.shape[0], np.int64)
    ub = np.zeros(flat.shape[0], np.int64)
    for i, row in enumerate(flat):
        nz = np.nonzero(row)[0]
        if len(nz):
            lo_w, hi_w = int(nz[0]), int(nz[-1])
            lo_b = (int(row[lo_w]) & -int(row[lo_w])).bit_length() - 1
            hi_b = int(row[hi_w]).bit_length() - 1
            lb[i], ub[i] = base + 64 * lo_w + lo_b, base + 64 * hi_w + hi_b
    return lb.reshape(bits.shape[:-1]).astype(np.int32), ub.reshape(bits.shape[:-1]).astype(np.int32)


class CStore:
    """The model part of propagation::store::Store: an append-only list of units."""

    def __init__(self):
        self.units: List[Union[Elementary, Conjunction]] = []

    def alloc(self, p: Union[Elementary, Conjunction]) -> int:  # propagation/store.rs:223-230
        self.units.append(p)
        return len(self.units) - 1

    def __len__(self):
        return len(self.units)

    def truncate(self, n_units: int):  # FrozenStore::restore, propagation/store.rs:319-323
        del self.units[n_units:]

    def lower(self, n_vars: int) -> np.ndarray:
        return lower_units(self.units, n_vars)

    def pushes(self, n_vars: int, sums_out=None):
        """The model as a sequence of pushes, in unit order: ("props", rows) for runs of elementary / Conjunction units,
        ("formula", nodes, leaves) for each formula unit (And / Or trees)."""
        out, run = [], []
        for u in self.units:
            if is_formula_unit(u):
                if run:
                    out.append(("props", lower_units(run, n_vars, sums_out=sums_out)))
                    run = []
                out.append(("formula",) + lower_formula(u, n_vars, sums_out))
            else:
                run.append(u)
        if run:
            out.append(("props", lower_units(run, n_vars, sums_out=sums_out)))
        return out


def join_distinct(vstore: VStore, cstore: CStore, vars: Sequence[View]) -> None:
    """propagators/distinct.rs:26-45: the same pairs as standalone units."""
    if len(vars) == 0:
        raise ContractViolation("Variable array in `Distinct` must be non-empty.")
    for i in range(len(vars) - 1):
        for j in range(i + 1, len(vars)):
            cstore.alloc(XNeqY(vars[i], vars[j]))


def _flat_operand(op, sums_out):
    """(var code, offset) of a view; Sum views are appended to sums_out (their members' variable indices) and referred to as
    PCP_SUM | term.  Also returns the variables the operand subscribes to."""
    extra = 0
    while isinstance(op, Addition) and isinstance(_base(op), Sum):
        extra += op.v
        op = op.x
    if isinstance(op, Sum):
        if sums_out is None:
            raise ContractViolation("the model has Sum views: pass sums_out=[] to lower_units and give it to set_model(..., sums=)")
        if len(op.vars) == 0:
            raise ContractViolation("At least one variable in sum.")
        members, const = [], 0
        for m in op.vars:
            var, off = m.flat()
            const += off
            if var != PCP_CONST:
                members.append(var)
        if not members:
            return PCP_CONST, const + extra, []
        sums_out.append(members)
        return PCP_SUM | (len(sums_out) - 1), const + extra, members
    var, off = op.flat()
    return var, off, ([] if var == PCP_CONST else [var])


def _base(op):
    while isinstance(op, Addition):
        op = op.x
    return op


def lower_units(units: Sequence[Union[Elementary, Conjunction]], n_vars: int, gid_base: int = 0, sums_out=None) -> np.ndarray:
    """Flatten units to pcp_prop rows.  Checks what the reference would panic on.  `group` = gid_base + the unit's position
    (the engine only compares it between consecutive rows of one push)."""
    n_rows = sum(max(1, len(u.rows())) for u in units)
    out = np.zeros(n_rows, dtype=PROP_DTYPE)
    out["var"][:] = PCP_NOVAR
    r = 0
    for gid, u in enumerate(units):
        rows = u.rows()
        grouped = isinstance(u, Conjunction)
        if grouped and not rows:
            # An empty conjunction (Distinct over one variable, distinct.rs:170) is entailed at its first
            # evaluation: lower it to one trivially true member, 0 != 1.
            rows = [(NEQ, (Constant(0), Constant(1)))]
        seen_unit = set()
        for kind, ops in rows:
            rec = out[r]
            rec["kind"] = kind
            rec["group_kind"] = u.group_kind if grouped else 0
            rec["group"] = gid_base + gid
            seen = set()
            for k, op in enumerate(ops):
                var, off, deps = _flat_operand(op, sums_out)
                if abs(off) > PCP_BOUND_MAX:
                    raise ContractViolation("offset outside +-PCP_BOUND_MAX")
                for dv in deps:
                    if not (0 <= dv < n_vars):
                        raise ContractViolation(f"variable {dv} is not in the vstore (size {n_vars})")
                    if dv in seen:
                        # the reactor would panic: "propagator already subscribed to this variable"
                        # (propagation/reactors/indexed_deps.rs:69-77)
                        raise ContractViolation("propagator already subscribed to this variable")
                    seen.add(dv)
                rec["var"][k] = var
                rec["off"][k] = off
            seen_unit |= seen
            r += 1
    return out[:r]


def _lower_rows(rows, n_vars: int, sums_out, dedup_unit: bool = True) -> np.ndarray:
    """pcp_prop rows of a formula's leaves (group fields unused).  A variable twice in ONE leaf is the reactor's panic; across
    leaves it is fine (a formula's dependencies are the de-duplicated union, conjunction.rs:107-118, disjunction.rs:119-131)."""
    out = np.zeros(len(rows), dtype=PROP_DTYPE)
    out["var"][:] = PCP_NOVAR
    for r, (kind, ops) in enumerate(rows):
        out[r]["kind"] = kind
        seen = set()
        for k, op in enumerate(ops):
            var, off, deps = _flat_operand(op, sums_out)
            if abs(off) > PCP_BOUND_MAX:
                raise ContractViolation("offset outside +-PCP_BOUND_MAX")
            for dv in deps:
                if not (0 <= dv < n_vars):
                    raise ContractViolation(f"variable {dv} is not in the vstore (size {n_vars})")
                if dv in seen:
                    raise ContractViolation("propagator already subscribed to this variable")
                seen.add(dv)
            out[r]["var"][k] = var
            out[r]["off"][k] = off
    return out


def push_model(target, cstore: "CStore", n_vars: int, set_words: int = 0):
    """Send a model with formula units to `target` — an engine Context or an OracleModel: both offer reset_model / push_sum /
    push_props / push_formula.  Sum views are registered first (their numbers are fixed by the lowering order)."""
    sums = []
    pushes = cstore.pushes(n_vars, sums_out=sums)
    target.reset_model(n_vars, set_words)
    for members in sums:
        target.push_sum(members)
    for p in pushes:
        if p[0] == "props":
            target.push_props(p[1])
        else:
            target.push_formula(p[1], p[2])
    return pushes


class Cumulative:
    """propagators/cumulative.rs:59-114 — the decomposition of Schutt et al.: for each task j, the resources of the tasks that
    overlap j's start must fit the capacity.  join() allocates, per ordered pair (j, i != j): a Boolean b_i, the unit
    b_i <=> (s_i <= s_j /\\ s_j < s_i + d_i), an intermediate r = b_i * r_i (XEqYMulZ), then c >= r_j + Sum(r).  Variable and
    propagator allocation order as in the reference (the tests pin statuses, not indices, but the order is kept)."""

    def __init__(self, starts, durations, resources, capacity):
        assert len(starts) == len(durations) == len(resources)
        self.starts, self.durations, self.resources, self.capacity = list(starts), list(durations), list(resources), capacity
        self.intermediate: List[List[int]] = []

    def join(self, vstore: VStore, cstore: CStore) -> None:
        tasks = len(self.starts)
        if tasks == 1:
            cstore.alloc(x_geq_y(self.capacity, self.resources[0]))       # c >= r[j]   (cumulative.rs:67-70)
            return
        for j in range(tasks):
            resource_vars = []
            self.intermediate.append([])
            for i in range(tasks):
                if i == j:
                    continue
                conj = And((x_leq_y(self.starts[i], self.starts[j]),                                   # s[i] <= s[j]
                            XLessYPlusZ(self.starts[j], self.starts[i], self.durations[i])))           # s[j] < s[i] + d[i]
                bi = vstore.alloc((0, 1))                                                                # Boolean::new (boolean.rs:38-43)
                cstore.alloc(equivalence(Boolean(bi), conj))
                ri = self.resources[i]
                ri_ub = ri.value if isinstance(ri, Constant) else vstore.ub[ri.flat()[0]] + ri.flat()[1]
                r = vstore.alloc((0, ri_ub))
                self.intermediate[-1].append(r.idx)
                cstore.alloc(XEqYMulZ(r, bi, ri))                                                        # r = bi * r[i]
                resource_vars.append(r)
            cstore.alloc(x_geq_y_plus_z(self.capacity, self.resources[j], Sum(tuple(resource_vars))))  # c >= r[j] + sum


# ----------------------------------------------------------------------------------------------- models
def nqueens(n: int, distinct: str = "join") -> Tuple[VStore, CStore]:
    """example/src/nqueens.rs:28-50 with Interval<i32> domains: vars in [1,n]; for i<j
    q_i != q_j + (j-i), q_i != q_j - (j-i); then join_distinct (``distinct='join'``, what the example runs)
    or ONE Distinct unit (``distinct='global'``, the commented alternative at nqueens.rs:51)."""
    vs, cs = VStore(), CStore()
    queens = [vs.alloc((1, n)) for _ in range(n)]
    for i in range(n - 1):
        for j in range(i + 1, n):
            q1, q2 = i + 1, j + 1
            cs.alloc(XNeqY(queens[i], Addition(queens[j], q2 - q1)))
            cs.alloc(XNeqY(queens[i], Addition(queens[j], -q2 + q1)))
    if distinct == "join":
        join_distinct(vs, cs, queens)
    elif distinct == "global":
        cs.alloc(Distinct(queens))
    else:
        raise ValueError(distinct)
    return vs, cs


def nqueens_props(n: int) -> np.ndarray:
    """Vectorised lowering of ``nqueens(n, 'join')`` (identical rows, built without Python objects; n=1000 has
    1 498 500 units)."""
    i, j = np.triu_indices(n, k=1)  # row-major i<j order == the reference's nested loops
    m = i.shape[0]
    out = np.zeros(3 * m, dtype=PROP_DTYPE)
    out["kind"] = NEQ
    out["var"][:, 2] = PCP_NOVAR
    d = (j - i).astype(np.int32)
    # diagonals interleaved per pair: rows 2k, 2k+1
    out["var"][0 : 2 * m : 2, 0] = i
    out["var"][0 : 2 * m : 2, 1] = j
    out["off"][0 : 2 * m : 2, 1] = d
    out["var"][1 : 2 * m : 2, 0] = i
    out["var"][1 : 2 * m : 2, 1] = j
    out["off"][1 : 2 * m : 2, 1] = -d
    out["var"][2 * m :, 0] = i
    out["var"][2 * m :, 1] = j
    out["group"] = np.arange(3 * m, dtype=np.uint32)
    return out


def chained_lt(n: int) -> Tuple[VStore, CStore]:
    """X1 < X2 < ... < Xn, all in [1,10] (propagation/store.rs:362-375, commented-out test)."""
    vs, cs = VStore(), CStore()
    xs = [vs.alloc((1, 10)) for _ in range(n)]
    for i in range(n - 1):
        cs.alloc(XLessY(xs[i], xs[i + 1]))
    return vs, cs


def golomb(m: int = 10, length: int = 80):
    """BASELINE config 4: bound-consistent distinct + sum network, Golomb-ruler style.
    marks m_0..m_{m-1} in [0,length], m_0 = 0, m_i < m_{i+1}; differences d_ij in [1,length] with
    m_j = m_i + d_ij (XEqYPlusZ); ONE Distinct over all differences; symmetry d_01 < d_{m-2,m-1}."""
    vs, cs = VStore(), CStore()
    marks = [vs.alloc((0, length)) for _ in range(m)]
    cs.alloc(XEqY(marks[0], Constant(0)))
    for i in range(m - 1):
        cs.alloc(XLessY(marks[i], marks[i + 1]))
    diffs = {}
    for i in range(m - 1):
        for j in range(i + 1, m):
            d = vs.alloc((1, length))
            diffs[(i, j)] = d
            cs.alloc(XEqYPlusZ(marks[j], marks[i], d))
    cs.alloc(Distinct(list(diffs.values())))
    cs.alloc(XLessY(diffs[(0, 1)], diffs[(m - 2, m - 1)]))
    return vs, cs
