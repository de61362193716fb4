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

NEQ, EQ, LT, LT3, GT3, EQ3, MUL3 = range(7)
KIND_NAMES = ["NEQ", "EQ", "LT", "LT3", "GT3", "EQ3", "MUL3"]
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
