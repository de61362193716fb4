"""An INDEPENDENT pin for the oracle's IntervalSet<i32> instantiation (task 5c).

The reference holds no vector that observes an IntervalSet after propagation, and the crate that implements it
(intervallum ^1.2.0) is not in the tree, so the oracle's set mode was "parity unpinned" below the search level.  This file is a
second, deliberately naive restatement written from the reference's own propagator sources over plain Python `set[int]` — every
domain operation spelled as a set comprehension, the engine as "run every propagator until nothing changes" — and checks the C++
oracle against it:
  * the set algebra (difference, shrink_left/right, intersection, shift, is_disjoint, is_subset) on random sets, operation by
    operation, and the event a change raises (propagation/events/mod.rs:51-69);
  * `consistency_set` (status, every domain) on random small CSPs over all six propagator kinds, by brute-force fixpoint.
The two restatements share no code; they agree on thousands of random cases.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M

from util import random_csp, splitmix64

CONST = M.PCP_CONST


def to_bits(s, sw, base):
    w = np.zeros(sw, np.uint64)
    for v in s:
        w[(v - base) >> 6] |= np.uint64(1) << np.uint64((v - base) & 63)
    return w


def to_set(w, base):
    return {base + 64 * k + b for k, x in enumerate(w) for b in range(64) if (int(x) >> b) & 1}


# ---- the naive model ----------------------------------------------------------------------------------------------------
def event(old, new):
    """MonotonicEvent::new (events/mod.rs:51-69): None if the size is unchanged, Assignment if the new domain is a singleton,
    Bound if a bound moved, else Inner."""
    assert new <= old
    if len(new) == len(old):
        return None
    if len(new) == 1:
        return "Assignment"
    if min(new) != min(old) or max(new) != max(old):
        return "Bound"
    return "Inner"


class PyStore:
    def __init__(self, doms):
        self.d = [set(x) for x in doms]
        self.failed = False

    def read(self, var, off):  # Identity / Addition / Constant views (term/*.rs)
        return {off} if var == CONST else {v + off for v in self.d[var]}

    def update(self, var, off, new):  # StoreMonotonicUpdate::update through the view; variable/store.rs:151-166
        if var == CONST:
            return len(new) > 0 and off in new  # term/constant.rs:49-52
        tgt = {v - off for v in new}
        assert tgt <= self.d[var], "Domain update must be monotonic."
        if not tgt:
            return False
        self.d[var] = tgt
        return True


def lt3(st, x, y, z, xo):  # x_less_y_plus_z.rs:105-119, x seen through Addition(x, xo)
    X, Y, Z = st.read(x[0], x[1] + xo), st.read(*y), st.read(*z)
    return (st.update(x[0], x[1] + xo, {v for v in X if v < max(Y) + max(Z)})
            and st.update(*y, {v for v in Y if v > min(X) - max(Z)})
            and st.update(*z, {v for v in Z if v > min(X) - max(Y)}))


def gt3(st, x, y, z, xo):  # x_greater_y_plus_z.rs:106-118
    X, Y, Z = st.read(x[0], x[1] + xo), st.read(*y), st.read(*z)
    return (st.update(x[0], x[1] + xo, {v for v in X if v > min(Y) + min(Z)})
            and st.update(*y, {v for v in Y if v < max(X) - min(Z)})
            and st.update(*z, {v for v in Z if v < max(X) - min(Y)}))


def propagate(st, kind, ops):
    x, y = ops[0], ops[1]
    if kind == M.NEQ:  # x_neq_y.rs:82-93
        X, Y = st.read(*x), st.read(*y)
        if len(X) == 1:
            return st.update(*y, Y - X)
        if len(Y) == 1:
            return st.update(*x, X - Y)
        return True
    if kind == M.EQ:  # x_eq_y.rs:102-107
        X, Y = st.read(*x), st.read(*y)
        n = X & Y
        return st.update(*x, n) and st.update(*y, n)
    if kind == M.LT:  # x_less_y.rs:104-109, both from the pre-read values
        X, Y = st.read(*x), st.read(*y)
        return st.update(*x, {v for v in X if v < max(Y)}) and st.update(*y, {v for v in Y if v > min(X)})
    z = ops[2]
    if kind == M.LT3:
        return lt3(st, x, y, z, 0)
    if kind == M.GT3:
        return gt3(st, x, y, z, 0)
    if kind == M.EQ3:  # x_eq_y_plus_z.rs:85-87 with cmp/mod.rs:62-86: (x + 1) > y + z  &&  (x - 1) < y + z
        return gt3(st, x, y, z, 1) and lt3(st, x, y, z, -1)
    raise AssertionError(kind)


def entailed(st, kind, ops):
    x, y = ops[0], ops[1]
    X, Y = st.read(*x), st.read(*y)
    if kind == M.NEQ:  # not XEqY::is_subsumed (x_neq_y.rs:71-73): True iff the SETS are disjoint (x_eq_y.rs:87-93)
        return not (X & Y)
    if kind == M.EQ:
        return min(X) == max(Y) and max(X) == min(Y)
    if kind == M.LT:
        return max(X) < min(Y)
    Z = st.read(*ops[2])
    if kind == M.LT3:
        return max(X) < min(Y) + min(Z)
    if kind == M.GT3:
        return min(X) > max(Y) + max(Z)
    return (min(X) + 1 > max(Y) + max(Z)) and (max(X) - 1 < min(Y) + min(Z))  # Kleene and of the two halves


def brute_fixpoint(doms, props):
    st = PyStore(doms)
    rows = [(int(p["kind"]), [(int(p["var"][i]), int(p["off"][i])) for i in range(2 if p["kind"] <= M.LT else 3)]) for p in props]
    changed = True
    while changed:
        before = [frozenset(s) for s in st.d]
        for kind, ops in rows:
            if not propagate(st, kind, ops):
                return M.FALSE, None
        changed = before != [frozenset(s) for s in st.d]
    return (M.TRUE if all(entailed(st, k, o) for k, o in rows) else M.UNKNOWN), st.d


# ---- the oracle against it -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(4))
def test_set_algebra_matches_python_sets(seed):
    rng = splitmix64(4000 + seed)
    sw, base = 2, -17
    universe = list(range(base, base + 64 * sw))
    for _ in range(400):
        s = {v for v in universe if rng.random() < rng.choice([0.1, 0.5, 0.9])} or {int(rng.choice(universe))}
        t = {v for v in universe if rng.random() < 0.4} or {int(rng.choice(universe))}
        a = int(rng.integers(base - 3, base + 64 * sw + 3))
        bs, bt = to_bits(s, sw, base), to_bits(t, sw, base)
        assert to_set(orc.set_op("difference", bs, base, a)[0], base) == s - {a}
        assert to_set(orc.set_op("shrink_left", bs, base, a)[0], base) == {v for v in s if v >= a}
        assert to_set(orc.set_op("shrink_right", bs, base, a)[0], base) == {v for v in s if v <= a}
        assert to_set(orc.set_op("intersection", bs, base, 0, bt)[0], base) == s & t
        assert orc.set_op("is_disjoint", bs, base, 0, bt)[1] == (not (s & t))
        assert orc.set_op("is_subset", bs, base, 0, bt)[1] == (s <= t)
        k = int(rng.integers(-5, 6))
        if all(base <= v + k < base + 64 * sw for v in s):
            assert to_set(orc.set_op("shift", bs, base, k)[0], base) == {v + k for v in s}


def test_event_classification_of_set_changes():
    assert event({1, 2, 3}, {1, 2, 3}) is None
    assert event({1, 2, 3}, {2}) == "Assignment"
    assert event({1, 2, 3, 4}, {1, 2, 4}) == "Inner"       # an interior hole: only sets raise it (events/mod.rs:57-64)
    assert event({1, 2, 3, 4}, {2, 3, 4}) == "Bound"
    assert event({1, 2, 3, 4}, {1, 2, 3}) == "Bound"


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("planted", [True, False])
def test_consistency_set_matches_brute_force_fixpoint(seed, planted):
    """Random CSPs over NEQ / EQ / LT / LT3 / GT3 / EQ3 with Addition offsets and Constant operands, domains with holes: the
    oracle's engine (IndexedDeps + RelaxedFifo over IntervalSet) against the naive fixpoint over Python sets."""
    V, P, N = 7 + seed % 5, 14 + 3 * (seed % 4), 24
    kinds = [M.NEQ, M.EQ, M.LT, M.LT3, M.GT3, M.EQ3]
    lo, hi = -3, 20
    props, lb, ub, sol = random_csp(5000 + seed, V, P, planted=planted, dom=(lo, hi), kinds=kinds)
    sw, base = 1, -8
    rng = splitmix64(6000 + seed)
    om = orc.OracleModel(V, props)
    n_false = n_checked = 0
    for _ in range(N):
        doms = []
        for v in range(V):
            s = {x for x in range(lo, hi + 1) if rng.random() < 0.6}
            if planted:
                s.add(int(sol[v]))
            doms.append(s or {int(rng.integers(lo, hi + 1))})
        bits = np.stack([to_bits(s, sw, base) for s in doms])[None]
        r = om.consistency_set(bits, base, None)
        st_ref, d_ref = brute_fixpoint(doms, props)
        assert int(r[4][0]) == st_ref, (seed, planted, int(r[4][0]), st_ref)
        n_checked += 1
        if st_ref == M.FALSE:
            n_false += 1
            continue
        for v in range(V):
            assert to_set(r[2][0, v], base) == d_ref[v], (seed, v)
            assert (int(r[0][0, v]), int(r[1][0, v])) == (min(d_ref[v]), max(d_ref[v]))
    assert n_checked == N
