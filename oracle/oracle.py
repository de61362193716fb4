"""ctypes binding of the CPU oracle (oracle/libpcp_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this module; the
product package ``pcp_amd`` never does.  See oracle/pcp_oracle.hpp for what the oracle restates and how it
is pinned (tests/golden/*.json, transcribed from the reference's own #[test] tables).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpcp_oracle.so")

PROP_DTYPE = np.dtype(
    [("kind", "u1"), ("group_kind", "u1"), ("reserved", "u2"), ("group", "u4"), ("var", "u4", (3,)), ("off", "i4", (3,))],
    align=True,
)


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("steps", "pops", "narrowings", "nodes", "failed_nodes", "subscriptions")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OrcSearchStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("num_solution", "num_failed_node", "num_prune", "num_nodes")] + [("end_of_search", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OraclePanic(Exception):
    """The reference would have panicked (assert!) on this input."""


def _src_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in ("pcp_oracle.hpp", "pcp_oracle_engine.inc", "pcp_oracle_capi.cpp", "../include/pcp_hip.h"):
        with open(os.path.join(_HERE, f), "rb") as fh:
            h.update(fh.read() + b"\0")
    return h.hexdigest()


def build(force: bool = False) -> str:
    """Compile the oracle with g++ (oracle/Makefile) when it is missing or its sources changed (content hash)."""
    tag = _LIB_PATH + ".srchash"
    try:
        fresh = os.path.exists(_LIB_PATH) and open(tag).read().strip() == _src_hash()
    except OSError:
        fresh = False
    if force or not fresh:
        subprocess.run(["make", "-B", "-C", _HERE, "libpcp_oracle.so"], check=True, capture_output=True)
        with open(tag, "w") as f:
            f.write(_src_hash())
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_model_new.restype = C.c_void_p
        L.orc_model_new.argtypes = [C.c_uint32]
        L.orc_model_free.argtypes = [C.c_void_p]
        L.orc_model_push_props.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_model_push_sum.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_model_push_formula.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_is_subsumed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_model_n_units.argtypes = [C.c_void_p]
        L.orc_model_n_units.restype = C.c_uint32
        L.orc_consistency.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_kat.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p] + [C.c_void_p] * 6
        L.orc_vstore_update.argtypes = [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]
        L.orc_interval_op.argtypes = [C.c_int] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]
        L.orc_reactor_new.restype = C.c_void_p
        L.orc_reactor_new.argtypes = [C.c_uint32]
        L.orc_reactor_free.argtypes = [C.c_void_p]
        for f in (L.orc_reactor_subscribe, L.orc_reactor_unsubscribe):
            f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_reactor_react.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_reactor_is_empty.argtypes = [C.c_void_p]
        L.orc_fifo_new.restype = C.c_void_p
        L.orc_fifo_new.argtypes = [C.c_uint32]
        L.orc_fifo_free.argtypes = [C.c_void_p]
        L.orc_fifo_schedule.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_fifo_unschedule.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_fifo_pop.argtypes = [C.c_void_p]
        L.orc_fifo_pop.restype = C.c_int64
        L.orc_fifo_is_empty.argtypes = [C.c_void_p]
        L.orc_middle_val.argtypes = [C.c_int32, C.c_int32]
        L.orc_middle_val.restype = C.c_int32
        L.orc_first_smallest_var.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_first_smallest_var.restype = C.c_int64
        L.orc_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 9
        L.orc_consistency_set.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_search_set.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 9
        L.orc_search_set_root.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 9
        L.orc_set_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise OraclePanic(lib().orc_last_error().decode())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleModel:
    """A model held by the oracle: n_vars + pcp_prop rows (same rows as fed to the HIP engine)."""

    def __init__(self, n_vars: int, props: Optional[np.ndarray] = None, sums=None):
        self.n_vars = int(n_vars)
        self._h = lib().orc_model_new(self.n_vars)
        if props is None:  # built push by push (pcp_amd.model.push_model: models with formula units)
            self.n_units = self.words = 0
            return
        props = np.ascontiguousarray(props, dtype=PROP_DTYPE)
        try:
            for members in (sums or []):  # term::Sum views, numbered in order
                mv = np.ascontiguousarray(members, np.uint32)
                t = C.c_uint32()
                _check(lib().orc_model_push_sum(self._h, len(mv), _ptr(mv), C.byref(t)))
            _check(lib().orc_model_push_props(self._h, len(props), _ptr(props)))
        except Exception:
            lib().orc_model_free(self._h)
            self._h = None
            raise
        self.n_units = int(lib().orc_model_n_units(self._h))
        self.words = (self.n_units + 63) // 64

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_model_free(self._h)
            self._h = None

    # ---- the push interface pcp_amd.model.push_model drives (same calls as the engine's Context) --------------------------
    def reset_model(self, n_vars: int, set_words: int = 0):
        lib().orc_model_free(self._h)
        self.n_vars = int(n_vars)
        self._h = lib().orc_model_new(self.n_vars)
        self._refresh()

    def _refresh(self):
        self.n_units = int(lib().orc_model_n_units(self._h))
        self.words = (self.n_units + 63) // 64

    def push_sum(self, members):
        mv = np.ascontiguousarray(members, np.uint32)
        t = C.c_uint32()
        _check(lib().orc_model_push_sum(self._h, len(mv), _ptr(mv), C.byref(t)))
        return t.value

    def push_props(self, props):
        props = np.ascontiguousarray(props, dtype=PROP_DTYPE)
        _check(lib().orc_model_push_props(self._h, len(props), _ptr(props)))
        self._refresh()

    def push_formula(self, nodes, leaves):
        from pcp_amd.model import FNODE_DTYPE
        nodes = np.ascontiguousarray(nodes, dtype=FNODE_DTYPE)
        leaves = np.ascontiguousarray(leaves, dtype=PROP_DTYPE)
        _check(lib().orc_model_push_formula(self._h, len(nodes), _ptr(nodes), len(leaves), _ptr(leaves)))
        self._refresh()

    def is_subsumed(self, lb, ub) -> int:
        """Store::is_subsumed (propagation/store.rs:232-238) of ONE store: Kleene-and over all propagators, no propagation."""
        lb = np.ascontiguousarray(lb, np.int32); ub = np.ascontiguousarray(ub, np.int32)
        out = C.c_uint8()
        _check(lib().orc_is_subsumed(self._h, _ptr(lb), _ptr(ub), C.byref(out)))
        return int(out.value)

    def consistency(self, lb: np.ndarray, ub: np.ndarray, active: Optional[np.ndarray] = None, check_dup: bool = False):
        """≡ Consistency::consistency per node.  lb/ub: [n_nodes, n_vars] int32 (copied).  Returns
        (lb, ub, active, status, stats)."""
        lb = np.array(lb, dtype=np.int32, order="C")
        ub = np.array(ub, dtype=np.int32, order="C")
        n = lb.shape[0] if lb.ndim == 2 else 1
        lb = lb.reshape(n, self.n_vars)
        ub = ub.reshape(n, self.n_vars)
        if active is None:
            active = full_active(n, self.n_units)
        else:
            active = np.array(active, dtype=np.uint64, order="C").reshape(n, self.words)
        status = np.zeros(n, dtype=np.uint8)
        st = OrcStats()
        _check(lib().orc_consistency(self._h, n, _ptr(lb), _ptr(ub), _ptr(active), _ptr(status), C.byref(st), int(check_dup)))
        return lb, ub, active, status, st.as_dict()

    def consistency_set(self, bits: np.ndarray, base: int, active: Optional[np.ndarray] = None, check_dup: bool = False):
        """≡ Consistency::consistency per node over IntervalSet<i32> domains (VStoreSet).  bits: [n_nodes, n_vars, set_words]
        uint64 (copied), value v = bit (v - base).  Returns (lb, ub, bits, active, status, stats)."""
        bits = np.array(bits, dtype=np.uint64, order="C")
        if bits.ndim == 2:
            bits = bits[None]
        n, V, sw = bits.shape
        assert V == self.n_vars
        lb = np.zeros((n, V), np.int32)
        ub = np.zeros((n, V), np.int32)
        if active is None:
            active = full_active(n, self.n_units)
        else:
            active = np.array(active, dtype=np.uint64, order="C").reshape(n, self.words)
        status = np.zeros(n, dtype=np.uint8)
        st = OrcStats()
        _check(lib().orc_consistency_set(self._h, n, _ptr(lb), _ptr(ub), _ptr(bits), sw, int(base), _ptr(active), _ptr(status), C.byref(st), int(check_dup)))
        return lb, ub, bits, active, status, st.as_dict()

    def search_set(self, lb0, ub0, set_words: int, base: int, all_solutions=False, node_limit=0, check_dup=False, max_records=0, root_bits=None):
        """DFS over FDSpace (IntervalSet domains, the reference's default); returns (search_stats, prop_stats, records, first_solution).
        root_bits ([n_vars, set_words] uint64): the root's domains as sets instead of the intervals lb0..ub0 (a subtree below an open node
        of a breadth-first expansion, whose domains have holes)."""
        if root_bits is not None:
            root_bits = np.ascontiguousarray(root_bits, dtype=np.uint64).reshape(self.n_vars, int(set_words))
            lb0 = np.zeros(self.n_vars, np.int32) if lb0 is None else lb0
            ub0 = np.zeros(self.n_vars, np.int32) if ub0 is None else ub0
        lb0 = np.ascontiguousarray(lb0, dtype=np.int32)
        ub0 = np.ascontiguousarray(ub0, dtype=np.int32)
        V, W, R, sw = self.n_vars, max(self.words, 1), int(max_records), int(set_words)
        rec = {
            "bits_in": np.zeros((R, V, sw), np.uint64), "bits_out": np.zeros((R, V, sw), np.uint64),
            "lb_out": np.zeros((R, V), np.int32), "ub_out": np.zeros((R, V), np.int32),
            "active_in": np.zeros((R, W), np.uint64), "active_out": np.zeros((R, W), np.uint64),
            "status": np.zeros(R, np.uint8),
        }
        nrec = C.c_uint32(0)
        ss, ps = OrcSearchStats(), OrcStats()
        first = np.zeros(V, np.int32)
        _check(lib().orc_search_set_root(self._h, _ptr(lb0), _ptr(ub0), _ptr(root_bits), sw, int(base), int(all_solutions), int(node_limit), int(check_dup),
                                    C.byref(ss), C.byref(ps), R, _ptr(rec["bits_in"]), _ptr(rec["bits_out"]), _ptr(rec["lb_out"]),
                                    _ptr(rec["ub_out"]), _ptr(rec["active_in"]), _ptr(rec["active_out"]), _ptr(rec["status"]),
                                    C.byref(nrec), _ptr(first)))
        k = nrec.value
        rec = {key: val[:k] for key, val in rec.items()}
        if self.words == 0:
            rec["active_in"] = rec["active_in"][:, :0]
            rec["active_out"] = rec["active_out"][:, :0]
        return ss.as_dict(), ps.as_dict(), rec, first

    def search(self, lb0, ub0, all_solutions=False, node_limit=0, check_dup=False, max_records=0):
        """DFS with the reference's default engine; returns (search_stats, prop_stats, records, first_solution)."""
        lb0 = np.ascontiguousarray(lb0, dtype=np.int32)
        ub0 = np.ascontiguousarray(ub0, dtype=np.int32)
        V, W, R = self.n_vars, max(self.words, 1), int(max_records)
        rec = {
            "lb_in": np.zeros((R, V), np.int32), "ub_in": np.zeros((R, V), np.int32),
            "lb_out": np.zeros((R, V), np.int32), "ub_out": np.zeros((R, V), np.int32),
            "active_in": np.zeros((R, W), np.uint64), "active_out": np.zeros((R, W), np.uint64),
            "status": np.zeros(R, np.uint8),
        }
        nrec = C.c_uint32(0)
        ss, ps = OrcSearchStats(), OrcStats()
        first = np.zeros(V, np.int32)
        _check(lib().orc_search(self._h, _ptr(lb0), _ptr(ub0), int(all_solutions), int(node_limit), int(check_dup),
                                C.byref(ss), C.byref(ps), R, _ptr(rec["lb_in"]), _ptr(rec["ub_in"]), _ptr(rec["lb_out"]),
                                _ptr(rec["ub_out"]), _ptr(rec["active_in"]), _ptr(rec["active_out"]), _ptr(rec["status"]),
                                C.byref(nrec), _ptr(first)))
        k = nrec.value
        rec = {key: val[:k] for key, val in rec.items()}
        if self.words == 0:
            rec["active_in"] = rec["active_in"][:, :0]
            rec["active_out"] = rec["active_out"][:, :0]
        return ss.as_dict(), ps.as_dict(), rec, first


def full_active(n_nodes: int, n_units: int) -> np.ndarray:
    words = (n_units + 63) // 64
    a = np.full((n_nodes, words), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    if words and n_units % 64:
        a[:, -1] = np.uint64((1 << (n_units % 64)) - 1)
    return a


def kat(n_vars: int, lb, ub, props: np.ndarray):
    """The reference's propagator fixture (propagators/mod.rs:108-129) on ONE unit.
    Returns dict(before, ok, after, delta=[(var, ev)], lb, ub)."""
    lb = np.array(lb, dtype=np.int32)
    ub = np.array(ub, dtype=np.int32)
    props = np.ascontiguousarray(props, dtype=PROP_DTYPE)
    before, ok, after = C.c_uint8(), C.c_uint8(), C.c_uint8()
    dn = C.c_uint32()
    dvar = np.zeros(max(n_vars, 1), np.uint32)
    dev = np.zeros(max(n_vars, 1), np.uint8)
    _check(lib().orc_kat(n_vars, _ptr(lb), _ptr(ub), len(props), _ptr(props), C.byref(before), C.byref(ok), C.byref(after),
                         C.byref(dn), _ptr(dvar), _ptr(dev)))
    return {"before": before.value, "ok": bool(ok.value), "after": after.value,
            "delta": [(int(dvar[i]), int(dev[i])) for i in range(dn.value)], "lb": lb, "ub": ub}


def vstore_update(dom, new) -> Tuple[bool, int]:
    ok, ev = C.c_uint8(), C.c_int32()
    _check(lib().orc_vstore_update(dom[0], dom[1], new[0], new[1], C.byref(ok), C.byref(ev)))
    return bool(ok.value), ev.value


def interval_op(op: str, dom, a: int, b: int = 0) -> Tuple[int, int]:
    code = {"shrink_left": 0, "shrink_right": 1, "intersection": 2, "difference": 3, "strict_shrink_left": 4, "strict_shrink_right": 5}[op]
    rl, ru = C.c_int32(), C.c_int32()
    _check(lib().orc_interval_op(code, dom[0], dom[1], a, b, C.byref(rl), C.byref(ru)))
    return rl.value, ru.value


def set_op(op: str, x: np.ndarray, base: int, a: int = 0, y: Optional[np.ndarray] = None):
    """IntervalSet algebra on bitset words (value v = bit v - base): returns (result words, flag)."""
    code = {"difference": 0, "shrink_left": 1, "shrink_right": 2, "intersection": 3, "shift": 4, "is_disjoint": 5, "is_subset": 6}[op]
    x = np.ascontiguousarray(x, np.uint64)
    y = np.ascontiguousarray(x if y is None else y, np.uint64)
    out = np.zeros_like(x)
    flag = C.c_int32()
    _check(lib().orc_set_op(code, _ptr(x), _ptr(y), len(x), int(base), int(a), _ptr(out), C.byref(flag)))
    return out, bool(flag.value)


class Reactor:
    def __init__(self, num_vars: int):
        self._h = lib().orc_reactor_new(num_vars)

    def __del__(self):
        lib().orc_reactor_free(self._h)

    def subscribe(self, var, ev, prop):
        _check(lib().orc_reactor_subscribe(self._h, var, ev, prop))

    def unsubscribe(self, var, ev, prop):
        _check(lib().orc_reactor_unsubscribe(self._h, var, ev, prop))

    def react(self, var, ev):
        out = np.zeros(64, np.uint32)
        n = C.c_uint32()
        _check(lib().orc_reactor_react(self._h, var, ev, _ptr(out), 64, C.byref(n)))
        return [int(x) for x in out[: n.value]]

    def is_empty(self):
        return bool(lib().orc_reactor_is_empty(self._h))


class Fifo:
    def __init__(self, cap: int):
        self._h = lib().orc_fifo_new(cap)

    def __del__(self):
        lib().orc_fifo_free(self._h)

    def schedule(self, i):
        _check(lib().orc_fifo_schedule(self._h, i))

    def unschedule(self, i):
        _check(lib().orc_fifo_unschedule(self._h, i))

    def pop(self):
        r = lib().orc_fifo_pop(self._h)
        return None if r < 0 else int(r)

    def is_empty(self):
        return bool(lib().orc_fifo_is_empty(self._h))


def middle_val(lb, ub) -> int:
    return int(lib().orc_middle_val(lb, ub))


def first_smallest_var(doms) -> int:
    lb = np.array([d[0] for d in doms], np.int32)
    ub = np.array([d[1] for d in doms], np.int32)
    r = lib().orc_first_smallest_var(len(doms), _ptr(lb), _ptr(ub))
    if r < 0:
        raise OraclePanic(lib().orc_last_error().decode())
    return int(r)
