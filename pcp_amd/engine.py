"""ctypes binding of libpcp_hip.so — the C ABI declared in include/pcp_hip.h.

This is the product path: every call goes through the C-ABI into the hand-written gfx950 kernels.  There is
no CPU fallback: importing works anywhere (so that the non-GPU tests can check the exported symbols), but
``Context()`` raises ``EngineUnavailable`` when the library or a HIP device is missing.
PyTorch is used only as plumbing (device buffers, streams); the ABI itself takes raw pointers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .model import PROP_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCP_HIP_LIB", os.path.join(_HERE, "libpcp_hip.so"))  # PCP_HIP_LIB: profiling builds (tools/ablate.sh)

PCP_OK = 0
ERRORS = {-1: "PCP_ERR_ARG", -2: "PCP_ERR_CONTRACT", -3: "PCP_ERR_HIP", -4: "PCP_ERR_NOMEM", -5: "PCP_ERR_UNSUPPORTED", -6: "PCP_ERR_NODEVICE"}

# Every symbol include/pcp_hip.h declares (tests/test_abi.py checks the .so exports each of them).
ABI_SYMBOLS = [
    "pcp_ctx_create", "pcp_ctx_destroy", "pcp_last_error", "pcp_strerror", "pcp_abi_version",
    "pcp_model_reset", "pcp_model_push_props", "pcp_model_push_formula", "pcp_model_push_sum", "pcp_model_truncate", "pcp_model_n_units", "pcp_model_set_hull",
    "pcp_propagate", "pcp_propagate_device", "pcp_propagate_device_units", "pcp_branch_device", "pcp_branch_device_hint", "pcp_pack_rows", "pcp_unpack_rows", "pcp_branch_device_cells", "pcp_branch_device_set", "pcp_dfs_device", "pcp_dfs_forest_device", "pcp_dfs_forest_device_set", "pcp_dfs_forest_split_set", "pcp_stats_reset", "pcp_stats_read", "pcp_debug_counters", "pcp_last_kernel_ms", "pcp_last_plan", "pcp_set_option",
]


class PcpStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("steps", "steps3", "narrowings", "waves", "failed_nodes", "nodes", "evaluated", "full_evals")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PcpPlan(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("nodes_per_block", "team", "packed", "word_level", "global_dom", "compact", "implicit_active", "set_mode",
                                          "grid", "block", "lds_bytes", "list_cap", "path")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class DeviceBatch(C.Structure):
    _fields_ = [("lb_in", C.c_void_p), ("ub_in", C.c_void_p), ("lb_out", C.c_void_p), ("ub_out", C.c_void_p),
                ("active_in", C.c_void_p), ("active_out", C.c_void_p), ("status", C.c_void_p), ("bits_in", C.c_void_p), ("bits_out", C.c_void_p),
                ("dirty_var", C.c_void_p), ("cell_format", C.c_uint32), ("reserved", C.c_uint32)]


class DfsState(C.Structure):
    _fields_ = [("lb", C.c_void_p), ("ub", C.c_void_p), ("capacity", C.c_uint32), ("sp", C.c_void_p), ("stop", C.c_void_p), ("status", C.c_void_p),
                ("counters", C.c_void_p), ("first_solution", C.c_void_p), ("dirty", C.c_void_p)]


class ForestState(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("level_capacity", C.c_uint32), ("trail_capacity", C.c_uint32), ("reserved", C.c_uint32),
                ("bits", C.c_void_p), ("tree", C.c_void_p), ("levels", C.c_void_p), ("trail", C.c_void_p), ("counters", C.c_void_p),
                ("total_nodes", C.c_void_p), ("stop", C.c_void_p), ("first_solution", C.c_void_p), ("solution_flag", C.c_void_p)]


DFS_FULL = 0xFFFFFFFF


class EngineUnavailable(RuntimeError):
    """libpcp_hip.so is missing or no HIP device is present.  There is deliberately no fallback."""


class PcpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


_lib = None


def load_library():
    """dlopen libpcp_hip.so and declare the prototypes.  Raises EngineUnavailable if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.  If PyTorch is going to be used in
    # this process (device buffers, torch.distributed) its runtime has to be the one that gets loaded, otherwise a
    # later `import torch` finds "No HIP GPUs".  Importing it first makes libpcp_hip.so bind to the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # missing HIP runtime etc.
        raise EngineUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    L.pcp_abi_version.restype = u32
    L.pcp_strerror.restype = C.c_char_p
    L.pcp_strerror.argtypes = [i32]
    L.pcp_last_error.restype = C.c_char_p
    L.pcp_last_error.argtypes = [vp]
    L.pcp_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.pcp_ctx_destroy.argtypes = [vp]
    L.pcp_ctx_destroy.restype = None
    L.pcp_model_reset.argtypes = [vp, u32, u32]
    L.pcp_model_push_props.argtypes = [vp, u32, vp]
    L.pcp_model_push_sum.argtypes = [vp, u32, vp, C.POINTER(u32)]
    L.pcp_model_push_formula.argtypes = [vp, u32, vp, u32, vp]
    L.pcp_model_push_formula.restype = i32
    L.pcp_model_truncate.argtypes = [vp, u32]
    L.pcp_model_n_units.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.pcp_model_set_hull.argtypes = [vp, i32, i32]
    L.pcp_propagate.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
    L.pcp_propagate_device.argtypes = [vp, u32, C.POINTER(DeviceBatch), vp]
    L.pcp_branch_device.argtypes = [vp, u32] + [vp] * 9
    L.pcp_branch_device_hint.argtypes = [vp, u32] + [vp] * 10
    L.pcp_pack_rows.argtypes = [vp, u32, vp, vp, vp, vp]
    L.pcp_unpack_rows.argtypes = [vp, u32, vp, vp, vp, vp]
    L.pcp_branch_device_cells.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
    L.pcp_branch_device_set.argtypes = [vp, u32] + [vp] * 9
    L.pcp_dfs_device.argtypes = [vp, C.POINTER(DfsState), u32, u32, C.c_uint64, vp]
    L.pcp_dfs_forest_device_set.argtypes = [vp, C.POINTER(ForestState), u32, u32, C.c_uint64, vp]
    L.pcp_dfs_forest_split_set.argtypes = [vp, C.POINTER(ForestState), u32, vp, vp, vp]
    L.pcp_dfs_forest_device.argtypes = [vp, C.POINTER(DfsState), u32, u32, u32, C.c_uint64, vp]
    L.pcp_stats_reset.argtypes = [vp, vp]
    L.pcp_stats_read.argtypes = [vp, C.POINTER(PcpStats), vp]
    L.pcp_debug_counters.argtypes = [vp, C.POINTER(C.c_uint64), u32, vp]
    L.pcp_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.pcp_last_plan.argtypes = [vp, C.POINTER(PcpPlan)]
    L.pcp_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    for f in ("pcp_ctx_create", "pcp_model_reset", "pcp_model_push_props", "pcp_model_push_formula", "pcp_model_push_sum", "pcp_model_truncate", "pcp_model_n_units", "pcp_model_set_hull", "pcp_model_set_hull",
              "pcp_propagate", "pcp_propagate_device", "pcp_propagate_device_units", "pcp_branch_device", "pcp_branch_device_hint", "pcp_pack_rows", "pcp_unpack_rows", "pcp_branch_device_cells", "pcp_branch_device_set", "pcp_dfs_device", "pcp_dfs_forest_device", "pcp_dfs_forest_device_set", "pcp_dfs_forest_split_set", "pcp_stats_reset", "pcp_stats_read", "pcp_debug_counters", "pcp_last_kernel_ms", "pcp_last_plan", "pcp_set_option"):
        getattr(L, f).restype = i32
    _lib = L
    return L


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One pcp_ctx == one constraint store (CStore) on one GPU."""

    def __init__(self, device: int = 0):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.pcp_ctx_create(device, C.byref(h))
        if rc == -6:
            raise EngineUnavailable("pcp_ctx_create: no HIP device (PCP_ERR_NODEVICE); the engine has no CPU path")
        if rc != 0:
            raise PcpError(rc, self._L.pcp_strerror(rc).decode())
        self._h = h
        self.device = device
        self.n_vars = 0
        self.n_units = 0
        self.set_words = 0
        self.supports_hints = True  # pcp_device_batch.dirty_var / pcp_branch_device_hint (ABI v7): search drivers keep a hint per open node
        for kv in filter(None, os.environ.get("PCP_SET_OPTIONS", "").split(",")):  # experiments: pcp_set_option on every new context (k=v,k=v)
            k_, v_ = kv.split("=")
            self.set_option(k_, int(v_))

    def close(self):
        if getattr(self, "_h", None):
            self._L.pcp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise PcpError(rc, self._L.pcp_last_error(self._h).decode())

    # ---- model ------------------------------------------------------------------------------------------
    def set_model(self, n_vars: int, props: np.ndarray, set_words: int = 0, sums=None):
        """pcp_model_reset + pcp_model_push_props.  set_words > 0: IntervalSet<i32> domains (VStoreSet, the reference's default
        FDSpace) carried as bitsets — declare the hull with set_hull(lo, hi) before propagating (value v = bit v - lo)."""
        props = np.ascontiguousarray(props, dtype=PROP_DTYPE)
        self._check(self._L.pcp_model_reset(self._h, n_vars, int(set_words)))
        self.n_vars = int(n_vars)
        self.set_words = int(set_words)
        for members in (sums or []):  # term::Sum views, numbered in order (pcp_model_push_sum)
            mv = np.ascontiguousarray(members, np.uint32)
            t = C.c_uint32()
            self._check(self._L.pcp_model_push_sum(self._h, len(mv), _np_ptr(mv), C.byref(t)))
        self.push_props(props)

    def push_props(self, props: np.ndarray):
        props = np.ascontiguousarray(props, dtype=PROP_DTYPE)
        if len(props):
            self._check(self._L.pcp_model_push_props(self._h, len(props), _np_ptr(props)))
        self._refresh()

    # ---- the push interface pcp_amd.model.push_model drives (stores with formula propagators) ------------------------------
    def reset_model(self, n_vars: int, set_words: int = 0):
        self._check(self._L.pcp_model_reset(self._h, n_vars, int(set_words)))
        self.n_vars, self.set_words = int(n_vars), int(set_words)
        self._refresh()

    def push_sum(self, members) -> int:
        mv = np.ascontiguousarray(members, np.uint32)
        t = C.c_uint32()
        self._check(self._L.pcp_model_push_sum(self._h, len(mv), _np_ptr(mv), C.byref(t)))
        return t.value

    def push_formula(self, nodes: np.ndarray, leaves: np.ndarray):
        """pcp_model_push_formula: ONE unit = a tree of Conjunction / Disjunction nodes over elementary leaves."""
        from .model import FNODE_DTYPE
        nodes = np.ascontiguousarray(nodes, dtype=FNODE_DTYPE)
        leaves = np.ascontiguousarray(leaves, dtype=PROP_DTYPE)
        self._check(self._L.pcp_model_push_formula(self._h, len(nodes), _np_ptr(nodes), len(leaves), _np_ptr(leaves)))
        self._refresh()

    def set_hull(self, lo: int, hi: int):
        """pcp_model_set_hull: hull of the variables' initial domains (VStore.alloc); forgotten by set_model."""
        self._check(self._L.pcp_model_set_hull(self._h, int(lo), int(hi)))

    def truncate(self, n_units: int):
        self._check(self._L.pcp_model_truncate(self._h, n_units))
        self._refresh()

    def _refresh(self):
        nu, npr = C.c_uint32(), C.c_uint32()
        self._check(self._L.pcp_model_n_units(self._h, C.byref(nu), C.byref(npr)))
        self.n_units, self.n_props = nu.value, npr.value
        self.words = (self.n_units + 63) // 64

    def set_option(self, key: str, value: int):
        self._check(self._L.pcp_set_option(self._h, key.encode(), int(value)))

    # ---- propagation, host buffers (pcp_propagate) ----------------------------------------------------------
    def propagate_set(self, bits, active: Optional[np.ndarray] = None, want_stats: bool = True):
        """Set mode: ≡ Consistency::consistency on each node's IntervalSet domains.  bits: [n, n_vars, set_words] uint64.
        Returns (lb, ub, bits, active, status, stats) as fresh arrays (lb/ub = the bounds of the fixpoint sets)."""
        bits = np.array(bits, dtype=np.uint64, order="C")
        if bits.ndim == 2:
            bits = bits[None]
        n = bits.shape[0]
        assert bits.shape[1:] == (self.n_vars, self.set_words), bits.shape
        lb = np.zeros((n, self.n_vars), np.int32)
        ub = np.zeros((n, self.n_vars), np.int32)
        if active is not None:
            active = np.array(active, dtype=np.uint64, order="C").reshape(n, self.words)
        status = np.zeros(n, dtype=np.uint8)
        st = PcpStats()
        self._check(self._L.pcp_propagate(self._h, n, _np_ptr(lb), _np_ptr(ub), _np_ptr(bits), _np_ptr(active), _np_ptr(status),
                                          C.byref(st) if want_stats else None))
        return lb, ub, bits, active, status, st.as_dict()

    def propagate(self, lb, ub, active: Optional[np.ndarray] = None, want_stats: bool = True):
        """≡ Consistency::consistency on each row.  Returns (lb, ub, active, status, stats) as fresh arrays."""
        lb = np.array(lb, dtype=np.int32, order="C")
        ub = np.array(ub, dtype=np.int32, order="C")
        n = lb.shape[0] if lb.ndim == 2 else 1
        lb = lb.reshape(n, self.n_vars)
        ub = ub.reshape(n, self.n_vars)
        if active is not None:
            active = np.array(active, dtype=np.uint64, order="C").reshape(n, self.words)
        status = np.zeros(n, dtype=np.uint8)
        st = PcpStats()
        self._check(self._L.pcp_propagate(self._h, n, _np_ptr(lb), _np_ptr(ub), None, _np_ptr(active), _np_ptr(status),
                                          C.byref(st) if want_stats else None))
        return lb, ub, active, status, st.as_dict()

    def propagate_implicit(self, lb, ub, want_active: bool = True, in_place: bool = True):
        """Implicit-active nodes through the device-resident entry: active_in = NULL (every unit active, liveness derived
        from the domains), `active` materialised on request only.  Returns (lb, ub, active or None, status, stats)."""
        import torch
        lb = np.array(lb, dtype=np.int32, order="C")
        n = lb.shape[0] if lb.ndim == 2 else 1
        lb = lb.reshape(n, self.n_vars)
        ub = np.array(ub, dtype=np.int32, order="C").reshape(n, self.n_vars)
        dev = torch.device("cuda", self.device)
        t_lb, t_ub = torch.from_numpy(lb).to(dev), torch.from_numpy(ub).to(dev)
        o_lb, o_ub = (t_lb, t_ub) if in_place else (torch.empty_like(t_lb), torch.empty_like(t_ub))
        t_act = torch.zeros((n, max(self.words, 1)), dtype=torch.int64, device=dev) if want_active else None
        t_st = torch.zeros(n, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        before = self.stats_read(stream)
        self.propagate_device(n, t_lb, t_ub, o_lb, o_ub, None, t_act, t_st, stream)
        after = self.stats_read(stream)
        act = t_act.cpu().numpy().view(np.uint64)[:, : self.words] if want_active else None
        return o_lb.cpu().numpy(), o_ub.cpu().numpy(), act, t_st.cpu().numpy(), {k: after[k] - before[k] for k in after}

    # ---- propagation, device-resident (pcp_propagate_device) ------------------------------------------------
    def propagate_device(self, n_nodes: int, lb_in, ub_in, lb_out, ub_out, active_in, active_out, status, stream_ptr: int = 0,
                         bits_in=None, bits_out=None, dirty=None, cells=False):
        """All arguments are torch tensors on this context's device (or None for the optional masks); nothing is
        synchronised.  Tensors: lb/ub int32 [n,V]; active int64/uint64 [n,words]; status uint8 [n]; set mode: bits int64
        [n,V,set_words] (lb_in/ub_in ignored); dirty int32 [n]: per node the one variable in which it differs from a fixpoint of this
        model, -1 / >= n_vars = none (pcp_device_batch.dirty_var).  cells=True (pcp_device_batch.cell_format PCP_CELLS_PACKED16): lb_in / lb_out
        are int32 [n,V] tensors of packed CELLS (pack_rows), ub_* ignored."""
        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        bt = DeviceBatch(p(lb_in), p(ub_in), p(lb_out), p(ub_out), p(active_in), p(active_out), p(status), p(bits_in), p(bits_out), p(dirty),
                         1 if cells else 0, 0)
        self._check(self._L.pcp_propagate_device(self._h, n_nodes, C.byref(bt), C.c_void_p(stream_ptr)))

    def propagate_device_units(self, n_nodes: int, lb_in, ub_in, lb_out, ub_out, active_in, active_out, status, unit_off, units, stream_ptr: int = 0):
        """pcp_propagate_device_units (ABI v8): the batch of `propagate_device` where node i also carries its own unary propagators
        units[unit_off[i] : unit_off[i + 1]] — `unit_off` an int32 device tensor [n + 1], `units` a uint8 device tensor holding the pcp_prop
        records (model.PROP_DTYPE bytes).  Enumerate's x != v children (search/branching/enumerate.rs:48-59) in one launch."""
        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        bt = DeviceBatch(p(lb_in), p(ub_in), p(lb_out), p(ub_out), p(active_in), p(active_out), p(status), None, None, None, 0, 0)
        self._check(self._L.pcp_propagate_device_units(self._h, n_nodes, C.byref(bt), p(unit_off), p(units), C.c_void_p(stream_ptr)))

    def pack_rows(self, lb, ub, cells=None, stream_ptr: int = 0):
        """pcp_pack_rows: int32 bounds rows [n,V] -> rows of packed cells (an int32 [n,V] tensor whose bits are the cells)."""
        import torch
        if cells is None:
            cells = torch.empty_like(lb)
        self._check(self._L.pcp_pack_rows(self._h, lb.shape[0], C.c_void_p(lb.data_ptr()), C.c_void_p(ub.data_ptr()), C.c_void_p(cells.data_ptr()), C.c_void_p(stream_ptr)))
        return cells

    def unpack_rows(self, cells, lb=None, ub=None, stream_ptr: int = 0):
        """pcp_unpack_rows: rows of packed cells -> (lb, ub) int32 rows."""
        import torch
        lb = torch.empty_like(cells) if lb is None else lb
        ub = torch.empty_like(cells) if ub is None else ub
        self._check(self._L.pcp_unpack_rows(self._h, cells.shape[0], C.c_void_p(cells.data_ptr()), C.c_void_p(lb.data_ptr()), C.c_void_p(ub.data_ptr()), C.c_void_p(stream_ptr)))
        return lb, ub

    def branch_device_cells(self, n_nodes: int, cells, status, child_cells, counts, stream_ptr: int = 0, child_dirty=None):
        """pcp_branch_device_cells: branch_device over rows of packed cells (implicit nodes)."""
        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        self._check(self._L.pcp_branch_device_cells(self._h, n_nodes, p(cells), p(status), p(child_cells), p(child_dirty), p(counts), C.c_void_p(stream_ptr)))

    def branch_device(self, n_nodes: int, lb, ub, active, status, child_lb, child_ub, child_active, counts, stream_ptr: int = 0, child_dirty=None):
        """pcp_branch_device(_hint) on torch tensors of this context's device (counts: int32[5]; child_dirty: int32 [2 n] capacity, receives
        the variable each child was branched on); nothing is synchronised."""
        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        self._check(self._L.pcp_branch_device_hint(self._h, n_nodes, p(lb), p(ub), p(active), p(status), p(child_lb), p(child_ub), p(child_active),
                                                   p(child_dirty), p(counts), C.c_void_p(stream_ptr)))

    def branch_device_set(self, n_nodes: int, bits, lb, ub, active, status, child_bits, child_active, counts, stream_ptr: int = 0):
        """pcp_branch_device_set (set mode) on torch tensors of this context's device (counts: int32[5])."""
        def p(t):
            return None if t is None else C.c_void_p(t.data_ptr())
        self._check(self._L.pcp_branch_device_set(self._h, n_nodes, p(bits), p(lb), p(ub), p(active), p(status), p(child_bits), p(child_active),
                                                  p(counts), C.c_void_p(stream_ptr)))

    def dfs_device(self, lb0, ub0, n_steps: int, capacity: int = 4096, stop_on_solution: bool = True, node_limit: int = 0, chunk: int = 256):
        """pcp_dfs_device: the reference's one-node-per-step DFS entirely on the device.  Runs until the stack is empty, a stop
        condition holds or n_steps steps were enqueued (in chunks of `chunk` steps between two reads of the 8-byte state).
        Returns dict(nodes, solutions, failed, error, open, steps_enqueued, first_solution)."""
        import torch
        dev = torch.device("cuda", self.device)
        V = self.n_vars
        lb = torch.zeros((capacity, V), dtype=torch.int32, device=dev)
        ub = torch.zeros((capacity, V), dtype=torch.int32, device=dev)
        lb[0] = torch.from_numpy(np.ascontiguousarray(lb0, np.int32)).to(dev)
        ub[0] = torch.from_numpy(np.ascontiguousarray(ub0, np.int32)).to(dev)
        state = torch.tensor([1, 0], dtype=torch.int32, device=dev)  # sp, stop
        status = torch.zeros(capacity, dtype=torch.uint8, device=dev)
        counters = torch.zeros(5, dtype=torch.int64, device=dev)
        sol = torch.zeros(V, dtype=torch.int32, device=dev)
        dirty = torch.full((capacity,), -1, dtype=torch.int32, device=dev)  # per stack row: the variable it was branched on (the root: none)
        st = DfsState(lb.data_ptr(), ub.data_ptr(), capacity, state.data_ptr(), state.data_ptr() + 4, status.data_ptr(), counters.data_ptr(), sol.data_ptr(), dirty.data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        done = 0
        while done < n_steps:
            k = min(chunk, n_steps - done)
            self._check(self._L.pcp_dfs_device(self._h, C.byref(st), k, int(bool(stop_on_solution)), int(node_limit), C.c_void_p(stream)))
            done += k
            sp, stop = state.cpu().tolist()
            if sp == 0 or stop:
                break
        cn = counters.cpu().tolist()
        sp, stop = state.cpu().tolist()
        return {"nodes": cn[0], "solutions": cn[1], "failed": cn[2], "error": cn[3], "open": sp, "stopped": bool(stop), "steps_enqueued": done,
                "first_solution": sol.cpu().numpy() if cn[1] else None}

    def dfs_forest(self, root_lb, root_ub, stop_on_solution: bool = False, node_limit_per_tree: int = 0, steps_per_launch: int = 256, capacity: int = 2048,
                   max_launches: int = 1 << 30, node_budget: int = 0, want_solution: bool = False, rebalance: bool = True, info: dict | None = None, dist=None, max_capacity: int = 0,
                   sp0=None, hints: bool = True):
        """pcp_dfs_forest_device: the reference's search loop (interval mode, all-XNeqY models) on many subtrees at once, one workgroup per
        tree, each exactly a pcp_dfs_device instance.  root_lb / root_ub: [n_trees, n_vars] int32 (numpy or CUDA tensors): the roots, not yet
        propagated.  Launches of steps_per_launch nodes per tree are repeated until every stack is empty, a tree stopped (solution with
        stop_on_solution / error / its node limit), node_budget nodes were explored by all trees together (checked between launches) or
        max_launches is reached.  With ``dist`` (a torch.distributed group / module, one rank per GPU) the launches run in lockstep on every
        rank, node_budget and the end of the search are GLOBAL (one small all_reduce per launch), and a rank whose trees ran dry takes open
        nodes from the other ranks (search_forest.refill_across_ranks).  ``capacity`` rows per tree are allocated up front and doubled, up to
        ``max_capacity``, whenever a tree fills its stack (error 1 is only reported beyond that).  Returns dict(nodes, solutions, failed, error, open, launches,
        per_tree=[n_trees, 5], first_solutions) — this rank's counters."""
        import torch
        dev = torch.device("cuda", self.device)
        V = self.n_vars
        to_dev = lambda x: (torch.from_numpy(np.ascontiguousarray(x, np.int32)) if isinstance(x, np.ndarray) else x).to(dev).reshape(-1, V)
        rl, ru = to_dev(root_lb), to_dev(root_ub)
        T = rl.shape[0]
        lb = torch.empty((T, capacity, V), dtype=torch.int32, device=dev)
        ub = torch.empty((T, capacity, V), dtype=torch.int32, device=dev)
        lb[:, 0] = rl; ub[:, 0] = ru
        sp = torch.ones(T, dtype=torch.int32, device=dev)
        if sp0 is not None:  # (trees that start with an empty stack: a rank that only takes part in the collectives until a refill reaches it)
            sp = torch.tensor(list(sp0), dtype=torch.int32, device=dev)
        stop = torch.zeros(T, dtype=torch.int32, device=dev)
        status = torch.zeros((T, capacity), dtype=torch.uint8, device=dev)
        counters = torch.zeros((T, 5), dtype=torch.int64, device=dev)
        sol = torch.zeros((T, V), dtype=torch.int32, device=dev) if want_solution else None
        dirty = torch.full((T, capacity), -1, dtype=torch.int32, device=dev) if hints else None  # pcp_dfs_state.dirty: one word per stack row
        from .search_forest import ForestStacks, run_forest_loop
        stream = torch.cuda.current_stream(dev).cuda_stream
        box = {}

        def point(fs):  # (re)build the pcp_dfs_state over the forest's current buffers
            box["st"] = DfsState(fs.lb.data_ptr(), fs.ub.data_ptr(), fs.capacity, sp.data_ptr(), stop.data_ptr(), fs.status.data_ptr(), counters.data_ptr(),
                                 sol.data_ptr() if want_solution else None, fs.dirty.data_ptr() if fs.dirty is not None else None)

        fs = ForestStacks(lb, ub, status, sp, stop, counters, max_capacity=max(max_capacity, capacity), on_grow=point, dirty=dirty)
        del lb, ub, status, dirty
        point(fs)

        def launch():
            self._check(self._L.pcp_dfs_forest_device(self._h, C.byref(box["st"]), T, int(steps_per_launch), int(bool(stop_on_solution)), int(node_limit_per_tree), C.c_void_p(stream)))

        loop = run_forest_loop(launch, fs, stop_on_solution=stop_on_solution, node_budget=node_budget,
                               rebalance=rebalance and not node_limit_per_tree, max_launches=max_launches, dist=dist)
        launches, steals = loop["launches"], loop["steals"]
        if info is not None:
            info.update(loop)
        cn = counters.cpu().numpy()
        return {"nodes": int(cn[:, 0].sum()), "solutions": int(cn[:, 1].sum()), "failed": int(cn[:, 2].sum()), "error": int(cn[:, 3].max()),
                "open": int(sp.sum().item()), "launches": launches, "per_tree": cn, "trees": T, "steals": steals,
                "first_solutions": sol.cpu().numpy() if want_solution else None}

    def dfs_forest_set(self, root_bits, stop_on_solution: bool = False, node_limit: int = 0, steps_per_launch: int = 256,
                       level_capacity: int = 0, trail_capacity: int = 0, max_launches: int = 1 << 30, want_solution: bool = True,
                       info: dict | None = None, rebalance: bool = True):
        """pcp_dfs_forest_device_set: the reference's search loop over FDSpace on the device, one tree per workgroup, the current
        node in LDS, an undo trail in HBM.  root_bits: [n_trees, n_vars, set_words] uint64 (numpy or a CUDA int64 tensor): the roots,
        not yet propagated.  Launches of steps_per_launch nodes per tree are repeated until every tree is finished, the forest
        stopped (solution / node limit / error) or max_launches is reached.
        Returns dict(nodes, solutions, failed, error, finished_trees, stopped, launches, first_solution, per_tree=[n_trees, 4])."""
        import torch
        dev = torch.device("cuda", self.device)
        V, sw = self.n_vars, self.set_words
        # 0 = the model's own bound: a trail entry takes at least one value out of a set and is popped before the value can return, so a
        # tree never holds more entries (or open levels) than the root has values
        bound = V * sw * 64 + 16
        trail_capacity = int(trail_capacity) or bound
        level_capacity = int(level_capacity) or min(bound, 1 << 14)
        if isinstance(root_bits, np.ndarray):
            bits = torch.from_numpy(np.ascontiguousarray(root_bits).view(np.int64)).to(dev)
        else:
            bits = root_bits.to(dev).contiguous().clone()
        bits = bits.reshape(-1, V, sw)
        T = bits.shape[0]
        tree = torch.zeros((T, 4), dtype=torch.int32, device=dev)
        tree[:, 2] = -1  # PCP_DFS_FULL
        levels = torch.empty((T, level_capacity, 4), dtype=torch.int32, device=dev)  # (written before they are read: no fill)
        trail = torch.empty((T, trail_capacity, 4), dtype=torch.int32, device=dev)
        counters = torch.zeros((T, 4), dtype=torch.int64, device=dev)
        glob = torch.zeros(4, dtype=torch.int64, device=dev)  # total nodes | stop (low word of [1]) | solution flag (low word of [2])
        sol = torch.zeros(V, dtype=torch.int32, device=dev) if want_solution else None
        st = ForestState(T, level_capacity, trail_capacity, 0, bits.data_ptr(), tree.data_ptr(), levels.data_ptr(), trail.data_ptr(), counters.data_ptr(),
                         glob.data_ptr(), glob.data_ptr() + 8, sol.data_ptr() if want_solution else None, glob.data_ptr() + 16 if want_solution else None)
        stream = torch.cuda.current_stream(dev).cuda_stream
        launches = splits = 0
        while launches < max_launches:
            self._check(self._L.pcp_dfs_forest_device_set(self._h, C.byref(st), int(steps_per_launch), int(bool(stop_on_solution)), int(node_limit), C.c_void_p(stream)))
            launches += 1
            g = glob.cpu().tolist()  # (the launch's only synchronisation)
            ts = tree.cpu().numpy()
            fin = (ts[:, 3] & 1) != 0
            if (g[1] & 0xFFFFFFFF) or (node_limit and g[0] >= node_limit) or fin.all():
                break
            if rebalance and fin.any():
                # finished trees take the oldest open right branch of the trees with the most levels left (pcp_dfs_forest_split_set)
                left = np.where(fin, 0, ts[:, 0].astype(np.int64) - (ts[:, 3] >> 8))
                donors = [int(i) for i in np.argsort(-left) if left[i] > 0]
                recv = [int(i) for i in np.nonzero(fin)[0]]
                k = min(len(donors), len(recv))
                if k:
                    pairs = torch.tensor(np.stack([donors[:k], recv[:k]], axis=1).astype(np.int32).ravel(), dtype=torch.int32, device=dev)
                    done = torch.zeros(k, dtype=torch.int32, device=dev)
                    self._check(self._L.pcp_dfs_forest_split_set(self._h, C.byref(st), k, C.c_void_p(pairs.data_ptr()), C.c_void_p(done.data_ptr()), C.c_void_p(stream)))
                    splits += int(done.sum().item())
        cn = counters.cpu().numpy()
        g = glob.cpu().tolist()
        if info is not None:
            info.update(trail_max=int(tree[:, 1].max().item()), levels_max=int(tree[:, 0].max().item()), trees=T, splits=splits)
        return {"nodes": int(cn[:, 0].sum()), "solutions": int(cn[:, 1].sum()), "failed": int(cn[:, 2].sum()), "error": int(cn[:, 3].max()),
                "finished_trees": int((tree[:, 3] & 1).sum().item()), "splits": splits, "stopped": bool(g[1] & 0xFFFFFFFF), "launches": launches, "total_nodes": int(g[0]),
                "first_solution": sol.cpu().numpy() if (want_solution and (g[2] & 0xFFFFFFFF)) else None, "per_tree": cn}

    def stats_reset(self, stream_ptr: int = 0):
        self._check(self._L.pcp_stats_reset(self._h, C.c_void_p(stream_ptr)))

    def stats_read(self, stream_ptr: int = 0) -> dict:
        st = PcpStats()
        self._check(self._L.pcp_stats_read(self._h, C.byref(st), C.c_void_p(stream_ptr)))
        return st.as_dict()

    DBG = {"big_dense": 0, "big_sparse": 1, "neq_tiles": 2, "neq_overlap": 3, "small_nodes": 4, "neq_lean": 8, "neq_lean_passes": 9, "neq_lean_handover": 10}

    def debug_counters(self, stream_ptr: int = 0) -> dict:
        """Kernel-internal diagnostic counters since the last stats_reset (pcp_debug_counters, ABI v6): which code paths ran."""
        out = (C.c_uint64 * 16)()
        self._check(self._L.pcp_debug_counters(self._h, out, 16, C.c_void_p(stream_ptr)))
        d = {k: int(out[i]) for k, i in self.DBG.items()}
        d["raw"] = [int(x) for x in out]
        return d

    def last_plan(self) -> dict:
        """pcp_last_plan: the launch geometry of the last propagate call."""
        pl = PcpPlan()
        self._check(self._L.pcp_last_plan(self._h, C.byref(pl)))
        return pl.as_dict()

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._check(self._L.pcp_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)


def full_active(n_nodes: int, n_units: int) -> np.ndarray:
    """`active` rows with every unit set (Store::alloc inserts each new propagator, propagation/store.rs:227)."""
    words = (n_units + 63) // 64
    a = np.full((n_nodes, words), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    if words and n_units % 64:
        a[:, -1] = np.uint64((1 << (n_units % 64)) - 1)
    return a
