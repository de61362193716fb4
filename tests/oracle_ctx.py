"""A stand-in for pcp_amd.engine.Context backed by the CPU oracle — TEST INFRASTRUCTURE.  It lets the CPU suite
drive the host-side search / worklist logic (pcp_amd.search, pcp_amd.distributed), which in production only
ever talks to the HIP engine."""
import numpy as np

from oracle import oracle as orc


class OracleCtx:
    def __init__(self, n_vars, props):
        self._m = orc.OracleModel(n_vars, props)
        self.n_vars = n_vars
        self.n_units = self._m.n_units
        self.words = self._m.words

    def propagate(self, lb, ub, active=None, want_stats=True):
        lb, ub, act, status, st = self._m.consistency(lb, ub, active, check_dup=False)
        return lb, ub, act, status, {"steps": st["steps"], "steps3": 0, "narrowings": st["narrowings"], "nodes": st["nodes"]}


class OracleDeviceCtx(OracleCtx):
    """The device-resident entry points of pcp_amd.engine.Context (propagate_device / branch_device / stats) over CPU tensors,
    backed by the oracle and the numpy brancher — lets the CPU suite run pcp_amd.search_device.DeviceSearch and
    pcp_amd.distributed.parallel_search_device end to end over gloo.  TEST INFRASTRUCTURE."""

    def __init__(self, n_vars, props):
        super().__init__(n_vars, props)
        self.device = "cpu"
        self.supports_hints = True  # (the engine's pcp_device_batch.dirty_var: DeviceSearch then keeps a hint row per open node and the drivers move it along)
        self.hints_seen = 0
        self._opts = {}
        self._stats = {"steps": 0, "steps3": 0, "narrowings": 0, "nodes": 0, "evaluated": 0, "full_evals": 0, "waves": 0, "failed_nodes": 0}

    def set_option(self, key, value):
        self._opts[key] = int(value)

    def stats_reset(self, stream=0):
        for k in self._stats:
            self._stats[k] = 0

    def stats_read(self, stream=0):
        return dict(self._stats)

    # ---- rows of packed cells (pcp_device_batch.cell_format PCP_CELLS_PACKED16): cell = (-lb & 0xffff) | ub << 16, as int32 tensors
    @staticmethod
    def _pack(L, U):
        assert (np.abs(L) <= 16383).all() and (np.abs(U) <= 16383).all()
        return (((-L.astype(np.int64)) & 0xffff) | ((U.astype(np.int64) & 0xffff) << 16)).astype(np.uint32).view(np.int32)

    @staticmethod
    def _unpack(cells):
        c = np.ascontiguousarray(cells).view(np.uint32)
        return (-(c & 0xffff).astype(np.uint16).view(np.int16).astype(np.int32)), (c >> 16).astype(np.uint16).view(np.int16).astype(np.int32)

    def pack_rows(self, lb, ub, cells=None, stream_ptr=0):
        import torch
        out = torch.from_numpy(self._pack(lb.numpy(), ub.numpy()))
        if cells is None:
            return out
        cells[:] = out
        return cells

    def unpack_rows(self, cells, lb=None, ub=None, stream_ptr=0):
        import torch
        L, U = self._unpack(cells.numpy())
        return torch.from_numpy(L), torch.from_numpy(U)

    def branch_device_cells(self, n, cells, status, child_cells, counts, stream=0, child_dirty=None):
        import torch
        L, U = self._unpack(cells[:n].numpy())
        cl, cu = torch.zeros((2 * n, self.n_vars), dtype=torch.int32), torch.zeros((2 * n, self.n_vars), dtype=torch.int32)
        self.branch_device(n, torch.from_numpy(L), torch.from_numpy(U), None, status, cl, cu, None, counts, stream, child_dirty=child_dirty)
        k = int(counts[0])
        if k:
            child_cells[:k] = torch.from_numpy(self._pack(cl[:k].numpy(), cu[:k].numpy()))

    def propagate_device(self, n, lb_in, ub_in, lb_out, ub_out, active_in, active_out, status, stream=0, bits_in=None, bits_out=None, dirty=None, cells=False):
        if cells:
            import torch
            assert ub_in is None and ub_out is None and active_in is None and active_out is None
            L, U = self._unpack(lb_in[:n].numpy())
            lo, uo = torch.from_numpy(L.copy()), torch.from_numpy(U.copy())
            self.propagate_device(n, lo, uo, lo, uo, None, None, status, stream, dirty=dirty)
            ok = status[:n].numpy() != 0  # (a failed node's domains are unspecified — and may be empty, which no cell need carry)
            out = lb_in[:n].numpy().copy()
            out[ok] = self._pack(lo.numpy()[ok], uo.numpy()[ok])
            lb_out[:n] = torch.from_numpy(out)
            self.cell_launches = getattr(self, "cell_launches", 0) + 1
            return
        if dirty is not None:
            # a hint must name a variable of the node (or none: -1) and — the promise — giving that variable back its parent's bound must
            # leave a fixpoint; the oracle ignores hints (same results by definition), the stand-in only checks their form
            d = dirty[:n].numpy()
            assert ((d == -1) | ((d >= 0) & (d < self.n_vars))).all(), d
            self.hints_seen += int((d >= 0).sum())
        L, U = lb_in[:n].numpy().copy(), ub_in[:n].numpy().copy()
        A = None if active_in is None else active_in[:n].numpy().view(np.uint64).copy()
        bad = (L > U).any(axis=1)  # the device entry reports an empty input domain as a failed node
        L[bad], U[bad] = 0, 0
        lb, ub, act, st, s = self._m.consistency(L, U, A, check_dup=False)
        st = st.copy(); st[bad] = 0
        import torch
        lb_out[:n] = torch.from_numpy(lb)
        ub_out[:n] = torch.from_numpy(ub)
        if active_out is not None:
            active_out[:n] = torch.from_numpy(act.view(np.int64))
        status[:n] = torch.from_numpy(st)
        self._stats["steps"] += s["steps"]
        self._stats["nodes"] += n

    def branch_device(self, n, lb, ub, active, status, child_lb, child_ub, child_active, counts, stream=0, child_dirty=None):
        import torch
        from pcp_amd import search as S
        st = status[:n].numpy()
        unk = st == 2
        L, U = lb[:n].numpy()[unk], ub[:n].numpy()[unk]
        A = None if active is None else active[:n].numpy().view(np.uint64)[unk]
        k = 0
        if unk.any():
            cl, cu, ca = S.branch(L, U, A)
            if self._opts.get("branch_reverse"):
                cl, cu, ca = cl[::-1].copy(), cu[::-1].copy(), (None if ca is None else ca[::-1].copy())
            k = cl.shape[0]
            if child_dirty is not None:  # the variable each child was branched on: where it differs from its parent's (propagated) row
                pl_, pu_ = np.repeat(L, 2, axis=0), np.repeat(U, 2, axis=0)
                if self._opts.get("branch_reverse"):
                    pl_, pu_ = pl_[::-1], pu_[::-1]
                diff = (cl != pl_) | (cu != pu_)
                assert (diff.sum(axis=1) == 1).all()
                child_dirty[:k] = torch.from_numpy(diff.argmax(axis=1).astype(np.int32))
            child_lb[:k] = torch.from_numpy(cl)
            child_ub[:k] = torch.from_numpy(cu)
            if ca is not None:
                child_active[:k] = torch.from_numpy(ca.view(np.int64))
        counts[:] = torch.tensor([k, int((st == 1).sum()), int((st == 0).sum()), int(unk.sum()), 0], dtype=counts.dtype)
