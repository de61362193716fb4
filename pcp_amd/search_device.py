"""Device-resident subtree search: the open-node stack, propagation AND branching stay on the GPU; only four
counters per round cross PCIe ("removes the host round-trip per node").

Per round: the top ``batch`` open nodes of the stack are propagated in place (`pcp_propagate_device`), every Unknown
node is branched on the device (`pcp_branch_device`: FirstSmallestVar / MiddleVal / BinarySplit, folded, children
inherit the parent's `active` row), the batch is popped and the children are pushed.  With ``batch=1`` the node order
is exactly the reference's left-first DFS (search/engine/one_solution.rs:46-51, 92-105).
PyTorch provides the device buffers; every kernel is this repository's.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

from .model import FALSE, TRUE


@dataclass
class DeviceSearchStats:
    num_nodes: int = 0
    num_solution: int = 0
    num_failed_node: int = 0
    rounds: int = 0
    filter_steps: int = 0
    evaluated: int = 0
    max_open: int = 0
    solutions: List[np.ndarray] = field(default_factory=list)


class DeviceSearch:
    """The open nodes live in one device buffer of ``capacity`` rows as a list of SEGMENTS (start, length), the last
    one on top.  A round propagates the top ``n`` rows of the last segment in place and lets `pcp_branch_device` write
    the children straight above them, in reverse order, as a new segment: nothing is copied or reordered.  The popped
    parents leave a hole below the new segment; it is reclaimed when that segment is used up (LIFO)."""

    def __init__(self, ctx, batch: int = 1024, capacity: int = 0, device=None, implicit: bool = False, hints=None, cells: bool = False):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.batch = int(batch)
        self.dev = device if device is not None else torch.device("cuda", ctx.device)
        V, W = ctx.n_vars, max(ctx.words, 1)
        self.V, self.W = V, W
        self.cap = int(capacity) if capacity else 8 * self.batch + 64
        i32, i64, u8 = torch.int32, torch.int64, torch.uint8
        # cells: the open nodes are rows of packed cells (pcp_device_batch.cell_format PCP_CELLS_PACKED16; all-XNeqY models with a declared hull
        # within +-16383, implicit nodes): `lb` holds the cells, there is no `ub` — half the bytes per open node, same search node for node
        self.cells = bool(cells)
        if self.cells and (not implicit or getattr(ctx, "set_words", 0)):
            raise ValueError("cells=True needs implicit nodes in interval mode")
        self.lb = torch.empty((self.cap, V), dtype=i32, device=self.dev)
        self.ub = None if self.cells else torch.empty((self.cap, V), dtype=i32, device=self.dev)
        # implicit: a node record is its domains only — no `active` rows are kept, the engine derives liveness from the
        # domains (a node that descends from an all-active root has active = not entailed)
        self.implicit = bool(implicit) or not ctx.words
        self.act = None if self.implicit else torch.empty((self.cap, W), dtype=i64, device=self.dev)
        # set mode (IntervalSet domains, the reference's FDSpace): the node is its sets; lb/ub hold the sets' bounds after
        # propagation (the brancher's MiddleVal reads them)
        self.set_words = int(getattr(ctx, "set_words", 0))
        self.base = 0
        self.bits = torch.empty((self.cap, V, self.set_words), dtype=i64, device=self.dev) if self.set_words else None
        # one hint per open node (pcp_device_batch.dirty_var): a child is its parent's fixpoint with ONE variable branched on, so the engine
        # may start the child's propagation from that variable alone; the root has none (-1).  Interval mode, implicit nodes, an engine
        # that knows the field (the CPU stand-in of the tests does not).
        want = bool(getattr(ctx, "supports_hints", False)) and self.implicit and not self.set_words
        self.dirty = torch.full((self.cap,), -1, dtype=i32, device=self.dev) if (want if hints is None else (hints and want)) else None
        self.status = torch.zeros(self.batch, dtype=u8, device=self.dev)
        self.counts = torch.zeros(5, dtype=i32, device=self.dev)
        self.segs: List[List[int]] = []  # [start, length], bottom to top
        self.stats = DeviceSearchStats()

    # ---- the stack as the drivers see it ------------------------------------------------------------------------------
    @property
    def size(self) -> int:
        return sum(l for _, l in self.segs)

    @size.setter
    def size(self, n: int):
        """Rows [0, n) are the open nodes (used by balance_stacks after it has moved rows around a compacted stack)."""
        self.segs = [[0, int(n)]] if n > 0 else []

    def _top_row(self) -> int:
        return self.segs[-1][0] + self.segs[-1][1] if self.segs else 0

    def _stream(self) -> int:
        """The HIP stream the engine is launched on (torch's current stream of this GPU; 0 off the GPU: the CPU tests drive
        this class with an oracle-backed stand-in context)."""
        return self.torch.cuda.current_stream(self.dev).cuda_stream if self.dev.type == "cuda" else 0

    def _rows(self):
        rows = (self.lb, self.ub) if self.act is None else (self.lb, self.ub, self.act)
        if self.cells:
            rows = (self.lb,)
        if self.dirty is not None:
            rows = rows + (self.dirty,)
        return rows if self.bits is None else rows + (self.bits,)

    def _merge_top(self, want: int):
        """Close the holes under the top segments until the top segment holds `want` nodes (or is the only one): only
        the small segments on top are moved, never the bulk of the stack."""
        while len(self.segs) > 1 and self.segs[-1][1] < want:
            s2, l2 = self.segs.pop()
            s1, l1 = self.segs[-1]
            dest = s1 + l1
            if l2 and s2 != dest:
                for t in self._rows():
                    t[dest:dest + l2] = t[s2:s2 + l2].clone() if s2 < dest + l2 else t[s2:s2 + l2]
            self.segs[-1][1] = l1 + l2

    def compact(self):
        """Make the open nodes one segment starting at row 0 (order kept)."""
        if len(self.segs) == 1 and self.segs[0][0] == 0:
            return
        pos = 0
        for s, l in self.segs:
            if l and s != pos:
                for t in self._rows():
                    t[pos:pos + l] = t[s:s + l].clone() if s < pos + l else t[s:s + l]
            pos += l
        self.segs = [[0, pos]] if pos else []

    def reset(self, lb0, ub0, base: int = 0):
        """Start a new search: the stack holds the root (set mode: the variables as IntervalSet::new(lb0, ub0), value v = bit
        v - base, base = the hull's lower bound declared on the context)."""
        torch, ctx = self.torch, self.ctx
        from .engine import full_active
        if self.bits is not None:
            from .model import interval_bits
            self.base = int(base)
            self.bits[0] = torch.from_numpy(interval_bits(np.asarray(lb0), np.asarray(ub0), self.set_words, self.base).view(np.int64)).to(self.dev)
        if self.cells:
            l0 = torch.from_numpy(np.ascontiguousarray(lb0, np.int32)).to(self.dev).reshape(1, -1)
            u0 = torch.from_numpy(np.ascontiguousarray(ub0, np.int32)).to(self.dev).reshape(1, -1)
            ctx.pack_rows(l0, u0, self.lb[0:1], self._stream())
        else:
            self.lb[0] = torch.from_numpy(np.ascontiguousarray(lb0, np.int32)).to(self.dev)
            self.ub[0] = torch.from_numpy(np.ascontiguousarray(ub0, np.int32)).to(self.dev)
        if self.act is not None:
            self.act[0] = torch.from_numpy(full_active(1, ctx.n_units).view(np.int64)[0]).to(self.dev)
        if self.dirty is not None:
            self.dirty[0] = -1  # the root is propagated from scratch
        self.segs = [[0, 1]]
        self.stats = DeviceSearchStats()
        ctx.stats_reset(self._stream())

    def advance(self, all_solutions: bool = True, node_limit: int = 0, max_rounds: int = 0, keep_solutions: int = 0, batch: int = 0, stop_at: int = 0) -> bool:
        """Run rounds on the current stack until it is empty, a limit is hit, or (not all_solutions) a solution is
        found.  Returns True when the search is over (stack empty or solution found).
        ``node_limit`` is the search's StopNode limit (stop_node.rs:47-62): the node that reaches it is counted as a node and as nothing
        else.  ``stop_at`` only ends this call after that many nodes in total (a caller's chunk of a larger budget, e.g. one rank's share
        between two exchanges of parallel_search_device): every node's status counts."""
        torch, ctx, st = self.torch, self.ctx, self.stats
        stream = self._stream()
        batch = min(int(batch) if batch else self.batch, self.batch)
        rounds = 0
        done = False
        ctx.set_option("branch_reverse", 1)  # children arrive in pop order (a context-wide knob: restored below)
        while self.segs:
            if max_rounds and rounds >= max_rounds:
                break
            self._merge_top(batch)
            start, length = self.segs[-1]
            top = start + length
            # the children (at most 2n rows) go right above the popped parents: when the buffer is nearly full, first
            # squeeze out the holes, then take fewer nodes (a deeper, narrower dive) instead of overflowing
            if self.cap - top < 2 * min(batch, length):
                self.compact()
                start, length = self.segs[-1]
                top = start + length
            room = self.cap - top
            if room < 2:
                raise RuntimeError(f"open-node stack full ({self.size} of {self.cap}); raise `capacity`")
            n = min(batch, length, room // 2)
            for cap_nodes in (node_limit, stop_at):
                if cap_nodes:
                    n = min(n, cap_nodes - st.num_nodes)
            if n <= 0:
                break
            lo = top - n
            lb, ub = self.lb[lo:top], (None if self.cells else self.ub[lo:top])
            act = None if self.act is None else self.act[lo:top]
            status = self.status[:n]
            if self.cells:
                dirty = None if self.dirty is None else self.dirty[lo:top]
                ctx.propagate_device(n, lb, None, lb, None, None, None, status, stream, dirty=dirty, cells=True)
                ctx.branch_device_cells(n, lb, status, self.lb[top:], self.counts, stream, child_dirty=None if self.dirty is None else self.dirty[top:])
            elif self.bits is None and self.dirty is not None:
                ctx.propagate_device(n, lb, ub, lb, ub, act, act, status, stream, dirty=self.dirty[lo:top])
                ctx.branch_device(n, lb, ub, act, status, self.lb[top:], self.ub[top:], None if self.act is None else self.act[top:],
                                  self.counts, stream, child_dirty=self.dirty[top:])
            elif self.bits is None:
                ctx.propagate_device(n, lb, ub, lb, ub, act, act, status, stream)
                ctx.branch_device(n, lb, ub, act, status, self.lb[top:], self.ub[top:], None if self.act is None else self.act[top:],
                                  self.counts, stream)
            else:
                bits = self.bits[lo:top]
                ctx.propagate_device(n, None, None, lb, ub, act, act, status, stream, bits_in=bits, bits_out=bits)
                ctx.branch_device_set(n, bits, lb, ub, act, status, self.bits[top:], None if self.act is None else self.act[top:], self.counts, stream)
            n_children, n_true, n_false, _, n_other = (int(x) for x in self.counts.cpu().tolist())  # the round's only D2H sync
            if n_other:
                raise RuntimeError(f"{n_other} nodes were refused by the engine (bounds outside the declared hull): the search cannot continue")
            rounds += 1
            st.rounds += 1
            st.num_nodes += n
            limit_row_true = False
            if node_limit and st.num_nodes >= node_limit:
                # the node that reaches the limit (the last one in pop order: the lowest row of the round) is counted as a node, never as a
                # solution or a failure: StopNode hands EndOfSearch to the monitor (stop_node.rs:57-62 under Monitor, stop_node.rs:90-97)
                s_last = int(status[0].item())
                limit_row_true = s_last == TRUE
                n_true -= int(s_last == TRUE)
                n_false -= int(s_last == FALSE)
            st.num_solution += n_true
            st.num_failed_node += n_false
            if n_true and len(st.solutions) < keep_solutions:
                rows = torch.nonzero(status == TRUE).flatten()
                if limit_row_true:
                    rows = rows[rows != 0]  # (not counted: not kept either)
                rows = rows[: keep_solutions - len(st.solutions)]
                sol = ctx.unpack_rows(lb[rows].contiguous(), stream_ptr=stream)[0] if self.cells else lb[rows]
                for r in sol.cpu().numpy():
                    st.solutions.append(r)
            # pop the parents; the children, already in left-first order (branch_reverse), become the new top segment
            self.segs[-1][1] = length - n
            if self.segs[-1][1] == 0:
                self.segs.pop()
            if n_true and not all_solutions:
                done = True
                break
            if n_children:
                self.segs.append([top, n_children])
            st.max_open = max(st.max_open, self.size)
        ctx.set_option("branch_reverse", 0)
        s = ctx.stats_read(stream)
        st.filter_steps = s["steps"] + s["steps3"]
        st.evaluated = s.get("evaluated", 0)
        return done or not self.segs

    def run(self, lb0, ub0, all_solutions: bool = True, node_limit: int = 0, keep_solutions: int = 0, base: int = 0) -> DeviceSearchStats:
        self.reset(lb0, ub0, base)
        self.advance(all_solutions=all_solutions, node_limit=node_limit, keep_solutions=keep_solutions)
        return self.stats

    def top(self, k: int):
        """The k open nodes on top of the stack (device tensors, views)."""
        if self.segs and self.segs[-1][1] < min(k, self.size):
            self.compact()
        if not self.segs:
            return self.lb[0:0], (None if self.cells else self.ub[0:0]), (None if self.act is None else self.act[0:0])
        s, l = self.segs[-1]
        lo = max(s, s + l - k)
        return self.lb[lo:s + l], (None if self.cells else self.ub[lo:s + l]), (None if self.act is None else self.act[lo:s + l])
