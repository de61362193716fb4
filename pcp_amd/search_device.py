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

from .model import TRUE


@dataclass
class DeviceSearchStats:
    num_nodes: int = 0
    num_solution: int = 0
    num_failed_node: int = 0
    rounds: int = 0
    filter_steps: int = 0
    max_open: int = 0
    solutions: List[np.ndarray] = field(default_factory=list)


class DeviceSearch:
    def __init__(self, ctx, batch: int = 1024, capacity: int = 0, device=None):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.batch = int(batch)
        self.dev = device if device is not None else torch.device("cuda", ctx.device)
        V, W = ctx.n_vars, max(ctx.words, 1)
        self.V, self.W = V, W
        self.cap = int(capacity) if capacity else 8 * self.batch + 64
        i32, i64, u8 = torch.int32, torch.int64, torch.uint8
        self.lb = torch.empty((self.cap, V), dtype=i32, device=self.dev)
        self.ub = torch.empty((self.cap, V), dtype=i32, device=self.dev)
        self.act = torch.empty((self.cap, W), dtype=i64, device=self.dev)
        self.status = torch.zeros(self.batch, dtype=u8, device=self.dev)
        self.c_lb = torch.empty((2 * self.batch, V), dtype=i32, device=self.dev)
        self.c_ub = torch.empty((2 * self.batch, V), dtype=i32, device=self.dev)
        self.c_act = torch.empty((2 * self.batch, W), dtype=i64, device=self.dev)
        self.counts = torch.zeros(4, dtype=i32, device=self.dev)

    def reset(self, lb0, ub0):
        """Start a new search: the stack holds the root."""
        torch, ctx = self.torch, self.ctx
        from .engine import full_active
        self.lb[0] = torch.from_numpy(np.ascontiguousarray(lb0, np.int32)).to(self.dev)
        self.ub[0] = torch.from_numpy(np.ascontiguousarray(ub0, np.int32)).to(self.dev)
        if ctx.words:
            self.act[0] = torch.from_numpy(full_active(1, ctx.n_units).view(np.int64)[0]).to(self.dev)
        self.size = 1
        self.stats = DeviceSearchStats()
        ctx.stats_reset(torch.cuda.current_stream(self.dev).cuda_stream)

    def advance(self, all_solutions: bool = True, node_limit: int = 0, max_rounds: int = 0, keep_solutions: int = 0, batch: int = 0) -> bool:
        """Run rounds on the current stack until it is empty, a limit is hit, or (not all_solutions) a solution is
        found.  Returns True when the search is over (stack empty or solution found)."""
        torch, ctx, st = self.torch, self.ctx, self.stats
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        batch = min(int(batch) if batch else self.batch, self.batch)
        rounds = 0
        done = False
        while self.size > 0:
            if max_rounds and rounds >= max_rounds:
                break
            size = self.size
            # popping n and pushing at most 2n children must fit: when the stack is nearly full, take fewer nodes
            # (a deeper, narrower dive) instead of overflowing
            room = self.cap - size
            if room <= 0:
                raise RuntimeError(f"open-node stack full ({size} of {self.cap}); raise `capacity`")
            n = min(batch, size, max(1, room // 2))
            if node_limit:
                n = min(n, node_limit - st.num_nodes)
                if n <= 0:
                    break
            lo = size - n
            lb, ub, act = self.lb[lo:size], self.ub[lo:size], self.act[lo:size]
            status = self.status[:n]
            ctx.propagate_device(n, lb, ub, lb, ub, act if ctx.words else None, act if ctx.words else None, status, stream)
            ctx.branch_device(n, lb, ub, act if ctx.words else None, status, self.c_lb, self.c_ub, self.c_act if ctx.words else None,
                              self.counts, stream)
            n_children, n_true, n_false, _ = (int(x) for x in self.counts.cpu().tolist())  # the round's only D2H sync
            rounds += 1
            st.rounds += 1
            st.num_nodes += n
            st.num_solution += n_true
            st.num_failed_node += n_false
            if n_true and len(st.solutions) < keep_solutions:
                rows = torch.nonzero(status == TRUE).flatten()[: keep_solutions - len(st.solutions)]
                for r in lb[rows].cpu().numpy():
                    st.solutions.append(r)
            size = lo
            if n_true and not all_solutions:
                self.size = size
                done = True
                break
            if n_children:
                # reversed, so that the first node's left child ends on top of the stack (left-first DFS)
                self.lb[size:size + n_children] = torch.flip(self.c_lb[:n_children], dims=[0])
                self.ub[size:size + n_children] = torch.flip(self.c_ub[:n_children], dims=[0])
                if ctx.words:
                    self.act[size:size + n_children] = torch.flip(self.c_act[:n_children], dims=[0])
                size += n_children
            self.size = size
            st.max_open = max(st.max_open, size)
        s = ctx.stats_read(stream)
        st.filter_steps = s["steps"] + s["steps3"]
        return done or self.size == 0

    def run(self, lb0, ub0, all_solutions: bool = True, node_limit: int = 0, keep_solutions: int = 0) -> DeviceSearchStats:
        self.reset(lb0, ub0)
        self.advance(all_solutions=all_solutions, node_limit=node_limit, keep_solutions=keep_solutions)
        return self.stats

    def top(self, k: int):
        """The k open nodes on top of the stack (device tensors, views)."""
        lo = max(0, self.size - k)
        return self.lb[lo:self.size], self.ub[lo:self.size], self.act[lo:self.size]
