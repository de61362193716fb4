"""Config 5's OWN engine at its own size: the RCCL worklist search (`pcp_amd.distributed.parallel_search_device` over
`pcp_amd.search_device.DeviceSearch`) on N-queens n = 1000, as `bench.py`'s `c5_*` legs run it — implicit nodes, hints on, 64 nodes per
round, int32 rows AND rows of packed cells.

What is pinned, node for node:
  * every row a round hands the engine is captured BEFORE the launch; the oracle (`Store::consistency`, propagation/store.rs:125-164,
    247-257 — which knows nothing of hints or cells) is run on the same rows: status equal, fixpoint domains bit-identical;
  * every child row the device brancher writes equals the host brancher's (`Brancher<FirstSmallestVar, MiddleVal, BinarySplit>::enter`,
    search/branching/brancher.rs:52-71, binary_split.rs:46-57) over the device's own fixpoints, in pop order (left first:
    search/engine/one_solution.rs:46-51), and every child's hint names the branched variable (branch.rs:36-55: ONE branch propagator);
  * the launch took the all-XNeqY kernel (plan.path == 1) and an open-node record is 8 004 B / 4 004 B;
  * `parallel_search_device` through a ONE-rank `nccl` (= RCCL) group under a node budget leaves the same counters as `DeviceSearch.run`
    under the same StopNode limit (stop_node.rs:47-62), in both formats, and both formats visit the same tree.
Bit-exact (integer work)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
from pcp_amd import search as S
import pcp_amd.engine as E

pytestmark = pytest.mark.gpu

N = 1000
BATCH = 64


@pytest.fixture(scope="module")
def env():
    import torch
    ctx = E.Context(0)
    props = M.nqueens_props(N)
    ctx.set_model(N, props)
    ctx.set_hull(1, N)
    om = orc.OracleModel(N, props)
    yield ctx, om, torch
    ctx.close()


class _OneRank:
    """What seed_frontier asks of a process group, for a single rank (no communication happens at world size 1)."""
    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_rank():
        return 0


class _Recorder:
    """Wraps the context's device entries: clones what goes in and what comes out of every launch of a round."""

    def __init__(self, ctx, torch, cells):
        self.ctx, self.torch, self.cells = ctx, torch, cells
        self.rounds = []
        self._orig = (ctx.propagate_device, ctx.branch_device, ctx.branch_device_cells)

    def _rows(self, t):
        """int32 (lb, ub) numpy rows of a tensor of rows (cells: unpacked through pcp_unpack_rows)."""
        if self.cells:
            l, u = self.ctx.unpack_rows(t.contiguous())
            return l.cpu().numpy(), u.cpu().numpy()
        return t[0].cpu().numpy().copy(), t[1].cpu().numpy().copy()

    def __enter__(self):
        ctx, torch = self.ctx, self.torch
        p0, b0, bc0 = self._orig

        def propagate_device(n, lb_in, ub_in, lb_out, ub_out, a_in, a_out, status, stream_ptr=0, **kw):
            assert bool(kw.get("cells", False)) == self.cells and kw.get("dirty") is not None
            rec = {"n": n, "in": self._rows(lb_in if self.cells else (lb_in, ub_in)), "hint": kw["dirty"].cpu().numpy().copy()}
            p0(n, lb_in, ub_in, lb_out, ub_out, a_in, a_out, status, stream_ptr, **kw)
            rec["plan"] = ctx.last_plan()
            torch.cuda.synchronize()
            rec["out"] = self._rows(lb_out if self.cells else (lb_out, ub_out))
            rec["status"] = status[:n].cpu().numpy().copy()
            self.rounds.append(rec)

        def branch_device(n, lb, ub, act, status, c_lb, c_ub, c_act, counts, stream_ptr=0, child_dirty=None):
            b0(n, lb, ub, act, status, c_lb, c_ub, c_act, counts, stream_ptr, child_dirty=child_dirty)
            torch.cuda.synchronize()
            nc = int(counts[0].item())
            self.rounds[-1]["children"] = (c_lb[:nc].cpu().numpy().copy(), c_ub[:nc].cpu().numpy().copy())
            self.rounds[-1]["child_hint"] = child_dirty[:nc].cpu().numpy().copy()

        def branch_device_cells(n, cells, status, c_cells, counts, stream_ptr=0, child_dirty=None):
            bc0(n, cells, status, c_cells, counts, stream_ptr, child_dirty=child_dirty)
            torch.cuda.synchronize()
            nc = int(counts[0].item())
            self.rounds[-1]["children"] = self._rows(c_cells[:nc])
            self.rounds[-1]["child_hint"] = child_dirty[:nc].cpu().numpy().copy()

        ctx.propagate_device, ctx.branch_device, ctx.branch_device_cells = propagate_device, branch_device, branch_device_cells
        return self

    def __exit__(self, *exc):
        for name in ("propagate_device", "branch_device", "branch_device_cells"):
            self.ctx.__dict__.pop(name, None)  # (instance attributes shadowing the class's methods)


@pytest.mark.parametrize("cells", [False, True])
def test_worklist_rounds_node_for_node_at_n1000(env, cells):
    from pcp_amd import distributed as D
    from pcp_amd.search_device import DeviceSearch
    ctx, om, torch = env
    lb0, ub0 = np.ones(N, np.int32), np.full(N, N, np.int32)
    ds = DeviceSearch(ctx, batch=BATCH, capacity=4096, implicit=True, hints=True, cells=cells)
    assert ds.dirty is not None and ds.cells == cells
    rec_bytes = sum(int(np.prod(t.shape[1:])) * t.element_size() for t in ds._rows())
    assert rec_bytes == (4 * N + 4 if cells else 8 * N + 4)  # an open-node record: the domains + the hint (what balance_stacks moves)
    # the frontier the worklist engine is seeded with (seed_frontier: the same breadth-first expansion on every rank), then two rounds of 64
    D.seed_frontier(ds, lb0, ub0, _OneRank, BATCH)
    assert ds.size >= BATCH
    seeded = ds.stats.num_nodes
    with _Recorder(ctx, torch, cells) as rec:
        ds.advance(all_solutions=True, max_rounds=2)
    assert len(rec.rounds) == 2 and ds.stats.num_nodes == seeded + sum(r["n"] for r in rec.rounds)
    checked = 0
    for k, r in enumerate(rec.rounds):
        n = r["n"]
        assert n == BATCH and r["plan"]["path"] == 1 and r["plan"]["implicit_active"] == 1, r["plan"]
        Lin, Uin = r["in"]
        take = n if k == 0 else 32  # 96 nodes against the oracle (≈ 0.2 s each)
        ref = om.consistency(np.ascontiguousarray(Lin[:take]), np.ascontiguousarray(Uin[:take]), None, check_dup=False)
        st = r["status"]
        assert np.array_equal(ref[3], st[:take]), (k, ref[3], st[:take])
        ok = ref[3] != 0
        assert np.array_equal(ref[0][ok], r["out"][0][:take][ok]) and np.array_equal(ref[1][ok], r["out"][1][:take][ok]), f"round {k}: fixpoint domains"
        checked += take
        # every row came with a hint, and the hint is honest: the row differs from a fixpoint in that variable only — i.e. giving the
        # variable ANY wider domain than the one the brancher cut cannot be checked here, but the engine's result above IS the oracle's
        assert ((r["hint"] >= 0) & (r["hint"] < N)).all()
        # the children: the host brancher over the device's own fixpoints, reversed into pop order (option branch_reverse)
        unk = st == 2
        hl, hu, _ = S.branch(r["out"][0][unk], r["out"][1][unk], None)
        cl, cu = r["children"]
        assert cl.shape[0] == 2 * int(unk.sum())
        assert np.array_equal(cl, hl[::-1]) and np.array_equal(cu, hu[::-1]), f"round {k}: children"
        var = S.first_smallest_var(r["out"][0][unk], r["out"][1][unk])
        assert np.array_equal(r["child_hint"], np.repeat(var, 2)[::-1]), f"round {k}: child hints"
    assert checked == 96
    # the second round's inputs ARE the first round's children (the top 64 of them, in pop order): the stack is the worklist
    c0 = rec.rounds[0]["children"]
    assert np.array_equal(rec.rounds[1]["in"][0], c0[0][-BATCH:]) and np.array_equal(rec.rounds[1]["in"][1], c0[1][-BATCH:])


def test_one_rank_rccl_worklist_equals_device_search_under_a_budget(env):
    """parallel_search_device through a 1-rank nccl group (all_gather, batch_isend_irecv plan, all_reduce all run; nothing to move) at n = 1000
    under a node budget: the counters are DeviceSearch.run's under the same StopNode limit — in both formats — and the formats agree."""
    import torch.distributed as dist
    from pcp_amd import distributed as D
    from pcp_amd.search_device import DeviceSearch
    ctx, om, torch = env
    lb0, ub0 = np.ones(N, np.int32), np.full(N, N, np.int32)
    budget = 3000
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29547"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    got = {}
    try:
        for cells in (False, True):
            solo = DeviceSearch(ctx, batch=BATCH, capacity=budget + 8 * BATCH, implicit=True, hints=True, cells=cells).run(lb0, ub0, all_solutions=True, node_limit=budget)
            ds = DeviceSearch(ctx, batch=BATCH, capacity=budget + 8 * BATCH, implicit=True, hints=True, cells=cells)
            info = {}
            nodes, sols, fails, steps, moved = D.parallel_search_device(ds, lb0, ub0, dist, all_solutions=True, node_limit=budget, rounds_per_exchange=4, info=info)
            assert (nodes, sols, fails) == (solo.num_nodes, solo.num_solution, solo.num_failed_node), (cells, nodes, sols, fails, solo)
            assert nodes == budget and moved == 0 and info["exchanges"] >= 3 and steps > 0
            assert ctx.last_plan()["path"] == 1
            got[cells] = (nodes, sols, fails)
    finally:
        dist.destroy_process_group()
    assert got[False] == got[True]
