"""pcp_propagate_device_units (ABI v8): a BATCH of nodes that carry unary propagators of their own — the children of an Enumerate search in
interval mode.  `Enumerate::distribute` (search/branching/enumerate.rs:48-59) gives a node two children, `x = v` and `x != v`; on an
`Interval` domain `x != v` with v inside the domain removes nothing (x_neq_y.rs:82-93: a value goes only at a bound) and stays in the child's
cstore until v reaches a bound, so it cannot be folded into the child's bounds the way BinarySplit's constraints are.  Here the children of
many nodes are propagated in ONE launch, each with its own propagators, and compared with the oracle run on the SAME row over the model plus
that node's propagators allocated the way `Branch::distribute` allocates them (branch.rs:36-55): status and domains, bit-exact."""
import numpy as np
import pytest

from oracle import oracle as orc
from pcp_amd import model as M
from pcp_amd import search as S
import pcp_amd.engine as E

pytestmark = pytest.mark.gpu


def _unary(var, kind, value, const_first=False, off=0):
    """One pcp_prop over (variable + off) and Constant(value), in either operand order."""
    p = np.zeros(1, dtype=M.PROP_DTYPE)
    p["kind"] = kind
    p["var"][0] = [M.PCP_CONST, var, M.PCP_NOVAR] if const_first else [var, M.PCP_CONST, M.PCP_NOVAR]
    p["off"][0] = [value, off, 0] if const_first else [off, value, 0]
    return p


def _with_units(props, extra):
    """The model's props followed by a node's own props, each a unit of its own (Store::alloc order: appended behind the model's)."""
    if not len(extra):
        return props
    e = np.concatenate(extra)
    e["group"] = np.arange(len(e)) + (int(props["group"].max()) + 1 if len(props) else 0)
    return np.concatenate([props, e])


def test_enumerate_children_of_nqueens8_in_one_launch():
    import torch
    n = 8
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    _, _, rec, _ = om.search(lb0, ub0, all_solutions=True, node_limit=120, max_records=120)
    unk = rec["status"] == 2
    L, U = rec["lb_out"][unk], rec["ub_out"][unk]  # propagated Unknown nodes: the parents
    var = S.first_smallest_var(L, U)
    val = S.middle_val(L[np.arange(len(var)), var], U[np.arange(len(var)), var])
    rows_l, rows_u, units = [], [], []
    rng = np.random.default_rng(3)
    for i in range(len(var)):
        x, v = int(var[i]), int(val[i])
        # Enumerate's two children over the parent's fixpoint: x = v | x != v  (enumerate.rs:48-59)
        rows_l += [L[i], L[i]]; rows_u += [U[i], U[i]]
        units += [[_unary(x, M.EQ, v)], [_unary(x, M.NEQ, v)]]
        if i % 3 == 0:  # a grandchild down the right branches: two values excluded, the second written constant-first with an Addition offset
            w = int(rng.integers(int(L[i, x]), int(U[i, x]) + 1))
            rows_l.append(L[i]); rows_u.append(U[i])
            units.append([_unary(x, M.NEQ, v), _unary(x, M.NEQ, w + 2, const_first=True, off=2)])
        if i % 4 == 0:  # bounds as propagators: x < v + 1 and v - 1 < x (what BinarySplit would have folded), and a node without any
            rows_l += [L[i], L[i], L[i]]; rows_u += [U[i], U[i], U[i]]
            units += [[_unary(x, M.LT, v + 1)], [_unary(x, M.LT, v - 1, const_first=True)], []]
    Ln, Un = np.ascontiguousarray(np.stack(rows_l)), np.ascontiguousarray(np.stack(rows_u))
    N = Ln.shape[0]
    assert N >= 96
    off = np.zeros(N + 1, np.int32)
    off[1:] = np.cumsum([len(u) for u in units])
    flat = np.concatenate([p for u in units for p in u])
    ctx = E.Context(0)
    try:
        ctx.set_model(n, props)
        dev = torch.device("cuda", 0)
        lb, ub = torch.from_numpy(Ln).to(dev), torch.from_numpy(Un).to(dev)
        st = torch.full((N,), 255, dtype=torch.uint8, device=dev)
        d_off = torch.from_numpy(off).to(dev)
        d_units = torch.from_numpy(flat.view(np.uint8).copy()).to(dev)
        ctx.propagate_device_units(N, lb, ub, lb, ub, None, None, st, d_off, d_units)
        assert ctx.last_plan()["path"] == 4
        torch.cuda.synchronize()
        g_lb, g_ub, g_st = lb.cpu().numpy(), ub.cpu().numpy(), st.cpu().numpy()
        seen = set()
        for i in range(N):
            omi = orc.OracleModel(n, _with_units(props, units[i]))
            ref_i = omi.consistency(Ln[i:i + 1], Un[i:i + 1], None)
            r_lb, r_ub, r_st = ref_i[0], ref_i[1], ref_i[3]
            assert int(r_st[0]) == int(g_st[i]), (i, units[i], int(r_st[0]), int(g_st[i]))
            if r_st[0] != 0:
                assert np.array_equal(r_lb[0], g_lb[i]) and np.array_equal(r_ub[0], g_ub[i]), i
            seen.add(int(r_st[0]))
        assert {0, 2} <= seen  # failures and open nodes both occur
        # an interior x != v really was kept: some right child left its variable's bounds alone and is Unknown because of its own propagator alone?
        # (at least: some node's fixpoint differs from the fixpoint of the same row without its propagators)
        base = om.consistency(Ln, Un, None)
        assert any(int(base[3][i]) != int(g_st[i]) or not np.array_equal(base[0][i], g_lb[i]) or not np.array_equal(base[1][i], g_ub[i]) for i in range(N))
        # no offsets = the plain entry
        lb2, ub2 = torch.from_numpy(Ln).to(dev), torch.from_numpy(Un).to(dev)
        st2 = torch.zeros(N, dtype=torch.uint8, device=dev)
        ctx.propagate_device_units(N, lb2, ub2, lb2, ub2, None, None, st2, None, None)
        torch.cuda.synchronize()
        assert np.array_equal(st2.cpu().numpy(), base[3])
        # a malformed unit (two constants) refuses ITS node and no other
        bad = flat.copy()
        k = int(off[1])
        assert off[2] > off[1]
        bad["var"][k] = [M.PCP_CONST, M.PCP_CONST, M.PCP_NOVAR]
        lb3, ub3 = torch.from_numpy(Ln).to(dev), torch.from_numpy(Un).to(dev)
        st3 = torch.zeros(N, dtype=torch.uint8, device=dev)
        ctx.propagate_device_units(N, lb3, ub3, lb3, ub3, None, None, st3, d_off, torch.from_numpy(bad.view(np.uint8).copy()).to(dev))
        torch.cuda.synchronize()
        s3 = st3.cpu().numpy()
        assert s3[1] not in (0, 1, 2) and np.array_equal(np.delete(s3, 1), np.delete(g_st, 1))
    finally:
        ctx.close()


def test_node_units_are_refused_where_no_kernel_reads_them():
    import torch
    n = 200  # 200 variables: not a small store
    props = M.nqueens_props(n)
    ctx = E.Context(0)
    try:
        ctx.set_model(n, props)
        dev = torch.device("cuda", 0)
        lb = torch.ones((2, n), dtype=torch.int32, device=dev)
        ub = torch.full((2, n), n, dtype=torch.int32, device=dev)
        st = torch.zeros(2, dtype=torch.uint8, device=dev)
        off = torch.tensor([0, 1, 1], dtype=torch.int32, device=dev)
        u = torch.from_numpy(_unary(0, M.NEQ, 5).view(np.uint8).copy()).to(dev)
        with pytest.raises(E.PcpError) as ei:
            ctx.propagate_device_units(2, lb, ub, lb, ub, None, None, st, off, u)
        assert ei.value.code == -5  # PCP_ERR_UNSUPPORTED
    finally:
        ctx.close()
