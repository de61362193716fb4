"""CPU checks of the forest checker itself (oracle/forest_check.py) and of the oracle's subtree entry (`search_set(root_bits=...)`):
the simulators that turn the oracle's DFS records into the state a device-side forest must leave behind are run against the oracle's own
records on small boards, where whole trees (failures, solutions, backtracking) fit."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import forest_check as FC
from pcp_amd import model as M


@pytest.mark.parametrize("n", [6, 8])
def test_interval_stack_simulation_follows_the_oracle(n):
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    ss, _, rec, _ = om.search(lb0, ub0, all_solutions=True, max_records=100000)
    K = rec["status"].shape[0]
    assert K == ss["num_nodes"]
    # (the simulation asserts, node by node, that the top row it derives is the oracle's next input)
    stack, (nodes, sols, fails) = FC.simulate_interval_stack(rec, K)
    assert stack == [] and (nodes, sols, fails) == (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
    half, (_, s2, f2) = FC.simulate_interval_stack(rec, K // 2)
    assert len(half) >= 1 and s2 <= sols and f2 <= fails
    assert np.array_equal(half[-1][0], rec["lb_in"][K // 2]) and np.array_equal(half[-1][1], rec["ub_in"][K // 2])


@pytest.mark.parametrize("n", [6, 8])
def test_set_levels_simulation_and_subtree_roots(n):
    props = M.nqueens_props(n)
    om = orc.OracleModel(n, props)
    sw = 1
    lb0, ub0 = np.ones(n, np.int32), np.full(n, n, np.int32)
    ss, _, rec, _ = om.search_set(lb0, ub0, sw, 1, all_solutions=True, max_records=100000)
    K = rec["status"].shape[0]
    levels, (nodes, sols, fails) = FC.simulate_set_levels(rec, K)
    assert levels == [] and (nodes, sols, fails) == (ss["num_nodes"], ss["num_solution"], ss["num_failed_node"])
    # the same root as sets: the same tree
    rb = M.interval_bits(lb0, ub0, sw, 1)
    ss2, _, rec2, _ = om.search_set(None, None, sw, 1, all_solutions=True, max_records=8, root_bits=rb)
    assert ss2 == ss and np.array_equal(rec2["bits_in"], rec["bits_in"][:8])
    # a subtree: every visited node as a root gives the oracle's following nodes until that subtree ends (depth-first order)
    for k in (1, 2, 5, K // 3):
        sub, _, srec, _ = om.search_set(None, None, sw, 1, all_solutions=True, max_records=16, root_bits=rec["bits_in"][k])
        m = min(16, sub["num_nodes"])
        assert np.array_equal(srec["bits_in"][:m], rec["bits_in"][k:k + m]) and np.array_equal(srec["status"][:m], rec["status"][k:k + m])


def test_branching_rules():
    assert [FC.middle_val(*p) for p in ((1, 10), (2, 4), (1, 2), (-3, -2), (-3, 2))] == [5, 3, 1, -2, 0]  # binary_split.rs:110-133; `/` truncates
    assert FC.first_smallest_var(np.array([1, 5, 3, 3, 1])) == 2
