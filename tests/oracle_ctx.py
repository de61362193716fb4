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
