"""pcp_amd — MI355X-native propagation-fixpoint engine behind libpcp's propagator surface.

`pcp_amd.model`   host-side mirror of the reference's model-building API, lowered to C-ABI records
`pcp_amd.engine`  ctypes binding of libpcp_hip.so (include/pcp_hip.h); fails loudly without the HIP library
"""
from . import model  # noqa: F401

__all__ = ["model"]
