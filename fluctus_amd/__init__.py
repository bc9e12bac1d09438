"""fluctus_amd -- MI355X-native wavefront path-tracing hot path (HIP/CDNA4) behind a C ABI.

Python here is harness plumbing only (ctypes bindings for tests / bench.py); the product is
libfluctus_hip.so (fluctus_amd/csrc, C ABI in include/fluctus_hip.h) plus the C++ host side
(fluctus_amd/host -> libfluctus_host.so).
"""
from .wire import (TRIANGLE, NODE, MATERIAL, TEXDESC, RENDER_PARAMS, COUNTERS, default_params,
                   BXDF, COL, Q)
