"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/fluctus_hip.h declares, wire structs have the reference's sizes, and the product path fails
loudly (no CPU fallback) when no GPU is present."""
import os
import re
import numpy as np
import pytest
from fluctus_amd import device, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "fluctus_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(flx_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported():
    L = device.lib()
    decl = _declared()
    assert len(decl) >= 30
    for s in decl:
        assert hasattr(L, s), f"{s} declared in include/fluctus_hip.h but not exported"
    assert sorted(device.SYMBOLS) == decl


def test_wire_sizes_match_reference_abi():
    # reference: src/geom.h (sizes measured in SURVEY 8(a) A0)
    assert wire.TRIANGLE.itemsize == 160 and wire.TRIANGLE.fields["matId"][1] == 144
    assert wire.NODE.itemsize == 48 and wire.NODE.fields["parent"][1] == 32 and wire.NODE.fields["nPrims"][1] == 40
    assert wire.MATERIAL.itemsize == 80 and wire.MATERIAL.fields["type"][1] == 68
    assert wire.RENDER_PARAMS.itemsize == 240
    f = wire.RENDER_PARAMS.fields
    assert f["camera"][1] == 96 and f["width"][1] == 184 and f["maxBounces"][1] == 208 and f["worldRadius"][1] == 228
    assert wire.COUNTERS.itemsize == 32


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device|flx_create failed"):
        device.HipContext(1024)
