import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference's kernels built for x86; this container only)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the CPU-side libraries once (host lib, oracle, C-ABI library) if they are missing."""
    import __graft_entry__ as g
    g.build_cpu_libs()
