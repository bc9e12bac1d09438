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
    build_fake_rccl()
    build_wide_analysis()


def build_wide_analysis(force=False):
    """Test / experiment infrastructure only: tests/wide_analysis.cpp -> tests/_build/libwide_analysis.so (host emulation of the 4-wide traversal on
    the product's own tree builder; tests/test_wide_emulation.py, scripts/exp_tree_opt.py, scripts/exp_occluder_cache.py)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "wide_analysis.cpp")
    deps = [src, os.path.join(ROOT, "fluctus_amd", "csrc", "flx_wide.h"), os.path.join(ROOT, "scripts", "experiments", "flx_wide_opt.h"),
            os.path.join(ROOT, "include", "flx_math.h"), os.path.join(ROOT, "include", "fluctus_wire.h")]
    out = os.path.join(ROOT, "tests", "_build", "libwide_analysis.so")
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", src, "-o", out], check=True)
    return out


def build_fake_rccl(force=False):
    """Test infrastructure only (not part of the product build): tests/fake_rccl.cpp -> tests/_build/libfake_rccl.so, a librccl stand-in that
    moves tiles between host threads on one device (tests/test_gpu_rccl_fake.py; selected with FLX_RCCL_LIB + FLX_ALLOW_RCCL_OVERRIDE=1)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "fake_rccl.cpp")
    out = os.path.join(ROOT, "tests", "_build", "libfake_rccl.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    if not os.path.isdir("/opt/rocm/include"):
        return None
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", out,
                    "-L/opt/rocm/lib", "-lamdhip64", "-pthread"], check=True)
    return out
