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
    # tests/wide_analysis.cpp is built lazily by the one test module that uses it (test_wide_emulation._lib), which skips on a build failure:
    # a box without libgomp, or an edit to the archived experiment header it includes, must not fail every test of the session


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


# ---------------------------------------------------------------------------------------------------------------------------------
# Skip budget for `-m gpu` (round 4 lost its 21 strongest oracle tests to a skip nobody saw: the suite stayed green with 33 skips
# instead of 12).  On a box with a GPU every skip must (a) carry one of the reasons below, (b) never hit a test that compares the
# DEFAULT configuration with the oracle / the reference, and (c) -- when the whole suite runs -- match the expected count per reason
# exactly.  Anything else turns the session red.  FLX_ALLOW_SKIPS=1 switches the guard off (ad-hoc runs).
EXPECTED_GPU_SKIPS = {
    # tests/test_gpu_parity.py::test_full_size_properties_and_determinism: 3 non-kitchen workloads x the 4 modes that are not (shadow 4, xcd 0, overlap 2)
    "the A/B variants are exercised at full size on the kitchen scene only": 12,
    # tests/test_gpu_parity.py::test_full_size_free_run_vs_oracle: 3 workloads x the 6 non-default modes
    "default configuration only (the variants run on the small scenes)": 18,
}
_skips = []            # (nodeid, reason, protected)


def _gpu_session(config):
    if os.environ.get("FLX_ALLOW_SKIPS") == "1":
        return False
    if "gpu" not in (config.getoption("markexpr") or "") or "not gpu" in (config.getoption("markexpr") or ""):
        return False
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if rep.skipped and rep.when in ("setup", "call"):
        reason = rep.longrepr[2] if isinstance(rep.longrepr, tuple) else str(rep.longrepr)
        reason = reason[len("Skipped: "):] if reason.startswith("Skipped: ") else reason
        mode = getattr(getattr(item, "callspec", None), "params", {}).get("trace_mode", None)
        default_mode = mode is None
        if mode is not None:
            import test_gpu_parity
            default_mode = tuple(mode) == tuple(test_gpu_parity.DEFAULT_MODE)
        protected = default_mode and any(k in item.name for k in ("vs_oracle", "vs_reference", "reference_on_gfx950"))
        _skips.append((item.nodeid, reason, protected))


def _skip_violations(config, whole_suite):
    bad = []
    counts = {}
    for nodeid, reason, protected in _skips:
        counts[reason] = counts.get(reason, 0) + 1
        if protected:
            bad.append(f"a default-configuration parity test was skipped: {nodeid} ({reason})")
        elif reason not in EXPECTED_GPU_SKIPS:
            bad.append(f"skip with a reason that is not in tests/conftest.py:EXPECTED_GPU_SKIPS: {nodeid} ({reason})")
    if whole_suite:
        for reason, n in EXPECTED_GPU_SKIPS.items():
            if counts.get(reason, 0) != n:
                bad.append(f"expected {n} skips with reason {reason!r}, saw {counts.get(reason, 0)}")
    return bad


def _whole_suite(config):
    args = [os.path.abspath(a) for a in config.args]
    return not config.getoption("keyword") and all(os.path.isdir(a) for a in args)


def pytest_sessionfinish(session, exitstatus):
    if _gpu_session(session.config) and _skip_violations(session.config, _whole_suite(session.config)) and session.exitstatus == 0:
        session.exitstatus = 1


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _gpu_session(config):
        return
    bad = _skip_violations(config, _whole_suite(config))
    terminalreporter.write_sep("=", f"skip budget: {len(_skips)} skipped, {len(bad)} violations")
    for b in bad:
        terminalreporter.write_line("SKIP BUDGET VIOLATION: " + b)
