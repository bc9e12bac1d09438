"""The `-m gpu` skip budget of tests/conftest.py, exercised on the CPU: the guard must notice a protected skip, an unknown reason and a
wrong count (round 4's failure mode: 33 skips instead of 12 and a green suite)."""
import conftest


def _run(skips, whole):
    saved = list(conftest._skips)
    conftest._skips[:] = skips
    try:
        return conftest._skip_violations(None, whole)
    finally:
        conftest._skips[:] = saved


def test_expected_skips_pass():
    skips = [(f"t{i}", r, False) for r, n in conftest.EXPECTED_GPU_SKIPS.items() for i in range(n)]
    assert _run(skips, True) == []


def test_round4_failure_mode_is_caught():
    r = "default configuration only (the variants run on the small scenes)"
    skips = [(f"t{i}", k, False) for k, n in conftest.EXPECTED_GPU_SKIPS.items() for i in range(n)]
    skips += [(f"tests/test_gpu_parity.py::test_full_size_free_run_vs_oracle[wide-shadow-w{i}]", r, True) for i in range(3)]
    bad = _run(skips, True)
    assert sum("default-configuration parity test was skipped" in b for b in bad) == 3
    assert any("expected 18 skips" in b for b in bad)


def test_unknown_reason_and_partial_runs():
    assert len(_run([("x", "fixture missing", False)], False)) == 1
    assert _run([("x", "the A/B variants are exercised at full size on the kitchen scene only", False)], False) == []      # no count check on a partial run
