"""The N > 1 branches of flx_gather / flx_gather_local (api.hip: grouped ncclSend on the peers, ncclRecv on the root) on a 1-GPU box.

Real RCCL refuses a communicator with duplicate devices, so with one GPU those branches never run.  tests/fake_rccl.cpp is a stand-in
transport with the same entry points (host threads / several communicators on device 0, the receiver copies device-to-device in stream
order); FLX_RCCL_LIB makes libfluctus_hip.so bind it.  What is tested is OUR side: which rank posts what, tile sizes of the ragged
partition, staging offsets, root != 0, the de-interleave, error paths that must return instead of hanging, groups that must be closed.
The reference has no counterpart (single device: src/clcontext.cpp:25-29)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "_build", "libfake_rccl.so")


def _run(n, root, mode, timeout=300, **extra):
    assert os.path.exists(FAKE), "tests/_build/libfake_rccl.so missing (tests/conftest.py: build_fake_rccl)"
    env = dict(os.environ, FLX_RCCL_LIB=FAKE, FLX_ALLOW_RCCL_OVERRIDE="1", FLX_NO_BUILD="1", **{k: str(v) for k, v in extra.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_fake_driver.py"), str(n), str(root), mode],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, f"driver printed no result (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}"
    return r.returncode, json.loads(lines[-1]), r.stderr


@pytest.mark.parametrize("n,root", [(3, 0), (3, 2), (8, 5)])
def test_gather_across_rank_threads(n, root):
    """One thread per rank: ncclCommInitRank rendezvous, n - 1 sends, n - 1 receives on the root, image == the interleaved tiles."""
    rc, out, err = _run(n, root, "threads")
    assert rc == 0 and not out["errors"], (out, err[-1500:])
    assert out["ok"], out
    assert out["tiles_differ"]
    send, recv, gstart, gend, init_rank = out["counters"][:5]
    assert (send, recv, init_rank) == (n - 1, n - 1, n), out["counters"]
    assert gstart == gend == n and set(out["depth"]) == {0}
    assert sorted(map(tuple, out["info"])) == [(n, r) for r in range(n)]       # ncclCommCount / ncclCommUserRank


@pytest.mark.parametrize("n,root", [(3, 1), (8, 0)])
def test_gather_local_through_the_communicator(n, root):
    """One thread, ncclCommInitAll: the sends and receives of all ranks inside ONE group."""
    rc, out, err = _run(n, root, "local")
    assert rc == 0 and not out["errors"], (out, err[-1500:])
    assert out["ok"], out
    assert out["counters"][0] == n - 1 and out["counters"][1] == n - 1 and out["counters"][5] == 1
    assert out["counters"][2] == out["counters"][3] == 1 and set(out["depth"]) == {0}


def test_wrong_bytes_on_the_wire_are_noticed():
    """A transport that delivers a wrong value: the comparison this test (and bench.py) makes catches it."""
    rc, out, _ = _run(3, 0, "threads", FAKE_RCCL_BREAK=1)
    assert not out["errors"] and out["ok"] is False


def test_lost_message_returns_an_error_instead_of_hanging():
    rc, out, _ = _run(3, 1, "threads", FAKE_RCCL_BREAK=2, FAKE_RCCL_TIMEOUT_MS=1500)
    assert any("rank 1" in e and "GroupEnd" in e for e in out["errors"]), out
    assert set(out["depth"]) == {0}


def test_failure_inside_the_group_still_closes_it():
    """ncclRecv fails between ncclGroupStart and ncclGroupEnd (api.hip: NcclGroup): the call returns the error AND the group is closed
    on every rank; the peers, whose sends are never matched, fail too instead of waiting forever."""
    rc, out, _ = _run(3, 0, "threads", FAKE_RCCL_BREAK=3, FAKE_RCCL_TIMEOUT_MS=1500)
    assert any("rank 0" in e and "Recv" in e for e in out["errors"]), out
    assert len(out["errors"]) == 3 and set(out["depth"]) == {0}
    assert out["counters"][2] == out["counters"][3]
    rc, out, _ = _run(3, 1, "local", FAKE_RCCL_BREAK=3, FAKE_RCCL_TIMEOUT_MS=1500)
    assert out["errors"] and set(out["depth"]) == {0} and out["counters"][2] == out["counters"][3] == 1


def test_bench_exits_nonzero_when_the_native_gather_fails(tmp_path):
    """bench.py as the driver launches it for N > 1 (one rank here, FLX_FORCE_DIST=1) with a transport whose communicator cannot be
    formed: the JSON line still comes out, carries the error, and the exit status is not 0."""
    env = dict(os.environ, FLX_FORCE_DIST="1", FLX_BENCH_TRIS="30000", FLX_BVH_CACHE=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0",
               FLX_RCCL_LIB=FAKE, FLX_ALLOW_RCCL_OVERRIDE="1", FAKE_RCCL_BREAK="4", FLX_NO_BUILD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--width", "320",
           "--height", "180", "--num-tasks", "65536", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(lines[-1])
    assert line.get("gather_native_error") and line["gather_matches_read_pixels"] is True
    assert r.returncode != 0
