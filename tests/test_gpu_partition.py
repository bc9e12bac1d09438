"""The multi-GPU device path on ONE GPU: pixel-interleaved partition (flx_set_partition), the local -> global pixel mapping of
k_raygen, the per-rank framebuffer, flx_copy_pixels_to_device, and the RCCL code path of bench.py (FLX_FORCE_DIST=1 under
torch.distributed.run with one rank).  The reference has no counterpart (one cl::CommandQueue, one device,
src/clcontext.cpp:25-29); the oracle mirrors the partition arithmetic (oracle/wf_oracle.cpp orc_wf_raygen), so the comparison is
the usual one: lockstep, bit-exact, through the C ABI."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
import common
from common import Q
from fluctus_amd import host, driver, multi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctxs(d, p, n, rank, world, env=None):
    from fluctus_amd.device import HipContext
    from oracle.binding import OracleContext
    g, o = HipContext(n), OracleContext(n, threads=8)
    g.set_option("extend_tree", 2)          # bit-exact comparisons: the reference's visit order
    for c in (g, o):
        c.upload_scene(d)
        if env is not None:
            c.upload_envmap(env)
        c.set_partition(rank, world)
        c.set_params(p)
        driver.reset_renderer(c)
    return g, o


def _compare(g, o, what):
    cg, co = g.get_counters(), o.get_counters()
    g.finish()
    assert (cg == co).all(), f"{what}: counters {cg} vs {co}"
    for q in range(8):
        n = int(co[q])
        assert np.array_equal(g.queue_read(q)[:n], o.queue_read(q)[:n]), f"{what}: queue {q} differs"
    fails = common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
    assert not fails, f"{what}: " + "; ".join(fails[:5])


@pytest.mark.parametrize("rank,world", [(0, 1), (1, 3), (7, 8), (2, 2 + 3)])
def test_partition_lockstep_vs_oracle(rank, world):
    """Every kernel of 6 iterations from the oracle's state, rank r of R: path state, queues, counters ==; the rank's
    framebuffer (local pixels) == in sample counts and within the float-atomic tolerance in the sums."""
    d = common.mixed_material_scene()
    w, h, n = 61, 47, 4096                   # w*h = 2867: not a multiple of 3, 5 or 8 -> ragged last stripe
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=1, useEnvMap=1, wfSeparateQueues=1)
    g, o = _ctxs(d, p, n, rank, world, env=host.synthetic_sky(64, 32))
    lp = multi.local_pixel_count(w * h, rank, world)
    assert g.local_pixels() == lp
    for it in range(6):
        for name, fn in (("logic", lambda c: c.wf_logic(False)), ("raygen", lambda c: c.wf_raygen()), ("materials", lambda c: c.wf_materials())):
            common.sync(g, o)
            fn(g); fn(o)
            _compare(g, o, f"rank {rank}/{world} it{it} {name}")
        cnt = o.get_counters().copy()
        for name, fn in (("extend", lambda c: c.wf_extend()), ("shadow", lambda c: c.wf_shadow())):
            common.sync(g, o)
            fn(g); fn(o)
            _compare(g, o, f"rank {rank}/{world} it{it} {name}")
        for c in (g, o):
            c.clear_queues()
            c.pixel_index_update(lp, int(cnt[Q.RAYGEN]))          # the cursor runs over the rank's LOCAL pixels
    st = g.state_export().view(np.uint32)
    assert st[common.COL.PIXEL_INDEX].max() < lp
    # lockstep re-synchronises the path state but not the framebuffers: both sides splatted the same paths into the same pixels
    pg, po = g.read_pixels(0), o.read_pixels(0)[:lp]
    assert pg.shape == (lp, 4)
    assert np.array_equal(pg[:, 3], po[:, 3]) and pg[:, 3].sum() > 0
    assert common.fb_close(pg, po)


@pytest.mark.parametrize("rank,world", [(0, 1), (1, 3), (7, 8)])
def test_partition_free_run_and_device_copy(rank, world):
    """24 free-running iterations on rank r of R, then the tile leaves the context the way the gather takes it:
    flx_copy_pixels_to_device into a caller-owned device buffer (a torch tensor) == flx_read_pixels."""
    import torch
    d = common.mixed_material_scene()
    w, h, n = 80, 50, 8192
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=1, useEnvMap=0, wfSeparateQueues=1)
    g, o = _ctxs(d, p, n, rank, world)
    lp = multi.local_pixel_count(w * h, rank, world)
    for it in range(24):
        cg = driver.benchmark_iteration(g, lp)
        co = driver.benchmark_iteration(o, lp)
        assert (cg == co).all(), f"iteration {it}: counters {cg} vs {co}"
    assert not common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
    pg, po = g.read_pixels(0), o.read_pixels(0)[:lp]
    assert common.fb_close(pg, po)
    maxlp = (w * h + world - 1) // world
    tile = torch.full((maxlp, 4), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()                   # the fill runs on torch's stream, the copy on the context's: order them
    g.copy_pixels_to_device(tile.data_ptr())
    g.finish()
    torch.cuda.synchronize()
    t = tile.cpu().numpy()
    assert np.array_equal(t[:lp], pg)
    assert (t[lp:] == -1.0).all()                                 # nothing written past the rank's local pixels


def test_microkernel_integrator_refuses_a_partition():
    """The microkernel integrator indexes the framebuffer by path id: with a partition the buffers only hold the local pixels,
    so every flx_mk_* call must fail cleanly instead of writing past them."""
    from fluctus_amd.device import HipContext
    d = common.simple_scene()
    p = common.scene_params(d, 32, 32, maxBounces=3)
    g = HipContext(1024)
    g.upload_scene(d); g.set_partition(1, 2); g.set_params(p)
    for call in (g.mk_reset, g.mk_raygen, g.mk_next_vertex, g.mk_sample_bsdf, g.mk_splat, g.mk_splat_preview):
        with pytest.raises(RuntimeError, match="single-GPU"):
            call()
    g.set_partition(0, 1)
    g.mk_reset(); g.mk_raygen(); g.finish()


def test_bench_rccl_path_single_rank(tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run ...), with one rank and
    FLX_FORCE_DIST=1: process-group init over RCCL, barrier, all-reduce of the timings, and the tile gather, whose result must
    equal flx_read_pixels (bench.py asserts it and reports gather_matches_read_pixels)."""
    env = dict(os.environ, FLX_FORCE_DIST="1", FLX_BENCH_TRIS="30000", FLX_BVH_CACHE=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3",
           "--num-tasks", "262144", "--width", "640", "--height", "360", "--no-cpu-baseline", "--scaling", "strong"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0
    # --scaling strong: the paths in flight are fixed over the whole job (num_tasks // world per rank; one rank here), and the line says so
    assert j["scaling"] == "strong" and j["config"]["num_tasks_per_gpu"] == 262144 and j["config"]["num_tasks_whole_job"] == 262144 and j["config"]["scaling_mode"].startswith("strong")
    assert "gather_ms" in j and j["gather_ms"] > 0
    assert j["gather_matches_read_pixels"] is True
    assert j["gather_native_matches_torch"] is True and j["gather_ms_native_rccl"] > 0     # flx_group_init + flx_gather beside torch's


@pytest.mark.parametrize("nranks,root", [(3, 0), (4, 2), (8, 7)])
def test_native_gather_local_group_on_one_device(nranks, root):
    """flx_group_init_local / flx_gather_local with N contexts on the ONE device of this box (tiles travel by device copies, RCCL
    refuses duplicate devices): every rank free-runs its partition, the gathered image must be exactly the interleave of the
    ranks' own framebuffers, and the whole must conserve the splat count."""
    from fluctus_amd import device
    d = common.mixed_material_scene()
    w, h, n = 75, 41, 4096                       # 3075 pixels: ragged for 4 and 8 ranks
    p = common.scene_params(d, w, h, maxBounces=4, useAreaLight=1, wfSeparateQueues=1)
    ctxs = []
    for r in range(nranks):
        g = device.HipContext(n)
        g.upload_scene(d); g.set_params(p)
        ctxs.append(g)
    device.group_init_local(ctxs)
    new = 0
    for r, g in enumerate(ctxs):
        assert g.local_pixels() == multi.local_pixel_count(w * h, r, nranks)
        driver.reset_renderer(g)
        for it in range(12):
            cnt = driver.benchmark_iteration(g, g.local_pixels())
            if it > 0:
                new += int(cnt[Q.RAYGEN])
    full = device.gather_local(ctxs, root)
    assert full.shape == (w * h, 4) and np.isfinite(full).all()
    for r, g in enumerate(ctxs):
        assert np.array_equal(full[r::nranks], g.read_pixels(0)), f"rank {r}'s tile is not at global pixels {r}, {r}+{nranks}, ..."
    assert int(full[:, 3].sum()) == new
    # a second gather (buffers reused) gives the same image
    assert np.array_equal(device.gather_local(ctxs, root), full)


def test_native_gather_over_rccl_single_rank():
    """The RCCL path proper on the one GPU there is: ncclGetUniqueId -> ncclCommInitRank(nranks = 1) -> flx_gather, and the
    single-process flavour ncclCommInitAll(1).  (N > 1 over xGMI is the driver's 8-GPU run: bench.py uses flx_gather there and
    checks it against torch.distributed's gather.)"""
    from fluctus_amd import device
    d = common.simple_scene()
    w, h, n = 40, 30, 2048
    p = common.scene_params(d, w, h, maxBounces=4)
    g = device.HipContext(n)
    g.upload_scene(d); g.set_params(p)
    g.group_init(0, 1, device.group_unique_id())
    driver.reset_renderer(g)
    for _ in range(8):
        driver.benchmark_iteration(g, w * h)
    full = g.gather(0)
    assert np.array_equal(full, g.read_pixels(0)) and full[:, 3].sum() > 0
    g2 = device.HipContext(n)
    g2.upload_scene(d); g2.set_params(p)
    device.group_init_local([g2])
    driver.reset_renderer(g2)
    for _ in range(8):
        driver.benchmark_iteration(g2, w * h)
    assert np.array_equal(device.gather_local([g2], 0), g2.read_pixels(0))
    # same scene, same seeds: the same render (sums up to the order of the float atomics of paths sharing a pixel)
    again = device.gather_local([g2], 0)
    assert common.fb_close(again, full)
