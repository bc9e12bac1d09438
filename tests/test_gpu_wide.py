"""The 4-wide quantised tree (csrc/flx_wide.h, flx_trace4.h, trace4.hip) against the oracle.

k_shadow4 (any-hit, the default of flx_wf_shadow): order-free query on conservative boxes with an exact leaf test -> demanded
BIT-IDENTICAL to the oracle's bvh_occluded restatement, like everything else (this file: lockstep on the small scenes incl. the
single-leaf / deep-chain edge cases; tests/test_gpu_parity.py runs its whole matrix with shadow_tree 4 and compares full-size runs
of the 4-wide and the binary kernel bit for bit).

k_extend4 (closest hit, the default of flx_wf_extend): same leaves, different visit order.  The closest triangle can differ from
the reference's only (a) in an exact tie of t between two triangles, (b) where box-vs-triangle rounding prunes a leaf under one
order and not the other.  SURVEY 8(c) budgets hit-index flips for <= 1e-5 of the rays; here every extension launch is compared
with the oracle ray by ray from the same input state, flips are COUNTED, the count is asserted against that budget, and where the
hit index agrees everything the kernel writes must agree bit for bit.
"""
import json
import os
import numpy as np
import pytest
import common
from common import COL, Q
from fluctus_amd import host, wire, driver

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLIP_BUDGET = 1e-5            # "a hit-index flip on a grazing ray is allowed for <= 1e-5 of rays and reported"
HIT_COLS = [COL.P, COL.P + 1, COL.P + 2, COL.N, COL.N + 1, COL.N + 2, COL.UV, COL.UV + 1, COL.HIT_T, COL.HIT_I, COL.AREA_LIGHT_HIT, COL.MAT_ID, COL.PATH_LEN]


def _report(name, payload):
    out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "r06_wide_flips.json")
    try:
        j = json.load(open(path))
    except Exception:
        j = {}
    j[name] = payload
    json.dump(j, open(path, "w"), indent=1)


def _extend_flips(g, o, what):
    """g and o hold the same state and queues; both trace their extension queue; returns (rays, flips) and checks that every
    ray whose hit index agrees carries identical hit records."""
    cnt = o.get_counters().copy()
    n = int(cnt[Q.EXTENSION])
    rays = o.queue_read(Q.EXTENSION)[:n]
    g.wf_extend(); o.wf_extend()
    g.finish()
    sg, so = g.state_export(), o.state_export()
    ig, io = sg.view(np.uint32), so.view(np.uint32)
    flip = ig[COL.HIT_I][rays] != io[COL.HIT_I][rays]
    same = rays[~flip]
    for c in HIT_COLS:
        a, b = ig[c][same], io[c][same]
        assert np.array_equal(a, b), f"{what}: column {common.colname(c)} differs on {int((a != b).sum())} rays whose hit index agrees"
    # a flipped ray still found A triangle at (nearly) the same distance: this is a tie, not a miss
    if flip.any():
        fr = rays[flip]
        tg, to = sg[COL.HIT_T][fr], so[COL.HIT_T][fr]
        assert (ig[COL.HIT_I][fr].view(np.int32) >= 0).all() and (io[COL.HIT_I][fr].view(np.int32) >= 0).all(), f"{what}: a flip between hit and miss"
        assert np.allclose(tg, to, rtol=1e-5, atol=1e-6), f"{what}: flipped rays differ in t by more than a rounding tie: {tg[:4]} vs {to[:4]}"
    # nothing but the rays of the queue was touched
    untouched = np.ones(sg.shape[1], bool); untouched[rays] = False
    for c in HIT_COLS:
        assert np.array_equal(ig[c][untouched], io[c][untouched])
    return n, int(flip.sum())


def _ctxs(d, p, n, env=None, ext=4, shadow=4):
    from fluctus_amd.device import HipContext
    from oracle.binding import OracleContext
    g, o = HipContext(n), OracleContext(n, threads=16)
    g.set_option("extend_tree", ext); g.set_option("shadow_tree", shadow)
    for c in (g, o):
        c.upload_scene(d)
        if env is not None:
            c.upload_envmap(env)
        c.set_params(p)
        driver.reset_renderer(c)
    return g, o


@pytest.mark.parametrize("scene", ["simple", "mixed"])
def test_wide_closest_hit_lockstep_small(scene):
    """Every iteration from the oracle's state: logic / raygen / materials / shadow (4-wide any-hit) bit-exact as always; the
    4-wide extension kernel compared ray by ray."""
    d = common.simple_scene() if scene == "simple" else common.mixed_material_scene()
    w, h, n = 64, 48, 4096
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=1, useEnvMap=int(scene == "mixed"), wfSeparateQueues=1)
    g, o = _ctxs(d, p, n, env=host.synthetic_sky(64, 32))
    info = g.scene_info()
    assert info["nested"] == 1 and info["wide_nodes"] >= 1
    rays = flips = 0
    for it in range(10):
        for fn in (lambda c: c.wf_logic(False), lambda c: c.wf_raygen(), lambda c: c.wf_materials()):
            common.sync(g, o)
            fn(g); fn(o)
        cnt = o.get_counters().copy()
        common.sync(g, o)
        r, f = _extend_flips(g, o, f"{scene} it{it}")
        rays += r; flips += f
        common.sync(g, o)
        g.wf_shadow(); o.wf_shadow(); g.finish()
        assert np.array_equal(g.state_export().view(np.uint32)[COL.SHADOW_BLOCKED], o.state_export().view(np.uint32)[COL.SHADOW_BLOCKED]), f"it{it}: shadowRayBlocked"
        for c in (g, o):
            c.clear_queues()
            c.pixel_index_update(w * h, int(cnt[Q.RAYGEN]))
    _report(f"lockstep_{scene}", {"rays": rays, "flips": flips})
    assert flips <= max(1, int(FLIP_BUDGET * rays)), (rays, flips)


def test_wide_tree_edge_cases_single_leaf_and_deep_chain():
    """(a) a scene that is ONE leaf (the wide root is a leaf reference), (b) a 40-level right-leaning chain (every wide node = three
    leaves + one inner child; the traversal stack outgrows its LDS levels and spills)."""
    d = common.small_mesh_scene(n=6)
    d.tris = d.tris[:2].copy()
    d.materials = np.array([common.default_material()], wire.MATERIAL)
    d.texdesc = np.zeros(0, wire.TEXDESC); d.texdata = np.zeros(0, np.uint8)
    host.build_bvh(d, "sbvh")
    assert d.nodes.size == 1
    p = common.scene_params(d, 32, 32, maxBounces=3)
    g, o = _ctxs(d, p, 1024)
    for it in range(6):
        cg = driver.benchmark_iteration(g, 32 * 32); co = driver.benchmark_iteration(o, 32 * 32)
        assert (cg == co).all()
    assert not common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
    # (b)
    d = common.small_mesh_scene(n=6)
    d.tris = d.tris[:61].copy()
    d.materials = np.array([common.default_material()], wire.MATERIAL)
    d.texdesc = np.zeros(0, wire.TEXDESC); d.texdata = np.zeros(0, np.uint8)
    nt = d.tris.size
    P = np.stack([np.stack([d.tris[v]["p"][k] for k in "xyz"], 1) for v in ("v0", "v1", "v2")], 1)
    tmin, tmax = P.min(1), P.max(1)
    nodes = np.zeros(2 * nt - 1, wire.NODE)
    sufmin = np.minimum.accumulate(tmin[::-1], 0)[::-1]; sufmax = np.maximum.accumulate(tmax[::-1], 0)[::-1]

    def setbox(k, mn, mx):
        for j, a in enumerate("xyz"):
            nodes[k]["bmin"][a] = mn[j]; nodes[k]["bmax"][a] = mx[j]
    idx = 0
    for t in range(nt - 1):
        setbox(idx, sufmin[t], sufmax[t]); nodes[idx]["parent"] = idx - 2 if t else -1; nodes[idx]["nPrims"] = 0; nodes[idx]["iStartOrRight"] = idx + 2
        setbox(idx + 1, tmin[t], tmax[t]); nodes[idx + 1]["parent"] = idx; nodes[idx + 1]["nPrims"] = 1; nodes[idx + 1]["iStartOrRight"] = t
        idx += 2
    setbox(idx, tmin[nt - 1], tmax[nt - 1]); nodes[idx]["parent"] = idx - 2; nodes[idx]["nPrims"] = 1; nodes[idx]["iStartOrRight"] = nt - 1
    d.nodes, d.indices = nodes, np.arange(nt, dtype=np.uint32)
    d.world_radius = float(0.5 * np.linalg.norm(sufmax[0] - sufmin[0]))
    p = common.scene_params(d, 32, 32, maxBounces=3)
    # the stack pages through the spill area in the persistent closest-hit kernel (default), in the thread-per-ray any-hit kernel (default)
    # and, second pass, in the persistent any-hit kernel and the thread-per-ray closest-hit kernel
    for refill_ext, refill_sh in ((None, None), (0, 16 | (32 << 8))):
        g, o = _ctxs(d, p, 1024)
        if refill_ext is not None:
            g.set_option("refill_extend", refill_ext); g.set_option("refill_shadow", refill_sh)
        info = g.scene_info()
        assert info["binary_depth"] == nt - 1 and info["wide_stack_bound"] > 16 and info["spill_levels"] >= info["binary_depth"] + 1 - 16
        rays = flips = 0
        for it in range(6):
            for fn in (lambda c: c.wf_logic(False), lambda c: c.wf_raygen(), lambda c: c.wf_materials()):
                common.sync(g, o); fn(g); fn(o)
            cnt = o.get_counters().copy()
            common.sync(g, o)
            r, f = _extend_flips(g, o, f"chain it{it}"); rays += r; flips += f
            common.sync(g, o)
            g.wf_shadow(); o.wf_shadow(); g.finish()
            assert np.array_equal(g.state_export().view(np.uint32)[COL.SHADOW_BLOCKED], o.state_export().view(np.uint32)[COL.SHADOW_BLOCKED])
            for c in (g, o):
                c.clear_queues(); c.pixel_index_update(32 * 32, int(cnt[Q.RAYGEN]))
        assert flips == 0, (rays, flips)
        g.close()


@pytest.mark.parametrize("workload", ["kitchen", "conference"])
def test_wide_closest_hit_flip_count_full_size_vs_oracle(workload):
    """BASELINE configs 1 / 2 at full size, 1 M paths: the device free-runs the DEFAULT configuration (both kernels on the 4-wide
    tree); at four points of the run the state goes to the oracle, both trace the same 1 M extension rays (primary + bounced mix)
    and the same shadow rays: hit-index flips counted against the 1e-5 budget, shadowRayBlocked demanded identical."""
    import bench
    d, p, env = bench.build_workload(name=workload)
    n, npix = 1 << 20, int(p["width"]) * int(p["height"])
    g, o = _ctxs(d, p, n, env=env)
    rays = flips = shadow_rays = 0
    checkpoints = (0, 3, 9, 14)
    for it in range(15):
        g.wf_logic(False); g.wf_raygen(); g.wf_materials()
        cnt = g.get_counters(); g.finish(); cnt = cnt.copy()
        if it in checkpoints:
            common.sync(o, g)
            r, f = _extend_flips(g, o, f"{workload} it{it}")
            rays += r; flips += f
            # continue the device from the ORACLE's hit records so that a flip does not fork the two runs
            g.state_import(o.state_export())
            g.wf_shadow(); o.wf_shadow(); g.finish()
            assert np.array_equal(g.state_export().view(np.uint32)[COL.SHADOW_BLOCKED], o.state_export().view(np.uint32)[COL.SHADOW_BLOCKED]), f"{workload} it{it}: shadowRayBlocked"
            shadow_rays += int(cnt[Q.SHADOW])
        else:
            g.wf_extend(); g.wf_shadow()
        g.clear_queues(); g.finish()
        g.pixel_index_update(npix, int(cnt[Q.RAYGEN]))
    _report(f"full_size_{workload}", {"extension_rays_compared": rays, "hit_index_flips": flips, "flip_rate": flips / max(1, rays),
                                      "shadow_rays_compared_bit_exact": shadow_rays})
    assert rays >= 4 * n
    assert flips <= FLIP_BUDGET * rays, (rays, flips)


@pytest.mark.parametrize("workload", ["kitchen", "conference", "courtyard-1440p"])
def test_wide_vs_binary_kernels_on_device_full_size(workload):
    """4 M rays per launch, 12 iterations, device vs device: two contexts free-run the same workload, one with both kernels on the
    4-wide tree, one on the reference's binary tree.  Iteration by iteration the states are re-synchronised, so every launch
    compares the two closest-hit kernels on identical rays (flip count) and the two any-hit kernels bit for bit."""
    from fluctus_amd.device import HipContext
    import bench
    d, p, env = bench.build_workload(name=workload)
    n, npix = 1 << 22, int(p["width"]) * int(p["height"])
    ctx = []
    for tree in (4, 2):
        g = HipContext(n)
        g.set_option("extend_tree", tree); g.set_option("shadow_tree", tree)
        g.upload_scene(d); g.upload_envmap(env); g.set_params(p); driver.reset_renderer(g)
        ctx.append(g)
    gw, gb = ctx
    rays = flips = sh = 0
    for it in range(12):
        for g in ctx:
            g.wf_logic(False); g.wf_raygen(); g.wf_materials()
        cw, cb = gw.get_counters(), gb.get_counters(); gw.finish(); gb.finish()
        assert (cw == cb).all()
        for g in ctx:
            g.wf_extend(); g.wf_shadow(); g.finish()
        sw, sb = gw.state_export().view(np.uint32), gb.state_export().view(np.uint32)
        q = gb.queue_read(Q.EXTENSION)[:int(cb[Q.EXTENSION])]
        flip = sw[COL.HIT_I][q] != sb[COL.HIT_I][q]
        rays += q.size; flips += int(flip.sum())
        same = q[~flip]
        for c in HIT_COLS:
            assert np.array_equal(sw[c][same], sb[c][same]), (it, common.colname(c))
        assert np.array_equal(sw[COL.SHADOW_BLOCKED], sb[COL.SHADOW_BLOCKED]), f"it{it}: any-hit kernels disagree"
        sh += int(cb[Q.SHADOW])
        if flip.any():
            gw.state_import(gb.state_export())          # keep the two runs on the same paths
        for g in ctx:
            g.clear_queues(); g.finish(); g.pixel_index_update(npix, int(cb[Q.RAYGEN]))
    _report(f"wide_vs_binary_{workload}", {"extension_rays_compared": rays, "hit_index_flips": flips, "flip_rate": flips / max(1, rays),
                                           "shadow_rays_compared_bit_exact": sh, "scene_info": gw.scene_info()})
    assert flips <= FLIP_BUDGET * rays, (rays, flips)


def _free_run_default_vs_oracle(workload, n, iterations, start_iterations=0):
    """The shipped DEFAULT configuration (k_extend4 + k_shadow4, fused logic + material pass, two streams) free-running beside the
    oracle.  Both sides run whole iterations on their own; after every extension launch the hit records are compared ray by ray and,
    should a tie have flipped a hit index, the device continues from the ORACLE's records (tests/test_gpu_wide.py's resync pattern),
    so one flip cannot fork the two runs and everything else stays comparable bit for bit: queue counters after every logic /
    raygen / material step, the extension queue as a set, the whole path state after every extension AND every shadow launch,
    the framebuffer at the end.  Returns (rays, flips)."""
    import bench
    d, p, env = bench.build_workload(name=workload)
    npix = int(p["width"]) * int(p["height"])
    g, o = _ctxs(d, p, n, env=env)
    # the shipped defaults: both traversals on the 4-wide tree, persistent-wave extension kernel with RAW hit records, fused logic pass,
    # the stream schedule flx_upload_scene picked for the scene
    assert g.get_option("extend_tree") == 4 and g.get_option("shadow_tree") == 4 and g.get_option("fuse") == 1
    assert g.get_option("refill_extend") > 0 and g.get_option("overlap") == 2 and g.get_option("refill_shadow") == 0
    for it in range(start_iterations):                  # device alone (cheap), then the oracle takes the state over ...
        cnt = driver.benchmark_iteration(g, npix)
        o.pixel_index_update(npix, int(cnt[Q.RAYGEN]))   # ... the pixel cursor included (replayed: it is host-side state of both)
    if start_iterations:
        common.sync(o, g)
    rays = flips = 0
    for it in range(iterations):
        for c in (g, o):
            c.wf_logic(False); c.wf_raygen(); c.wf_materials()
        cg, co = g.get_counters(), o.get_counters(); g.finish()
        cg, co = np.array(cg, copy=True), np.array(co, copy=True)
        assert (cg == co).all(), f"{workload} it{it}: counters {cg} vs {co}"
        ne = int(co[Q.EXTENSION])
        qo = o.queue_read(Q.EXTENSION)[:ne]
        assert np.array_equal(np.sort(g.queue_read(Q.EXTENSION)[:ne]), np.sort(qo)), f"{workload} it{it}: extension queues hold different paths"
        g.wf_extend(); o.wf_extend(); g.finish()
        sg, so = g.state_export(), o.state_export()
        flip = np.zeros(sg.shape[1], bool)
        flip[qo] = sg.view(np.uint32)[COL.HIT_I][qo] != so.view(np.uint32)[COL.HIT_I][qo]
        rays += ne; flips += int(flip.sum())
        fails = common.state_diff(sg, so, 0.0, 0.0, mask=~flip)
        assert not fails, f"{workload} it{it} after extend: " + "; ".join(fails[:4])
        if flip.any():
            fr = np.nonzero(flip)[0]
            assert np.allclose(sg[COL.HIT_T][fr], so[COL.HIT_T][fr], rtol=1e-5, atol=1e-6), f"{workload} it{it}: a flip that is not a tie in t"
            g.state_import(so)
        g.wf_shadow(); o.wf_shadow(); g.finish()
        fails = common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
        assert not fails, f"{workload} it{it} after shadow: " + "; ".join(fails[:4])
        for c in (g, o):
            c.clear_queues(); c.pixel_index_update(npix, int(co[Q.RAYGEN]))
    if not start_iterations:
        assert common.fb_close(g.read_pixels(0), o.read_pixels(0)), f"{workload}: framebuffers differ"
    g.close()
    return rays, flips


@pytest.mark.parametrize("workload", ["kitchen", "conference", "courtyard-1440p"])
def test_default_path_free_run_vs_oracle_full_size(workload):
    """BASELINE configs 1-3 at their full resolutions and bounce counts, 1 M paths, 10 whole iterations of the default path against the
    ORACLE (not against the binary device kernel): state / counters / queues identical throughout, hit-index flips counted."""
    n = 1 << 20
    rays, flips = _free_run_default_vs_oracle(workload, n, 10)
    _report(f"default_free_run_{workload}", {"paths": n, "iterations": 10, "extension_rays_compared_vs_oracle": rays, "hit_index_flips": flips,
                                             "flip_rate": flips / max(1, rays)})
    assert rays == 10 * n
    assert flips <= FLIP_BUDGET * rays, (rays, flips)


def test_default_path_2160p_checkpoint_vs_oracle():
    """BASELINE config 4's geometry (courtyard-proc, 3840x2160, 16 bounces): the device runs 20 iterations alone (deep paths in flight),
    then two iterations in lockstep with the oracle as above."""
    n = 1 << 20
    rays, flips = _free_run_default_vs_oracle("courtyard-2160p", n, 2, start_iterations=20)
    _report("default_2160p_checkpoint", {"paths": n, "extension_rays_compared_vs_oracle": rays, "hit_index_flips": flips})
    assert rays == 2 * n and flips <= FLIP_BUDGET * rays, (rays, flips)


def test_default_path_bench_path_count_checkpoint_vs_oracle():
    """The bench's OWN path count: every other device-vs-ORACLE run is 1 M paths, bench.py times bench.NUM_TASKS = 16 M since round 5 (other grid sizes, cursor
    traffic, persistent-grid : block ratio, a second level of the scan).  kitchen at n = bench.NUM_TASKS: the device runs 20 iterations alone
    (stationary queue mix, deep paths in flight), then two whole iterations in lockstep with the oracle, ray by ray as above."""
    import bench
    n = bench.NUM_TASKS
    rays, flips = _free_run_default_vs_oracle("kitchen", n, 2, start_iterations=20)
    _report("default_bench_path_count_checkpoint_kitchen", {"paths": n, "extension_rays_compared_vs_oracle": rays, "hit_index_flips": flips})
    assert rays == 2 * n and flips <= FLIP_BUDGET * rays, (rays, flips)


def test_egyptcat_reference_protocol_free_run_vs_oracle():
    """A REAL reference asset under the reference's own benchmark protocol: assets/egyptcat/egyptcat.obj (scene #1 of Tracer::runBenchmark,
    src/tracer.cpp:384-389; here from tests/golden/egyptcat_scene.npz) at 1024 x 1024 (:365-366) with the start-up parameters (default
    camera and area light, 10 bounces, single material queue) and the default wfBufferSize of 2^20 paths: 10 whole iterations of the
    shipped default path beside the ORACLE, the same assertions as test_default_path_free_run_vs_oracle_full_size."""
    n = 1 << 20
    rays, flips = _free_run_default_vs_oracle("egyptcat", n, 10)
    _report("default_free_run_egyptcat", {"paths": n, "iterations": 10, "extension_rays_compared_vs_oracle": rays, "hit_index_flips": flips,
                                          "flip_rate": flips / max(1, rays)})
    assert rays > 5 * n                              # (rays that leave the scene terminate: fewer than n per iteration after the first)
    assert flips <= FLIP_BUDGET * rays, (rays, flips)


@pytest.mark.parametrize("workload", ["kitchen", "conference", "courtyard-1440p"])
def test_bench_launch_chain_vs_oracle_full_size(workload):
    """The launch chain bench.py times, untouched: logic -> genRays -> materials -> extension -> shadow -> clear, nothing looking at the
    state in between, so that the persistent extension kernel's RAW hit records are committed by the next fused logic pass
    (k_logic<FUSE, RAW>) and never by k_materialise -- the free runs above export the state after every extension launch and therefore
    take the other route.  Device and ORACLE free-run BASELINE's configuration at 1 M paths: queue counters after every iteration,
    the whole path state every fifth iteration (that export commits the pending records in memory; the four
    iterations before it went through the RAW pass), the framebuffer at the end.  A tie in t resolved the other way (SURVEY 8(c)'s
    hit-index flips, counted ray by ray and shown to be ties by the tests above) cannot be resynchronised inside a stretch here; it
    sends ONE path another way -- paths are independent; queue ORDER and hence the pixel a later path is given are not, which is why the
    device is put back on the oracle's state as soon as a counter differs.  At a checkpoint (every fifth iteration, or the iteration a
    counter differs) the paths whose state differs are counted against the same 1e-5 budget and everything else must be bit-identical.  (Zero tolerance for the same chain: test_refill_kernels_are_bit_identical_to_thread_per_ray, device vs device.)
    FLX_SOAK_ITERS lengthens the run (default 15)."""
    _bench_launch_chain_vs_oracle(workload, 1 << 20, int(os.environ.get("FLX_SOAK_ITERS", "15")))


def test_bench_launch_chain_vs_oracle_at_bench_path_count():
    """The same chain at bench.NUM_TASKS (16 M paths since round 5: the path count bench.py times -- two-level queue scan, other grids, the persistent grid : block
    ratio of the timed run) on the headline workload: the device runs 10 iterations alone (deep paths in flight, stationary queue mix), the oracle
    takes over its state, then 5 whole iterations of the untouched launch chain on both sides -- k_logic<FUSE, RAW> commits the RAW hit records of
    all of them with nothing looking in between -- counters every iteration, the whole state at the end (round 4's verdict: "k_logic<1, true> at 8 M [the count then]
    has never been compared with anything but itself")."""
    import bench
    _bench_launch_chain_vs_oracle("kitchen", bench.NUM_TASKS, 5, start_iterations=10, tag="bench_path_count_")


def _bench_launch_chain_vs_oracle(workload, n, iters, start_iterations=0, tag=""):
    import bench
    d, p, env = bench.build_workload(name=workload)
    npix = int(p["width"]) * int(p["height"])
    g, o = _ctxs(d, p, n, env=env)
    assert g.get_option("refill_extend") > 0 and g.get_option("fuse") == 1 and g.get_option("extend_tree") == 4
    rays = ext_rays = forked = 0
    cursor0 = 0
    for _ in range(start_iterations):                     # the device alone: then the oracle takes over its state, queues (empty) and pixel cursor
        c0 = driver.benchmark_iteration(g, npix)
        cursor0 = (cursor0 + int(c0[Q.RAYGEN])) % npix
    if start_iterations:
        o.state_import(g.state_export())
        o.pixel_index_reset(); o.pixel_index_update(npix, cursor0)
        for c in (g, o):
            c.clear_queues()
        # the framebuffers differ from here on (the oracle's starts empty): compared as differences below
    skip = np.zeros(64, bool); skip[list(common.PAD_COLS)] = True; skip[COL.PHASE] = True
    cursor = cursor0                                      # the host-side pixel cursor both contexts carry (replayed from the oracle's counts)
    fb0 = g.read_pixels(0) if start_iterations else None
    common_state = [None, 0, 0]                           # oracle state, cursor, iteration count at the last point both sides were identical
    explained = []
    ray_counts = {}                                       # iteration -> the oracle's raygen-queue length of the main run (what both pixel cursors were advanced by)
    replay2 = [0, 0]                                      # times the oracle had to follow the device (explain_forks, step 2) | paths it reproduced that way
    shifted_total, last_shifted = [0], [0]                # paths whose pixel moved behind a tie that changed its path's termination (explain_forks)

    def explain_forks(bad_paths, upto, what, sg_main, so_main):
        """A path whose state differs at a checkpoint must have been forked by a TIE.  Two replays of the stretch since the last common state, both
        looking after every extension launch (the pattern of _free_run_default_vs_oracle):
          1. the DEVICE follows the oracle -- every path whose hit index differs at a launch must agree in `t` to 1e-5 there (a tie), everything else
             must be bit-identical at every launch; anything else (a wrong commit in the fused RAW pass, a wrong traversal result) fails here;
          2. (round 6; only if paths differ that replay 1 did not show tied) the ORACLE follows the device -- it adopts the device's hit records at
             every tied launch and otherwise runs on its own -- and must arrive, bit for bit on EVERY path, at the state the device's undisturbed main
             run reached (sg_main).  A tie can change WHETHER its path terminates in that iteration: the device's raygen queue then holds one path more
             or fewer and every path regenerated behind it gets the neighbouring pixel (src/wf_raygen.cl:25: pixel = cursor + queue index) -- a
             different camera ray, a different path from there on, for up to a fifth of the paths.  Round 5 accepted such paths by their pixel delta
             alone; now the oracle itself, given nothing but the device's tie choices, has to reproduce every one of them.
        The device's traversal is deterministic per ray, so the replays reproduce the main run's flips."""
        s0, cur0, it0 = common_state
        assert s0 is not None
        s1 = o.state_export()                             # where the main loop continues afterwards

        def replay(oracle_follows_device):
            for c in (g, o):
                c.state_import(s0); c.pixel_index_reset(); c.pixel_index_update(npix, cur0)
            tied_ = set()
            cur_ = cur0
            tagr = "replay 2 (oracle follows the device)" if oracle_follows_device else "replay"
            for j in range(it0, upto):
                for c in (g, o):
                    c.wf_logic(False); c.wf_raygen(); c.wf_materials()
                cgj, coj = g.get_counters(), o.get_counters(); g.finish()
                cgj, coj = np.array(cgj, copy=True), np.array(coj, copy=True)
                assert (cgj == coj).all(), f"{workload} {what}: {tagr} it{j}: counters {cgj} vs {coj}"
                qo = o.queue_read(Q.EXTENSION)[:int(coj[Q.EXTENSION])]
                g.wf_extend(); o.wf_extend(); g.finish()
                sg, so = g.state_export(), o.state_export()
                flip = np.zeros(sg.shape[1], bool)
                flip[qo] = sg.view(np.uint32)[COL.HIT_I][qo] != so.view(np.uint32)[COL.HIT_I][qo]
                fails = common.state_diff(sg, so, 0.0, 0.0, mask=~flip)
                assert not fails, f"{workload} {what}: {tagr} it{j} after extend, beyond hit-index flips: " + "; ".join(fails[:4])
                if flip.any():
                    fr = np.nonzero(flip)[0]
                    assert np.allclose(sg[COL.HIT_T][fr], so[COL.HIT_T][fr], rtol=1e-5, atol=1e-6), f"{workload} {what}: {tagr} it{j}: a flip that is not a tie in t"
                    tied_.update(int(x) for x in fr)
                    if oracle_follows_device:
                        o.state_import(sg)
                    else:
                        g.state_import(so)
                g.wf_shadow(); o.wf_shadow()
                # (the main loop advanced BOTH cursors by the oracle's count of that iteration: ray_counts)
                for c in (g, o):
                    c.clear_queues(); c.finish(); c.pixel_index_update(npix, ray_counts[j])
                cur_ = (cur_ + ray_counts[j]) % npix
            return tied_, cur_

        tied, cur = replay(False)
        # the replay ends where the oracle stood (it is deterministic)
        fails = common.state_diff(o.state_export(), s1, 0.0, 0.0)
        assert not fails, f"{workload} {what}: the oracle's replay does not reproduce its own run: " + "; ".join(fails[:3])
        assert cur == cursor
        rest = set(int(x) for x in bad_paths) - tied
        shifted = set()
        if rest:
            assert tied, f"{workload} {what}: paths {sorted(rest)[:8]} differ from the oracle and the replay shows no hit-index tie at all"
            tied2, cur2 = replay(True)
            assert cur2 == cursor
            so2 = o.state_export()
            still = ((so2.view(np.uint32) != sg_main.view(np.uint32)) & ~skip[:, None]).any(axis=0)
            if still.any():                               # diagnostics: which columns the device-following oracle and the device's main run disagree in
                for x in np.nonzero(still)[0][:4]:
                    cols = [c for c in range(64) if not skip[c] and sg_main.view(np.uint32)[c][x] != so2.view(np.uint32)[c][x]]
                    print(f"[fork] path {x}: device main run vs oracle following the device's ties differ in " + ", ".join(f"{common.colname(c)}: {sg_main[c][x]!r} vs {so2[c][x]!r}" for c in cols))
            assert not still.any(), (f"{workload} {what}: {int(still.sum())} paths of the device's main run are NOT what the oracle computes from the device's own tie choices "
                                     f"(first: {np.nonzero(still)[0][:8]}; ties: {sorted(tied)[:8]}, in replay 2: {sorted(tied2)[:8]})")
            replay2[0] += 1; replay2[1] += len(rest)
            shifted = rest - tied2                        # regenerated onto the neighbouring pixel behind a tie (or forked by a tie of their own on that new path: tied2)
            tied |= (tied2 & rest)
            for c in (g, o):                              # back to where the main loop continues: the oracle's own run
                c.state_import(s1); c.pixel_index_reset(); c.pixel_index_update(npix, cursor)
        shifted_total[0] += len(shifted)
        explained.extend(sorted(tied))
        last_shifted[0] = len(shifted)

    def checkpoint(what, upto):
        nonlocal forked
        sg, so = g.state_export(), o.state_export()
        bad = ((sg.view(np.uint32) != so.view(np.uint32)) & ~skip[:, None]).any(axis=0)
        fails = common.state_diff(sg, so, 0.0, 0.0, mask=~bad)
        assert not fails, f"{workload} {what}: " + "; ".join(fails[:4])
        if bad.any():
            explain_forks(np.nonzero(bad)[0], upto, what, sg, so)
            forked += int(bad.sum()) - last_shifted[0]
            g.state_import(o.state_export())
        common_state[0], common_state[1], common_state[2] = o.state_export(), cursor, upto

    checkpoint("start", 0)

    for it in range(iters):
        cnt = []
        for c in (g, o):                                   # fluctus_amd/driver.py: benchmark_iteration, with the ORACLE's count for both cursors
            c.wf_logic(False); c.wf_raygen(); c.wf_materials()
            cc = c.get_counters()
            c.wf_extend(); c.wf_shadow(); c.clear_queues(); c.finish()
            cnt.append(np.array(cc, copy=True))
        cg, co = cnt
        for c in (g, o):
            c.pixel_index_update(npix, int(co[Q.RAYGEN]))
        ray_counts[it] = int(co[Q.RAYGEN])
        cursor = (cursor + int(co[Q.RAYGEN])) % npix
        rays += int(co[Q.EXTENSION]) + int(co[Q.SHADOW]); ext_rays += int(co[Q.EXTENSION])
        if not (cg == co).all():
            # a forked path entered another material queue or ended at another bounce: at most a handful of paths, and the states must say so
            assert int(np.abs(cg.astype(np.int64) - co.astype(np.int64)).sum()) <= 8, f"{workload} it{it}: counters {cg} vs {co}"
            last_shifted[0] = 0
            before = forked
            checkpoint(f"it{it} (counters {cg} vs {co})", it + 1)
            assert forked > before, f"{workload} it{it}: counters differ but the states do not"
        elif it % 5 == 4 or it == iters - 1:
            checkpoint(f"after {it + 1} iterations", it + 1)
    assert forked <= max(1, int(FLIP_BUDGET * ext_rays)), (workload, forked, ext_rays)
    if not forked:
        pg, po = g.read_pixels(0), o.read_pixels(0)
        if fb0 is not None:                               # what the compared iterations ADDED (counts exact; sums to the any-order bound on the totals)
            assert np.array_equal(pg[:, 3] - fb0[:, 3], po[:, 3]), f"{workload}: splat counts differ"
            # (every fp32 add onto an accumulated value A rounds by <= 6e-8 A: a few dozen splats per pixel -> 4e-6 of the totals)
            err = np.abs((pg[:, :3] - fb0[:, :3]) - po[:, :3])
            assert (err <= 1e-5 * np.abs(po[:, :3]) + 4e-6 * (np.abs(fb0[:, :3]) + np.abs(pg[:, :3])) + 1e-6).all(), f"{workload}: framebuffers differ (max {err.max()})"
        else:
            assert common.fb_close(pg, po), f"{workload}: framebuffers differ"
    _report(f"bench_chain_vs_oracle_{tag}{workload}", {"paths": n, "iterations": iters, "rays": rays, "extension_rays": ext_rays,
                                                  "paths_forked_by_a_tie": forked, "forks_shown_to_be_ties_by_replay": len(explained),
                                                  "paths_regenerated_onto_the_neighbouring_pixel_behind_such_a_tie": shifted_total[0],
                                                  "stretches_replayed_with_the_oracle_following_the_device": replay2[0],
                                                  "paths_the_oracle_reproduced_bit_for_bit_from_the_device_tie_choices": replay2[1]})
    g.close()


# refill option = refillMin | waitMax << 8 (trace4r.hip)
@pytest.mark.parametrize("workload,refill", [("kitchen", 16 | (32 << 8)), ("conference", 8 | (16 << 8)), ("kitchen", 48 | (8 << 8)), ("kitchen", 16),
                                             ("courtyard-1440p", 16 | (24 << 8))])
def test_refill_kernels_are_bit_identical_to_thread_per_ray(workload, refill):
    """Persistent waves with lane refill (trace4r.hip) run the SAME per-ray code as k_extend4 / k_shadow4, whatever lane or wave a ray
    lands on and whenever it is handed out, and the commit of traceExtension -- done by the next fused logic pass from the RAW hit records
    (logic.hip: k_logic<FUSE, RAW>), or by k_materialise when the state is read first -- is the same arithmetic as commit_hit's: two contexts
    free-run the workload at 1 M paths, one with the thread-per-ray kernels, one with the refill kernels; counters after every iteration,
    the path state after iterations 3 and 12 (the first export commits raw records in memory, the run continues from there) and the
    framebuffers within the atomic-order bound."""
    from fluctus_amd.device import HipContext
    import bench
    d, p, env = bench.build_workload(name=workload)
    n, npix = 1 << 20, int(p["width"]) * int(p["height"])
    ctx = []
    for r in (0, refill):
        g = HipContext(n)
        g.set_option("refill_extend", r); g.set_option("refill_shadow", r)          # (0 = the thread-per-ray kernels: the defaults are not 0)
        g.upload_scene(d); g.upload_envmap(env); g.set_params(p); driver.reset_renderer(g)
        ctx.append(g)
    a, b = ctx
    assert b.get_option("refill_extend") == refill and a.get_option("refill_extend") == 0
    for it in range(12):
        ca, cb = driver.benchmark_iteration(a, npix), driver.benchmark_iteration(b, npix)
        assert (ca == cb).all(), f"{workload} it{it}: {ca} vs {cb}"
        if it == 2:
            fails = common.state_diff(a.state_export(), b.state_export(), 0.0, 0.0)
            assert not fails, "after 3 iterations: " + "; ".join(fails[:5])
    fails = common.state_diff(a.state_export(), b.state_export(), 0.0, 0.0)
    assert not fails, "; ".join(fails[:5])
    assert common.fb_close(a.read_pixels(0), b.read_pixels(0))
    for g in ctx:
        g.close()


@pytest.mark.parametrize("workload", ["kitchen", "conference"])
def test_in_kernel_regeneration_is_bit_identical_to_genrays(workload):
    """Option `regen` (logic.hip: REGEN): the fused RAW pass regenerates its terminating paths itself -- their index in the raygen queue (= pixel,
    src/wf_raygen.cl:25) comes from a decoupled look-back over the waves instead of from the queue scan -- and the genRays launch of the chain does nothing.
    Two contexts free-run the workload at 1 M paths (16 384 waves: the look-back walks windows of 64 predecessors, which the 65-wave fuzz scenes never
    make it do), one with genRays, one with the in-kernel regeneration; diffuse-inline pass with the merged extension queue on the kitchen, all-types
    pass with genRays' own block on the conference scene: counters after every iteration, the whole state after 3 and 12, the framebuffers."""
    from fluctus_amd.device import HipContext
    import bench
    d, p, env = bench.build_workload(name=workload)
    n, npix = 1 << 20, int(p["width"]) * int(p["height"])
    ctx = []
    for r in (0, 1):
        g = HipContext(n)
        g.upload_scene(d); g.upload_envmap(env); g.set_params(p); g.set_option("regen", r); driver.reset_renderer(g)
        ctx.append(g)
    a, b = ctx
    assert a.get_option("regen") == 0 and b.get_option("regen") == 1 and a.get_option("ext_order") == b.get_option("ext_order")
    for it in range(12):
        ca, cb = driver.benchmark_iteration(a, npix), driver.benchmark_iteration(b, npix)
        assert (ca == cb).all(), f"{workload} it{it}: {ca} vs {cb}"
        if it == 2:
            fails = common.state_diff(a.state_export(), b.state_export(), 0.0, 0.0)
            assert not fails, "after 3 iterations: " + "; ".join(fails[:5])
    fails = common.state_diff(a.state_export(), b.state_export(), 0.0, 0.0)
    assert not fails, "; ".join(fails[:5])
    pa, pb = a.read_pixels(0), b.read_pixels(0)
    assert np.array_equal(pa[:, 3], pb[:, 3]) and common.fb_close(pa, pb)
    for g in ctx:
        g.close()


@pytest.mark.parametrize("workload", ["kitchen", "conference", "egyptcat"])
def test_prepared_regeneration_is_bit_identical_to_genrays(workload):
    """Option `regen_prep` (default on; logic.hip: PREPARED REGENERATION): the fused RAW pass computes the seed-only half of genRays for its terminating lanes
    (jitter, thin-lens origin, throughput + new seed, the reset scalars: stored with its full-line stores) and the k_raygen of the chain only the direction and
    the pixel -- against the whole of genRays in k_raygen.  Two contexts free-run the workload at 1 M paths: counters after every iteration, the whole state after
    3 and 12 iterations (the first export happens right behind a chain that used the split), the framebuffers."""
    from fluctus_amd.device import HipContext
    import bench
    d, p, env = bench.build_workload(name=workload)
    n, npix = 1 << 20, int(p["width"]) * int(p["height"])
    ctx = []
    for r in (0, 1):
        g = HipContext(n)
        g.upload_scene(d); g.upload_envmap(env); g.set_params(p); g.set_option("regen_prep", r); driver.reset_renderer(g)
        ctx.append(g)
    a, b = ctx
    assert a.get_option("regen_prep") == 0 and b.get_option("regen_prep") == 1
    for it in range(12):
        ca, cb = driver.benchmark_iteration(a, npix), driver.benchmark_iteration(b, npix)
        assert (ca == cb).all(), f"{workload} it{it}: {ca} vs {cb}"
        if it in (2, 5):
            fails = common.state_diff(a.state_export(), b.state_export(), 0.0, 0.0)
            assert not fails, f"after {it + 1} iterations: " + "; ".join(fails[:5])
    # ... and with the state looked at right behind the chain (logic -> genRays -> materials), before the extension kernel: the regenerated paths as genRays leaves them
    for g in (a, b):
        g.wf_logic(False); g.wf_raygen(); g.wf_materials()
    fails = common.state_diff(a.state_export(), b.state_export(), 0.0, 0.0)
    assert not fails, "behind the chain: " + "; ".join(fails[:5])
    pa, pb = a.read_pixels(0), b.read_pixels(0)
    assert np.array_equal(pa[:, 3], pb[:, 3]) and common.fb_close(pa, pb)
    for g in ctx:
        g.close()


def test_raw_hit_records_call_patterns():
    """The persistent-wave extension kernel leaves RAW hit records; the host side of the boundary (api.hip: rawHits / KEEP_RAW / materialise)
    has to commit them the moment anything but the reference's own loop could observe a hit record -- and only then.  Call sequences that
    look in between, change options, repeat calls or leave the logic -> genRays -> materials chain are run on the device (shipped defaults)
    and on the oracle from the same state; the WHOLE exported state, queues and counters must agree after each, and `k_materialise` /
    the RAW variant of the fused pass must have run exactly when the pattern says so (kernel launch counts from the profile hooks are not
    available for them, so the distinction is made through the states: a pattern that skipped a needed commit would export raw records).
    The scene is the one whose 4-wide closest hit has no tie against the oracle (test_wide_closest_hit_lockstep_small: 0 flips)."""
    d = common.mixed_material_scene()
    w, h, n = 64, 48, 4096 + 37
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=1, useEnvMap=1, wfSeparateQueues=1)
    g, o = _ctxs(d, p, n, env=host.synthetic_sky(64, 32))
    assert g.get_option("refill_extend") > 0
    npix = w * h

    def cmp(what):
        cg, co = g.get_counters(), o.get_counters(); g.finish()
        assert (np.array(cg) == np.array(co)).all(), f"{what}: counters {cg} vs {co}"
        fails = common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
        assert not fails, f"{what}: " + "; ".join(fails[:4])
        # no raw marker may ever be visible in an exported hit index (bits 31:30 == 01)
        hi = g.state_export().view(np.uint32)[COL.HIT_I]
        assert not (((hi >> 30) & 3) == 1).any(), f"{what}: a RAW hit record was exported"

    def advance(k=1):
        for _ in range(k):
            cnts = []
            for c in (g, o):
                c.wf_logic(False); c.wf_raygen(); c.wf_materials()
                cc = c.get_counters(); c.finish(); cnts.append(np.array(cc, copy=True))
                c.wf_extend(); c.wf_shadow(); c.clear_queues()
            assert (cnts[0] == cnts[1]).all()
            for c in (g, o):
                c.pixel_index_update(npix, int(cnts[1][Q.RAYGEN]))

    advance(3); cmp("3 iterations of the reference's loop (raw records consumed by the fused pass each time)")
    # a read-back right after the extension kernel, then the loop goes on from the committed records
    for c in (g, o):
        c.wf_logic(False); c.wf_raygen(); c.wf_materials(); c.wf_extend()
    cmp("export right after flx_wf_extend")
    for c in (g, o):
        c.wf_shadow(); c.clear_queues(); c.pixel_index_update(npix, 64)
    advance(1); cmp("loop continued after the in-memory commit")
    # the extension kernel twice in a row (pathLen += 2, as the reference's kernel would)
    for c in (g, o):
        c.wf_logic(False); c.wf_raygen(); c.wf_materials(); c.wf_extend(); c.wf_extend(); c.wf_shadow(); c.clear_queues(); c.pixel_index_update(npix, 64)
    cmp("two extension launches back to back")
    # the separate kernels after a persistent extension launch: option change, then logic alone, raygen, materials
    for c in (g, o):
        c.wf_logic(False); c.wf_raygen(); c.wf_materials(); c.wf_extend(); c.wf_shadow(); c.clear_queues(); c.pixel_index_update(npix, 64)
    g.set_option("fuse", 0)
    advance(2); cmp("fuse 0 after raw records were pending")
    g.set_option("fuse", 1)
    advance(1)
    # chain without genRays: the fused pass may not take raw records (terminated paths would keep them), so they are committed first
    for c in (g, o):
        c.wf_logic(False); c.wf_materials(); c.wf_extend(); c.wf_shadow(); c.clear_queues(); c.pixel_index_update(npix, 0)
    advance(1); cmp("logic -> materials without genRays")
    # materials first, then genRays; counters and a queue read between the extension kernel and logic; parameters changed in between
    p2 = p.copy(); p2["maxBounces"] = 3
    for c in (g, o):
        c.wf_logic(False); c.wf_materials(); c.wf_raygen(); c.wf_extend(); c.wf_shadow()
        c.queue_read(Q.EXTENSION); c.clear_queues(); c.set_params(p2); c.pixel_index_update(npix, 17)
    advance(1)
    for c in (g, o):
        c.set_params(p)
    advance(1); cmp("queue read, parameter change")
    # the thread-per-ray kernel and the bit-exact binary kernel take over from raw records and hand back
    for tree, refill in ((4, 0), (2, 0), (4, 16 | (32 << 8))):
        g.set_option("extend_tree", tree); g.set_option("refill_extend", refill)
        advance(2); cmp(f"extend_tree {tree}, refill_extend {refill}")
    # first-frame sequence of Tracer::update (reset, genRays, extension, then three preview iterations with firstIteration set)
    for c in (g, o):
        driver.reset_renderer(c)
        driver.first_frame(c, p, npix)
    cmp("first frame")
    advance(2); cmp("after the first frame")
    assert common.fb_close(g.read_pixels(0), o.read_pixels(0))
    g.close()


def test_persistent_kernels_with_empty_and_tiny_queues():
    """Edge cases of the block hand-out: an EMPTY extension / shadow queue (every wave finds its XCD's list exhausted, tries the other seven
    and leaves), a queue of one ray, one of 65 (a full block and a block of one), with both traversals persistent; state and counters as the
    oracle's after each."""
    d = common.mixed_material_scene()
    w, h, n = 16, 8, 1024
    p = common.scene_params(d, w, h, maxBounces=4, useAreaLight=1, useEnvMap=1, wfSeparateQueues=1)
    g, o = _ctxs(d, p, n, env=host.synthetic_sky(64, 32))
    g.set_option("refill_shadow", 16 | (32 << 8))
    for c in (g, o):                                   # nothing generated yet: every queue is empty
        c.wf_extend(); c.wf_shadow()
    assert not common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
    assert (np.array(g.get_counters()) == np.array(o.get_counters())).all()
    g.finish()
    for it in range(3):
        for c in (g, o):
            c.wf_logic(False); c.wf_raygen(); c.wf_materials()
        cnt = np.array(o.get_counters(), copy=True)
        q = o.queue_read(Q.EXTENSION)[:int(cnt[Q.EXTENSION])].copy()
        for keep in (0, 1, 65):                        # trace only the first `keep` rays of the queue: on BOTH sides, from the same state
            for c in (g, o):
                cc = cnt.copy(); cc[Q.EXTENSION] = min(keep, q.size); cc[Q.SHADOW] = min(keep, int(cnt[Q.SHADOW]))
                c.set_counters(cc)
            common.sync(g, o)
            for c in (g, o):
                c.wf_extend(); c.wf_shadow()
            g.finish()
            fails = common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
            assert not fails, f"it{it}, {keep} rays: " + "; ".join(fails[:3])
        for c in (g, o):
            c.set_counters(cnt)
        common.sync(g, o)
        for c in (g, o):
            c.wf_extend(); c.wf_shadow(); c.clear_queues(); c.pixel_index_update(w * h, int(cnt[Q.RAYGEN]))
    assert not common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
    g.close()


@pytest.mark.parametrize("split", [1, 2 | (1 << 8), 8, 8 | (8 << 8), 3 | (200 << 8)])
def test_shadow_tail_split_is_bit_identical_small(split):
    """Tail splitting of the any-hit query (trace4.hip: k_shadow4s; option shadow_split = budget of the pass over the queue | budget of a second pass << 8):
    a ray that has spent its node-visit budget suspends into a 64-byte continuation record and a later launch resumes it.  The ray's own visit sequence is
    k_shadow4's, so shadowRayBlocked must be IDENTICAL -- here with budgets of 1-3 visits, where nearly every ray suspends (twice), on the all-BSDF scene with
    both light types (area light: last-hit-slot-first order; environment only: far -> near order), with an odd path count (a partial last wave), and with
    the record sub-lists cut to 5 slots (option shadow_split_limit), so that they OVERFLOW and the rays without a record finish inside their wave."""
    from fluctus_amd.device import HipContext
    d = common.mixed_material_scene()
    w, h = 64, 48
    for area, envm, n, limit in ((1, 1, 4096 + 37, 0), (0, 1, 4096 + 37, 0), (1, 0, 4096 + 37, 5), (0, 1, 192, 1)):
        p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=area, useEnvMap=envm, wfSeparateQueues=1)
        ctx = []
        for s in (0, split):
            g = HipContext(n)
            g.upload_scene(d); g.upload_envmap(host.synthetic_sky(64, 32)); g.set_params(p)
            g.set_option("shadow_split", s); g.set_option("shadow_split_limit", limit)
            driver.reset_renderer(g)
            ctx.append(g)
        a, b = ctx
        assert a.get_option("shadow_split") == 0 and b.get_option("shadow_split") == split
        rays = 0
        for it in range(10):
            ca, cb = driver.benchmark_iteration(a, w * h), driver.benchmark_iteration(b, w * h)
            assert (ca == cb).all(), f"it{it}: {ca} vs {cb}"
            rays += int(ca[Q.SHADOW])
            sa, sb = a.state_export(), b.state_export()
            assert np.array_equal(sa.view(np.uint32)[COL.SHADOW_BLOCKED], sb.view(np.uint32)[COL.SHADOW_BLOCKED]), f"area {area} env {envm} n {n} it{it}: shadowRayBlocked"
            fails = common.state_diff(sa, sb, 0.0, 0.0)
            assert not fails, "; ".join(fails[:4])
        assert rays > 2 * n
        for g in ctx:
            g.close()


@pytest.mark.parametrize("workload", ["kitchen", "conference", "courtyard-1440p"])
def test_shadow_tail_split_is_bit_identical_full_size(workload):
    """The same at BASELINE's sizes, 1 M paths, the budgets the bench uses and a three-pass setting: 12 free-running iterations of the shipped default
    path with and without the split, counters every iteration, the whole state at the end."""
    from fluctus_amd.device import HipContext
    import bench
    d, p, env = bench.build_workload(name=workload)
    n, npix = 1 << 20, int(p["width"]) * int(p["height"])
    ref = None
    for s in (0, 8, 8 | (8 << 8)):
        g = HipContext(n)
        g.upload_scene(d); g.upload_envmap(env); g.set_params(p)
        g.set_option("shadow_split", s)
        driver.reset_renderer(g)
        cnts = [driver.benchmark_iteration(g, npix) for _ in range(12)]
        st, px = g.state_export(), g.read_pixels(0)
        if ref is None:
            ref = (cnts, st, px)
        else:
            assert all((x == y).all() for x, y in zip(cnts, ref[0])), f"{workload} split {s}: counters"
            fails = common.state_diff(st, ref[1], 0.0, 0.0)
            assert not fails, f"{workload} split {s}: " + "; ".join(fails[:4])
            assert common.fb_close(px, ref[2])
        g.close()
