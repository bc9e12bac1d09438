"""Pin the CPU oracle (oracle/wf_oracle.cpp) against the reference's OWN kernels.

oracle/_ref/libfluctus_ref.so is the reference's wf_*.cl / mk_postprocess.cl compiled unmodified for
x86-64 (oracle/ref/Makefile).  It exists only in the build container (needs /root/reference), so these
tests skip elsewhere; the golden fixtures in tests/golden (made from the same build by
scripts/make_golden.py) carry the pin to the GPU box.

Method: LOCKSTEP.  Both implementations start every kernel from the SAME state (the oracle's state,
queues and counters are imported into the reference context), run one kernel, and are compared.
Integers (hit index, matId, seeds, flags, queue contents in canonical order, counters) must be exact;
floats within rtol 1e-5 (1e-4 downstream of atan2/acos/native_* as stated in SURVEY 8(c)), which is the
gap between include/flx_math.h and libm.
"""
import numpy as np
import pytest
import common
from common import COL, Q
from fluctus_amd import host, wire, driver
from oracle.binding import OracleContext, ref_available

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (needs /root/reference)")]

RTOL, ATOL = 2e-5, 2e-6


def _pair(d, p, n, env=None):
    from oracle.binding import RefContext
    a, b = OracleContext(n), RefContext(n)
    for c in (a, b):
        c.upload_scene(d)
        if env is not None:
            c.upload_envmap(env)
        c.set_params(p)
        driver.reset_renderer(c)
    return a, b


def _step(a, b, name, fn, rtol=RTOL, atol=ATOL, skip=(), undefined_pdfw=False):
    common.sync(b, a)
    fn(a)
    fn(b)
    ca, cb = a.get_counters(), b.get_counters()
    assert (ca == cb).all(), f"{name}: counters {ca} vs {cb}"
    for q in range(8):
        n = int(ca[q])
        qa, qb = a.queue_read(q)[:n], b.queue_read(q)[:n]
        assert np.array_equal(qa, qb), f"{name}: queue {q} order differs"
    sa, sb = a.state_export(), b.state_export()
    mask = None
    if undefined_pdfw:
        # reference leaves pdfW uninitialised when the glossy sampler rejects (glossy.cl:59-60); T is 0 there
        mask = ~((sa[COL.T] == 0) & (sa[COL.T + 1] == 0) & (sa[COL.T + 2] == 0))
    fails = common.state_diff(sa, sb, rtol, atol, skip_cols=skip, mask=mask)
    assert not fails, f"{name}: " + "; ".join(fails[:5])


def _iterate(a, b, npix, iters, rtol=RTOL, atol=ATOL, mat_rtol=None):
    for it in range(iters):
        _step(a, b, f"it{it} logic", lambda c: c.wf_logic(False), rtol, atol)
        _step(a, b, f"it{it} raygen", lambda c: c.wf_raygen(), rtol, atol)
        _step(a, b, f"it{it} materials", lambda c: c.wf_materials(), mat_rtol or rtol, atol, undefined_pdfw=True)
        cnt = a.get_counters().copy()
        _step(a, b, f"it{it} extend", lambda c: c.wf_extend(), rtol, atol)
        _step(a, b, f"it{it} shadow", lambda c: c.wf_shadow(), rtol, atol)
        for c in (a, b):
            c.clear_queues()
            c.pixel_index_update(npix, int(cnt[0]))
    pa, pb = a.read_pixels(0), b.read_pixels(0)
    assert np.array_equal(pa[:, 3], pb[:, 3])
    assert np.allclose(pa, pb, rtol=1e-4, atol=1e-5)
    for c in (a, b):
        c.postprocess()
    assert np.allclose(a.read_pixels(1), b.read_pixels(1), rtol=1e-4, atol=1e-5)


def test_teapot_area_light_single_queue():
    """Config 1 of BASELINE.json: teapot.ply, Lambertian, area light, 4 bounces."""
    d = host.load_scene(common.REF_ASSETS + "/teapot.ply")
    host.build_bvh(d, "sbvh")
    w = h = 96
    p = wire.default_params(w, h, d.world_radius, d.tris.size)
    p["maxBounces"] = 4
    a, b = _pair(d, p, w * h)
    _iterate(a, b, w * h, 10)


@pytest.mark.parametrize("area,env,expl,impl,sep,roulette", [
    (1, 0, 1, 1, 1, 0),
    (0, 1, 1, 1, 1, 0),
    (1, 1, 1, 1, 0, 1),
    (1, 1, 1, 0, 1, 0),
    (1, 1, 0, 1, 1, 0),
    (0, 0, 1, 1, 1, 1),
])
def test_all_bsdfs_flag_matrix(area, env, expl, impl, sep, roulette):
    """All six BSDFs + textures + normal map, env-map MIS / area light NEE, separate vs single queue, RR."""
    d = common.mixed_material_scene()
    w, h, n = 64, 48, 4096
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=area, useEnvMap=env, sampleExpl=expl, sampleImpl=impl,
                            wfSeparateQueues=sep, useRoulette=roulette, envMapStrength=1.5)
    e = host.synthetic_sky(64, 32)
    a, b = _pair(d, p, n, env=e)
    # env-map lookups and the GGX lobe sit downstream of atan2/acos/sin/cos: SURVEY 8(c) allows 1e-4 there
    # ... and GGX D/pdf for small alpha amplify ulp-level differences of sin/cos/atan2 (observed <= 2e-4 rel)
    _iterate(a, b, w * h, 9, rtol=1e-4, atol=1e-5, mat_rtol=1e-3)


def test_first_frame_preview_path():
    """Tracer::update iteration 0: reset, raygen, extend, then 3 logic rounds with firstIteration=1, 2 bounces."""
    d = common.simple_scene()
    w, h, n = 40, 30, 2048       # numTasks > numPixels exercises maxId = min(w*h, numTasks)
    p = common.scene_params(d, w, h, maxBounces=6)
    a, b = _pair(d, p, n)
    p2 = p.copy()
    p2["maxBounces"] = 2
    for c in (a, b):
        c.set_params(p2)
        c.pixel_index_reset()
    _step(a, b, "reset", lambda c: c.wf_reset())
    _step(a, b, "raygen0", lambda c: c.wf_raygen())
    _step(a, b, "extend0", lambda c: c.wf_extend())
    for c in (a, b):
        c.clear_queues()
    for r in range(3):
        _step(a, b, f"r{r} logic", lambda c: c.wf_logic(True))
        _step(a, b, f"r{r} raygen", lambda c: c.wf_raygen())
        _step(a, b, f"r{r} materials", lambda c: c.wf_materials())
        _step(a, b, f"r{r} extend", lambda c: c.wf_extend())
        _step(a, b, f"r{r} shadow", lambda c: c.wf_shadow())
        for c in (a, b):
            c.clear_queues()


def test_egyptcat_real_asset_textured_glossy():
    """A real reference asset end to end through the loaders: egyptcat.obj + .mtl (`shader glossy`, Ns 100000) +
    EgyptCat.png (1024^2 diffuse texture), area light, single material queue."""
    d = host.load_scene(common.REF_ASSETS + "/egyptcat/egyptcat.obj")
    host.build_bvh(d, "sbvh")
    assert d.texdesc.size == 1
    w, h, n = 64, 48, 4096
    p = wire.default_params(w, h, d.world_radius, d.tris.size)
    lo = np.array([d.nodes[0]["bmin"][k] for k in "xyz"]); hi = np.array([d.nodes[0]["bmax"][k] for k in "xyz"])
    c = 0.5 * (lo + hi)
    wire.look_at(p, c + np.array([0.0, 0.2, 1.4]) * d.world_radius, c)
    al = p["areaLight"]
    al["pos"]["x"], al["pos"]["y"], al["pos"]["z"] = c + np.array([1.2, 0.8, 0.3]) * d.world_radius
    al["size"] = (0.4 * d.world_radius, 0.4 * d.world_radius)
    p["maxBounces"] = 4
    a, b = _pair(d, p, n)
    _iterate(a, b, w * h, 8, rtol=1e-4, atol=1e-5, mat_rtol=2e-3)


# ---------------------------------------------------------------- microkernel integrator
MK_PHASES = (("raygen", lambda c: c.mk_raygen()), ("next_vertex", lambda c: c.mk_next_vertex()), ("sample_bsdf", lambda c: c.mk_sample_bsdf()))


def _mk_step(a, b, name, fn, rtol, atol):
    b.state_import(a.state_export())
    fn(a); fn(b)
    sa, sb = a.state_export(), b.state_export()
    # a path whose sampler rejected (bsdf == 0) divides by the reference's uninitialised pdfW: its T / lastPdfW are dead
    # (phase -> splat resets them) and undefined in the reference
    dead = np.zeros(sa.shape[1], bool)
    if name == "sample_bsdf":
        t = sa[COL.T:COL.T + 3]
        dead = ~np.isfinite(t).all(0) | ((t == 0).all(0))
    fails = common.state_diff(sa, sb, rtol, atol, mask=~dead)
    assert not fails, f"{name}: " + "; ".join(fails[:5])
    assert np.array_equal(sa.view(np.uint32)[COL.PHASE], sb.view(np.uint32)[COL.PHASE]), name


@pytest.mark.parametrize("area,env,expl,impl,roulette", [(1, 0, 1, 1, 0), (0, 1, 1, 1, 0), (1, 1, 1, 1, 1), (1, 1, 0, 1, 0), (1, 1, 1, 0, 0)])
def test_microkernel_integrator_lockstep(area, env, expl, impl, roulette):
    d = common.mixed_material_scene()
    w, h = 48, 32
    n = w * h
    p = common.scene_params(d, w, h, maxBounces=4, useAreaLight=area, useEnvMap=env, sampleExpl=expl, sampleImpl=impl, useRoulette=roulette,
                            envMapStrength=1.5)
    from oracle.binding import RefContext
    a, b = OracleContext(n), RefContext(n)
    e = host.synthetic_sky(64, 32)
    for c in (a, b):
        c.upload_scene(d); c.upload_envmap(e); c.set_params(p)
        c.mk_reset()
    fails = common.state_diff(a.state_export(), b.state_export(), 0, 0, skip_cols=[COL.P, COL.P + 1, COL.P + 2])
    for spp in range(3):
        _mk_step(a, b, "raygen", MK_PHASES[0][1], RTOL, ATOL)
        for bounce in range(int(p["maxBounces"]) + 1):
            _mk_step(a, b, "next_vertex", MK_PHASES[1][1], 1e-4, 1e-5)
            _mk_step(a, b, "sample_bsdf", MK_PHASES[2][1], 1e-3, 1e-5)
        b.state_import(a.state_export())
        a.mk_splat(); b.mk_splat()
        assert np.allclose(a.read_pixels(0), b.read_pixels(0), rtol=1e-3, atol=1e-4)
        assert np.array_equal(a.read_pixels(0)[:, 3], b.read_pixels(0)[:, 3])
        assert not common.state_diff(a.state_export(), b.state_export(), 1e-4, 1e-5)
    sa, sb = a.mk_stats(), b.mk_stats()
    assert np.array_equal(sa, sb)
    if not roulette:                                                  # renderSingle switches roulette off (src/tracer.cpp:104-108):
        assert (a.read_pixels(0)[:, 3] == 3).all()                  # then every pass adds exactly one sample to every pixel
        assert sa[3] == 3 * n and sa[0] == 3 * n
    a.mk_splat_preview(); b.mk_splat_preview()
    assert np.allclose(a.read_pixels(0), b.read_pixels(0), rtol=1e-3, atol=1e-4)


# ---------------------------------------------------------------- denoiser feature buffers
def _aov_close(a, b, name):
    for which, tag in ((4, "albedo accumulator"), (5, "normal accumulator")):
        xa, xb = a.read_pixels(which), b.read_pixels(which)
        assert np.array_equal(xa[:, 3], xb[:, 3]), f"{name}: {tag} counts differ"
        assert np.allclose(xa, xb, rtol=1e-4, atol=1e-5), f"{name}: {tag}"


@pytest.mark.parametrize("sep", [0, 1])
def test_denoiser_features_wavefront(sep):
    """The reference's USE_OPTIX_DENOISER build of `logic` / `process` (wf_logic.cl:186-209, mk_postprocess.cl:49-54) against
    the oracle's `denoiser` option, in lockstep: first-hit camera-space normals and first-diffuse-hit albedo per pixel."""
    d = common.mixed_material_scene()
    w, h, n = 48, 32, 4096
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=1, useEnvMap=1, wfSeparateQueues=sep)
    a, b = _pair(d, p, n, env=host.synthetic_sky(64, 32))
    for c in (a, b):
        c.set_option("denoiser", 1)
        driver.reset_renderer(c)
    al = a.read_pixels(4)
    assert np.array_equal(al, b.read_pixels(4)) and np.allclose(al, [0.1, 0.1, 0.1, 0.0])      # wf_reset.cl:23-24
    for it in range(10):
        _step(a, b, f"it{it} logic", lambda c: c.wf_logic(False), 1e-4, 1e-5)
        _aov_close(a, b, f"it{it}")
        _step(a, b, f"it{it} raygen", lambda c: c.wf_raygen(), 1e-4, 1e-5)
        _step(a, b, f"it{it} materials", lambda c: c.wf_materials(), 2e-3, 1e-5, undefined_pdfw=True)
        cnt = a.get_counters().copy()
        _step(a, b, f"it{it} extend", lambda c: c.wf_extend(), 1e-4, 1e-5)
        _step(a, b, f"it{it} shadow", lambda c: c.wf_shadow(), 1e-4, 1e-5)
        for c in (a, b):
            c.clear_queues(); c.pixel_index_update(w * h, int(cnt[0]))
    na, aa = a.read_pixels(5), a.read_pixels(4)
    assert na[:, 3].sum() > 0 and aa[:, 3].sum() > 0 and aa[:, 3].sum() <= na[:, 3].sum()   # every path: one normal, at most one albedo
    for c in (a, b):
        c.postprocess()
    for which in (2, 3):
        assert np.allclose(a.read_pixels(which), b.read_pixels(which), rtol=1e-4, atol=1e-5)
    nrm = a.read_pixels(3)
    done = nrm[:, 3] == 1.0                                           # resolved: sums divided by the count
    assert done.any() and (np.linalg.norm(nrm[done, :3], axis=1) <= 1.0 + 1e-4).all()


def test_denoiser_features_microkernel():
    """mk_next_vertex.cl:59-69, mk_sample_bsdf.cl:56-66, mk_reset.cl:24-25 with USE_OPTIX_DENOISER vs the oracle."""
    d = common.mixed_material_scene()
    w, h = 48, 32
    n = w * h
    p = common.scene_params(d, w, h, maxBounces=4, useAreaLight=1, useEnvMap=1)
    from oracle.binding import RefContext
    a, b = OracleContext(n), RefContext(n)
    e = host.synthetic_sky(64, 32)
    for c in (a, b):
        c.upload_scene(d); c.upload_envmap(e); c.set_params(p); c.set_option("denoiser", 1)
        c.mk_reset()
    for spp in range(3):
        _mk_step(a, b, "raygen", MK_PHASES[0][1], RTOL, ATOL)
        for bounce in range(int(p["maxBounces"]) + 1):
            _mk_step(a, b, "next_vertex", MK_PHASES[1][1], 1e-4, 1e-5)
            _aov_close(a, b, "next_vertex")
            _mk_step(a, b, "sample_bsdf", MK_PHASES[2][1], 1e-3, 1e-5)
            _aov_close(a, b, "sample_bsdf")
        b.state_import(a.state_export())
        a.mk_splat(); b.mk_splat()
    assert (a.read_pixels(5)[:, 3] == 3).all()                       # one first-hit normal per pixel per pass
    for c in (a, b):
        c.postprocess()
    for which in (2, 3):
        assert np.allclose(a.read_pixels(which), b.read_pixels(which), rtol=1e-4, atol=1e-5)


def test_reference_kernels_on_host_threads_agree_statistically():
    """bench.py's cpu_baseline (kind "reference") spreads the NDRanges of the reference kernels over host threads, like a CPU
    OpenCL device: queue ORDER then depends on timing (as it does on the reference's real devices), the estimate must not."""
    from oracle.binding import RefContext
    d = common.mixed_material_scene()
    w, h, n = 64, 48, 4096
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=1, useEnvMap=1, wfSeparateQueues=1)
    outs = []
    for threads in (1, 4):
        c = RefContext(n, threads=threads)
        c.upload_scene(d); c.upload_envmap(host.synthetic_sky(64, 32)); c.set_params(p)
        driver.reset_renderer(c)
        tot = np.zeros(8, np.int64)
        for _ in range(60):
            cnt = driver.benchmark_iteration(c, w * h)
            assert int(cnt[Q.RAYGEN]) + int(cnt[3:8].sum()) == n and int(cnt[Q.EXTENSION]) == n      # invariants hold under real atomics
            tot += cnt
        px = c.read_pixels(0)
        assert np.isfinite(px).all()
        outs.append((tot, px[:, :3].sum() / max(1.0, px[:, 3].sum())))
    (t1, m1), (t4, m4) = outs
    assert abs(int(t1[Q.RAYGEN]) - int(t4[Q.RAYGEN])) <= 0.03 * t1[Q.RAYGEN] and abs(int(t1[Q.SHADOW]) - int(t4[Q.SHADOW])) <= 0.03 * t1[Q.SHADOW]
    assert abs(m1 - m4) <= 0.05 * m1
