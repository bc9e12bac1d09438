"""GPU parity of the denoiser feature buffers (reference: USE_OPTIX_DENOISER kernel builds) vs the CPU oracle."""
import numpy as np
import pytest
import common
from common import COL
from fluctus_amd import host, wire, driver

pytestmark = pytest.mark.gpu


def _ctxs(d, p, n, env=None, overlap=2):
    from fluctus_amd.device import HipContext
    from oracle.binding import OracleContext
    g, o = HipContext(n), OracleContext(n, threads=8)
    g.set_option("extend_tree", 2)          # bit-exact comparisons: the reference's visit order
    g.set_option("overlap", overlap)
    for c in (g, o):
        c.set_option("denoiser", 1)
        c.upload_scene(d)
        if env is not None:
            c.upload_envmap(env)
        c.set_params(p)
    return g, o


@pytest.mark.parametrize("sep,overlap", [(0, 2), (1, 2), (1, 0)])
def test_wavefront_features_lockstep(sep, overlap):
    d = common.mixed_material_scene()
    w, h, n = 64, 48, 8192                    # several paths per pixel in flight: float atomics on the same pixel
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=1, useEnvMap=1, wfSeparateQueues=sep)
    g, o = _ctxs(d, p, n, env=host.synthetic_sky(64, 32), overlap=overlap)
    for c in (g, o):
        driver.reset_renderer(c)
    assert np.array_equal(g.read_pixels(4), o.read_pixels(4)) and np.array_equal(g.read_pixels(5), o.read_pixels(5))
    for it in range(10):
        common.sync(g, o)
        for c in (g, o):
            c.wf_logic(False)
        assert not common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0), f"it{it} logic"
        for which in (4, 5):
            a, b = g.read_pixels(which), o.read_pixels(which)
            assert np.array_equal(a[:, 3], b[:, 3])                                  # counts: exact
            # sums: the order of the float atomics within a pixel is free.  The normal buffer adds SIGNED components of unit vectors, so a
            # sum can cancel to near zero while every partial sum is O(count): absolute tolerance per add, relative tolerance on top
            tol = 1e-6 * np.abs(b) + 2e-7 * np.maximum(1.0, b[:, 3:4])
            assert (np.abs(a - b) <= tol).all(), (it, which, float(np.abs(a - b).max()))
        cnt = driver_step_rest(g, o, w * h)
    for c in (g, o):
        c.postprocess()
    for which in (2, 3):
        assert np.allclose(g.read_pixels(which), o.read_pixels(which), rtol=1e-6, atol=1e-6)       # resolved sums / count: see above
    assert g.read_pixels(5)[:, 3].sum() > 0


def driver_step_rest(g, o, npix):
    """raygen, materials, extend, shadow, end of iteration on both contexts (logic already ran)."""
    for c in (g, o):
        c.wf_raygen(); c.wf_materials()
    cnt = o.get_counters().copy()
    for c in (g, o):
        c.wf_extend(); c.wf_shadow(); c.clear_queues(); c.pixel_index_update(npix, int(cnt[0]))
    return cnt


def test_microkernel_features_bit_exact_and_off_by_default():
    d = common.mixed_material_scene()
    w, h, spp = 80, 60, 4
    p = common.scene_params(d, w, h, maxBounces=4, useAreaLight=1, useEnvMap=1)
    g, o = _ctxs(d, p, w * h, env=host.synthetic_sky(64, 32))
    driver.render_single(g, p, spp)
    driver.render_single(o, p, spp)
    for which in (0, 1, 2, 3, 4, 5):                                     # thread = pixel: no atomics, every buffer bit-identical
        assert np.array_equal(g.read_pixels(which).view(np.uint32), o.read_pixels(which).view(np.uint32)), which
    nrm = g.read_pixels(3)
    assert (g.read_pixels(5)[:, 3] == spp).all() and (nrm[:, 3] == 1.0).all()
    assert not common.state_diff(g.state_export(), o.state_export(), 0.0, 0.0)
    # switching the option off frees the buffers; reading them then fails loudly, rendering still works
    g.set_option("denoiser", 0)
    with pytest.raises(RuntimeError, match="denoiser"):
        g.read_pixels(2)
    driver.render_single(g, p, 1)
    assert (g.read_pixels(0)[:, 3] == 1).all()


def test_cpp_tracer_render_single_with_features():
    from fluctus_amd.tracer import Tracer
    from oracle.binding import OracleContext
    w, h = 64, 48
    t = Tracer(w, h, 0, w * h)
    t.init(w, h, "proc:kitchen:6000:3")
    p = t.params
    wire.look_at(p, (0.0, 1.2, 2.6), (0.0, 0.2, 0.0))
    p["maxBounces"] = 3
    t.params = p
    t.render_single(3, denoise=True)
    d = host.generate_scene("kitchen", 6000, 3)
    host.build_bvh(d, "sbvh")
    o = OracleContext(w * h, threads=8)
    o.set_option("denoiser", 1)
    o.upload_scene(d)
    driver.render_single(o, t.params, 3)
    for which in (0, 2, 3):
        assert np.array_equal(t.read_pixels(which).view(np.uint32), o.read_pixels(which).view(np.uint32)), which
