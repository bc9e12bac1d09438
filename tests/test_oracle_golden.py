"""The CPU oracle against the committed golden fixtures (tests/golden/*.npz), which hold outputs of the
REFERENCE's own kernels (generated in the build container by scripts/make_golden.py from oracle/_ref).
Runs everywhere (no /root/reference needed): this is what pins the oracle on the GPU box."""
import os
import numpy as np
import pytest
import common
from common import COL
from fluctus_amd import host, wire, driver
from oracle.binding import OracleContext


def _load_scene(z):
    return common.fixture_scene(z)


def _fixture(name):
    path = os.path.join(common.GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} missing")
    return np.load(path)


def test_raygen_golden():
    """G1+G2: RNG stream (4 draws per primary ray), pixel assignment, thin-lens camera rays."""
    z = _fixture("raygen.npz")
    n = int(z["num_tasks"])
    p = z["params"].view(wire.RENDER_PARAMS).reshape(())
    d = common.simple_scene()
    c = OracleContext(n)
    c.upload_scene(d); c.set_params(p)
    c.pixel_index_reset(); c.wf_reset(); c.wf_raygen()
    assert np.array_equal(c.get_counters(), z["counters"])
    assert np.array_equal(c.queue_read(1), z["ext_queue"])
    fails = common.state_diff(c.state_export(), z["state"], 2e-6, 1e-6)
    assert not fails, "; ".join(fails)


@pytest.mark.parametrize("tag", ["area_sep", "env_area_single_rr", "denoiser_env_area_sep", "egyptcat"])
def test_kernel_steps_golden(tag):
    """G3/G4/G6: every kernel of two iterations, all six BSDFs, textures, env-map MIS: oracle output from the
    reference's input state vs the reference's output state.  "egyptcat": the reference's own benchmark scene #1 (real asset, its
    1024^2 texture, the reference's start-up parameters; scripts/make_egyptcat_fixture.py)."""
    z = _fixture(f"steps_{tag}.npz")
    n = int(z["num_tasks"])
    p = z["params"].view(wire.RENDER_PARAMS).reshape(())
    d = _load_scene(z)
    w, h = int(z["env_wh"][0]), int(z["env_wh"][1])
    e = host.EnvMap(w, h, z["env_rgb"], z["env_prob"], z["env_alias"], z["env_pdf"])
    c = OracleContext(n)
    den = "aov" in z.files                          # made with the reference's USE_OPTIX_DENOISER kernel builds
    if den:
        c.set_option("denoiser", 1)
    c.upload_scene(d); c.upload_envmap(e); c.set_params(p)
    if den:
        c.wf_reset()
    names = [str(s) for s in z["names"]]
    npix = int(p["width"]) * int(p["height"])
    # the fixture starts after 6 free-running reference iterations; replay the cursor from the counters
    c.pixel_index_reset()
    fn = {"logic": lambda: c.wf_logic(False), "raygen": c.wf_raygen, "materials": c.wf_materials, "extend": c.wf_extend, "shadow": c.wf_shadow}
    for k in range(1, len(names)):
        if names[k] == "end":
            continue
        prev = k - 1
        c.state_import(z["states"][prev])
        for q in range(8):
            c.queue_write(q, z["queues"][prev][q])
        c.set_counters(z["counters"][prev])
        c.pixel_index_reset(); c.pixel_index_update(npix, int(z["pixel_cursor"][prev]))     # the reference's cursor at that point
        if den:
            before = np.stack([c.read_pixels(4), c.read_pixels(5)])
        fn[names[k]]()
        if den:                                     # the kernel's contribution to the denoiser feature accumulators
            want = z["aov"][k] - z["aov"][k - 1]
            got = np.stack([c.read_pixels(4), c.read_pixels(5)]) - before
            assert np.array_equal(got[..., 3], want[..., 3]) and np.allclose(got, want, rtol=1e-4, atol=1e-4), names[k]
        assert np.array_equal(c.get_counters(), z["counters"][k]), (k, names[k])
        for q in range(8):
            m = int(z["counters"][k][q])
            assert np.array_equal(c.queue_read(q)[:m], z["queues"][k][q][:m]), (names[k], q)
        sa, sb = c.state_export(), z["states"][k]
        mask = None
        if names[k] == "materials":   # pdfW is uninitialised in the reference when the glossy sampler rejects (T == 0 there)
            mask = ~((sa[COL.T] == 0) & (sa[COL.T + 1] == 0) & (sa[COL.T + 2] == 0))
        rtol = 1e-3 if names[k] == "materials" else 1e-4
        col_rtol = {COL.LAST_PDF_W: np.where(common.sharp_lobe_paths(d, sb), 2e-2, rtol)} if names[k] == "materials" else None
        fails = common.state_diff(sa, sb, rtol, 1e-5, mask=mask, col_rtol=col_rtol)
        assert not fails, f"step {k} {names[k]}: " + "; ".join(fails[:4])


def test_teapot_end_to_end_golden():
    """G7 / config 1: teapot.ply, 128x128, 4 bounces, area light, 24 free-running iterations."""
    z = _fixture("teapot_wf.npz")
    n = int(z["num_tasks"])
    p = z["params"].view(wire.RENDER_PARAMS).reshape(())
    d = _load_scene(z)
    w, h = int(p["width"]), int(p["height"])
    c = OracleContext(n, threads=4)
    c.upload_scene(d); c.set_params(p); driver.reset_renderer(c)
    cnts = z["counters"]
    for it in range(cnts.shape[0]):
        cnt = driver.benchmark_iteration(c, w * h)
        # free-running: ulp-level differences (libm vs flx_math) flip a few grazing rays per iteration (<= 1e-5 of rays
        # per kernel; they accumulate over iterations)
        assert np.all(np.abs(cnt.astype(np.int64) - cnts[it].astype(np.int64)) <= max(4, int(2e-3 * n))), (it, cnt, cnts[it])
    # One flipped grazing ray changes when its path terminates, hence the order of the raygen queue and the pixel/seed pairing of
    # every later regenerated path (src/wf_raygen.cl:25): a FREE run against the libm build forks.  The exact comparison is
    # test_teapot_resynchronised_iterations_golden (every iteration restarted from the reference's state); here only the conserved
    # quantities of the free run are checked.
    px, ref = c.read_pixels(0), z["pixels"]
    assert np.abs(px[:, 3] - ref[:, 3]).max() <= 3
    assert abs(px[:, 3].sum() - ref[:, 3].sum()) <= 2e-3 * ref[:, 3].sum()
    assert abs(px[:, :3].sum() / px[:, 3].sum() - ref[:, :3].sum() / ref[:, 3].sum()) <= 5e-3 * ref[:, :3].sum() / ref[:, 3].sum()


def resync_check(c, z, flip_budget):
    """Shared by the oracle (here) and the device (tests/test_gpu_parity.py): every iteration of tests/golden/teapot_resync.npz from
    the REFERENCE's own pre-iteration state.  Returns (rays, flips).  Per iteration: queue counters exact; path state vs the
    reference's next pre-state -- integers exact, floats rtol 1e-3 / atol 1e-4, except on paths whose hit
    index flipped (counted, asserted against the budget); framebuffer delta of the iteration: sample counts exact where no flip
    landed, sums rtol 1e-4."""
    p = z["params"].view(wire.RENDER_PARAMS).reshape(())
    npix = int(p["width"]) * int(p["height"])
    n = int(z["num_tasks"])
    rays = flips = 0
    for k in range(z["counters"].shape[0]):
        c.state_import(z["states"][k])
        c.set_counters(np.zeros(8, np.uint32))
        c.pixel_index_reset(); c.pixel_index_update(npix, int(z["pixel_cursor"][k]))
        before = c.read_pixels(0)
        cnt = driver.benchmark_iteration(c, npix)
        assert np.array_equal(cnt, z["counters"][k]), (k, cnt, z["counters"][k])
        sa, sb = c.state_export(), z["states"][k + 1]
        flip = sa.view(np.uint32)[COL.HIT_I] != sb.view(np.uint32)[COL.HIT_I]
        rays += n; flips += int(flip.sum())
        # one whole iteration chains the BSDF sample (sin / cos / atan2: libm vs flx_math, ~1e-7) into the next ray and its hit: on the
        # teapot's high-curvature patches that moves an interpolated unit normal by up to ~3e-5 -> absolute tolerance 1e-4 here
        # (per-kernel, from identical inputs: 1e-5, test_kernel_steps_golden)
        fails = common.state_diff(sa, sb, 1e-3, 1e-4, mask=~flip)
        assert not fails, f"iteration {k}: " + "; ".join(fails[:4])
        got, want = c.read_pixels(0) - before, z["pixels"][k + 1] - z["pixels"][k]
        assert np.array_equal(got[:, 3], want[:, 3]), f"iteration {k}: splat counts"
        assert np.allclose(got, want, rtol=1e-4, atol=1e-5), f"iteration {k}: splat sums"
    assert flips <= flip_budget, (rays, flips)
    return rays, flips


def test_teapot_resynchronised_iterations_golden():
    """BASELINE.json configs[0] geometry on the wavefront path: 12 iterations, each restarted from the reference kernels' state
    (tests/golden/teapot_resync.npz) -- the exact replacement for a statistical comparison of free-running images."""
    z = _fixture("teapot_resync.npz")
    c = OracleContext(int(z["num_tasks"]), threads=4)
    c.upload_scene(_load_scene(z)); c.set_params(z["params"].view(wire.RENDER_PARAMS).reshape(()))
    rays, flips = resync_check(c, z, flip_budget=1)
    assert rays == 12 * 4096


def test_thread_count_does_not_change_results():
    """The OpenMP oracle must reproduce the sequential (canonical) order for any thread count."""
    d = common.mixed_material_scene()
    w, h, n = 48, 32, 2048
    p = common.scene_params(d, w, h, maxBounces=5, useEnvMap=1, wfSeparateQueues=1)
    e = host.synthetic_sky(32, 16)
    outs = []
    for thr in (1, 5):
        c = OracleContext(n, threads=thr)
        c.upload_scene(d); c.upload_envmap(e); c.set_params(p); driver.reset_renderer(c)
        cn = [driver.benchmark_iteration(c, w * h) for _ in range(10)]
        outs.append((np.stack(cn), c.state_export(), np.stack([c.queue_read(q) for q in range(8)])))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert not common.state_diff(outs[0][1], outs[1][1], 0.0, 0.0)


def test_microkernel_teapot_16spp_golden():
    """BASELINE.json configs[0] ("teapot.ply, 4 bounces, Lambertian only, 16 spp") on the microkernel path, oracle vs the
    reference's own microkernels (tests/golden/mk_teapot.npz)."""
    z = _fixture("mk_teapot.npz")
    d = _load_scene(z)
    p = z["params"].view(wire.RENDER_PARAMS).reshape(())
    c = OracleContext(int(z["num_tasks"]), threads=4)
    c.upload_scene(d)
    driver.render_single(c, p, int(z["spp"]))
    px, ref = c.read_pixels(0), z["pixels"]
    assert np.array_equal(px[:, 3], ref[:, 3]) and (px[:, 3] == 16).all()
    # one path per pixel, no cursor: a flipped ray can only touch its own pixel.  Measured: every one of the 16 384 pixels within 1e-3,
    # ONE pixel beyond 1e-4 (libm vs flx_math in a 16-sample sum) -- asserted as such
    assert np.isclose(px[:, :3], ref[:, :3], rtol=1e-3, atol=1e-3).all()
    assert (~np.isclose(px[:, :3], ref[:, :3], rtol=1e-4, atol=1e-4).all(1)).sum() <= 2
    assert abs(px[:, :3].mean() - ref[:, :3].mean()) <= 1e-5 * ref[:, :3].mean()
    assert np.array_equal(c.mk_stats()[[0, 3]], z["stats"][[0, 3]])
