"""Generate tests/golden/*.npz from the REFERENCE's own kernels (oracle/_ref, this container only).

  python scripts/make_golden.py

Fixtures are data: input arrays (scene in wire format, parameters, states) and the outputs the
reference kernels produced for them.  No reference source is stored.
  teapot_wf.npz   config 1 of BASELINE.json (assets/teapot.ply, Lambertian, area light, 4 bounces, 128x128,
                  16384 paths): per-iteration queue counters and the accumulated image after 24 iterations.
  steps_*.npz     all-BSDF scene: path state / queues / counters after every kernel of two consecutive
                  iterations (state k -> kernel -> state k+1 chains), for two flag sets.
  steps_denoiser_* the same with the reference's USE_OPTIX_DENOISER kernel builds: + the feature accumulators after every kernel.
  raygen.npz      genRays outputs (seed stream, pixel index, ray origin/direction) for 4096 paths.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from fluctus_amd import host, wire, driver  # noqa: E402
from oracle.binding import RefContext  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def scene_arrays(d):
    return dict(tris=d.tris.view(np.uint8).reshape(-1), nodes=d.nodes.view(np.uint8).reshape(-1), indices=d.indices,
                materials=d.materials.view(np.uint8).reshape(-1), texdesc=d.texdesc.view(np.uint8).reshape(-1), texdata=d.texdata)


def snapshot(c, aov=False):
    cnt = c.get_counters().copy()
    s = dict(state=c.state_export(), counters=cnt, queues=np.stack([c.queue_read(q) for q in range(8)]))
    if aov:
        s["aov"] = np.stack([c.read_pixels(4), c.read_pixels(5)])          # denoiser feature accumulators: albedo, normal
    return s


def teapot():
    d = host.load_scene("/root/reference/assets/teapot.ply")
    host.build_bvh(d, "sbvh")
    w = h = 128
    n = w * h
    p = wire.default_params(w, h, d.world_radius, d.tris.size)
    p["maxBounces"] = 4
    c = RefContext(n)
    c.upload_scene(d); c.set_params(p); driver.reset_renderer(c)
    cnts = np.stack([driver.benchmark_iteration(c, w * h) for _ in range(24)])
    np.savez_compressed(os.path.join(OUT, "teapot_wf.npz"), num_tasks=n, params=np.asarray(p).reshape(1).view(np.uint8),
                        counters=cnts, pixels=c.read_pixels(0), **scene_arrays(d))
    print("teapot_wf.npz", cnts[-1])


def steps(tag, denoiser=False, scene=None, scene_file=None, w=48, h=32, n=1024, params=None, **flags):
    """scene / scene_file: another scene than the all-BSDF test scene, its triangle / material / texture arrays kept in a fixture of their
    own (tests/golden/<scene_file>) instead of being stored again; params "reference": the reference's start-up parameters untouched."""
    d = scene if scene is not None else common.mixed_material_scene()
    if params == "reference":
        p = wire.default_params(w, h, d.world_radius, d.tris.size)
    else:
        p = common.scene_params(d, w, h, maxBounces=5, envMapStrength=1.5, **flags)
    e = host.synthetic_sky(64, 32)
    c = RefContext(n)
    if denoiser:
        c.set_option("denoiser", 1)                                         # the -DUSE_OPTIX_DENOISER builds of logic / process
    c.upload_scene(d); c.upload_envmap(e); c.set_params(p); driver.reset_renderer(c)
    cursor = 0                                                              # the reference's host-side pixel cursor (src/clcontext.cpp:891-895)
    for _ in range(6):
        cursor = (cursor + int(driver.benchmark_iteration(c, w * h)[0])) % (w * h)
    cursors = [cursor]
    _snap = snapshot
    snapshot_ = lambda ctx: _snap(ctx, aov=denoiser)
    snaps, names = [snapshot_(c)], ["start"]
    for it in range(2):
        for name, fn in (("logic", lambda: c.wf_logic(False)), ("raygen", c.wf_raygen), ("materials", c.wf_materials),
                         ("extend", c.wf_extend), ("shadow", c.wf_shadow)):
            fn()
            snaps.append(snapshot_(c)); names.append(name); cursors.append(cursor)
        cnt = c.get_counters().copy()
        c.clear_queues(); c.pixel_index_update(w * h, int(cnt[0]))
        cursor = (cursor + int(cnt[0])) % (w * h)
        snaps.append(snapshot_(c)); names.append("end"); cursors.append(cursor)
    extra = dict(aov=np.stack([s["aov"] for s in snaps])) if denoiser else {}
    np.savez_compressed(os.path.join(OUT, f"steps_{tag}.npz"), num_tasks=n, **extra, params=np.asarray(p).reshape(1).view(np.uint8),
                        names=np.array(names), states=np.stack([s["state"] for s in snaps]),
                        counters=np.stack([s["counters"] for s in snaps]), queues=np.stack([s["queues"] for s in snaps]),
                        env_rgb=e.rgb, env_prob=e.prob, env_alias=e.alias, env_pdf=e.pdf, env_wh=np.array([e.w, e.h]),
                        pixel_cursor=np.array(cursors, np.uint32),                         # cursor in effect when snapshot k was taken
                        **(dict(scene_file=np.array(scene_file), nodes=d.nodes.view(np.uint8).reshape(-1), indices=d.indices) if scene_file else scene_arrays(d)))
    print(f"steps_{tag}.npz", names)


def raygen():
    d = common.simple_scene()
    w, h, n = 64, 64, 4096
    p = common.scene_params(d, w, h)
    p["camera"]["apertureSize"], p["camera"]["focalDist"] = 0.02, 2.5      # exercise the thin-lens branch
    c = RefContext(n)
    c.upload_scene(d); c.set_params(p)
    c.pixel_index_reset(); c.wf_reset(); c.wf_raygen()
    st = c.state_export()
    np.savez_compressed(os.path.join(OUT, "raygen.npz"), num_tasks=n, params=np.asarray(p).reshape(1).view(np.uint8), state=st,
                        ext_queue=c.queue_read(1), counters=c.get_counters().copy())
    print("raygen.npz")


def teapot_resync():
    """BASELINE.json configs[0] geometry (teapot.ply, Lambertian, area light, 4 bounces) on the wavefront path, 64x64, 4096 paths:
    the reference's full state BEFORE each of 12 consecutive iterations + the framebuffer after each.  A test restarts every
    iteration from the reference's state, so hit-index flips cannot fork the runs and every iteration is compared exactly."""
    d = host.load_scene("/root/reference/assets/teapot.ply")
    host.build_bvh(d, "sbvh")
    w = h = 64
    n = 4096
    p = wire.default_params(w, h, d.world_radius, d.tris.size)
    p["maxBounces"] = 4
    c = RefContext(n)
    c.upload_scene(d); c.set_params(p); driver.reset_renderer(c)
    cursor = 0
    for _ in range(4):                                                      # past the all-primary start
        cursor = (cursor + int(driver.benchmark_iteration(c, w * h)[0])) % (w * h)
    states, cursors, pixels, counters = [c.state_export()], [cursor], [c.read_pixels(0)], []
    for _ in range(12):
        cnt = driver.benchmark_iteration(c, w * h)
        cursor = (cursor + int(cnt[0])) % (w * h)
        counters.append(cnt); states.append(c.state_export()); cursors.append(cursor); pixels.append(c.read_pixels(0))
    np.savez_compressed(os.path.join(OUT, "teapot_resync.npz"), num_tasks=n, params=np.asarray(p).reshape(1).view(np.uint8),
                        states=np.stack(states), pixel_cursor=np.array(cursors, np.uint32), pixels=np.stack(pixels),
                        counters=np.stack(counters), **scene_arrays(d))
    print("teapot_resync.npz")


def mk_teapot():
    """BASELINE.json configs[0] on the microkernel path: teapot.ply, 128x128, 4 bounces, Lambertian, 16 spp."""
    d = host.load_scene("/root/reference/assets/teapot.ply")
    host.build_bvh(d, "sbvh")
    w = h = 128
    p = wire.default_params(w, h, d.world_radius, d.tris.size)
    p["maxBounces"] = 4
    c = RefContext(w * h)
    c.upload_scene(d)
    driver.render_single(c, p, 16)
    np.savez_compressed(os.path.join(OUT, "mk_teapot.npz"), num_tasks=w * h, spp=16, params=np.asarray(p).reshape(1).view(np.uint8),
                        pixels=c.read_pixels(0), stats=c.mk_stats(), **scene_arrays(d))
    print("mk_teapot.npz", c.mk_stats())


if __name__ == "__main__":
    mk_teapot()
    teapot()
    teapot_resync()
    steps("area_sep", useAreaLight=1, useEnvMap=0, wfSeparateQueues=1)
    steps("env_area_single_rr", useAreaLight=1, useEnvMap=1, wfSeparateQueues=0, useRoulette=1)
    steps("denoiser_env_area_sep", denoiser=True, useAreaLight=1, useEnvMap=1, wfSeparateQueues=1)
    raygen()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
