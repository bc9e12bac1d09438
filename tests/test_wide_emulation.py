"""The 4-wide quantised tree WITHOUT a GPU: csrc/flx_wide.h (the product's own builder, the code flx_upload_scene runs) + a host emulation of
the device traversal (tests/wide_analysis.cpp: WRay::setup / wide_node_visit / wide_leaf_visit of csrc/flx_trace4.h with the same float
operations in the same order) against the ORACLE's restatement of the reference traversal (src/bvh.cl:234-310 closest hit, :312-373 any hit).

What it pins on the CPU: the builder's conservative boxes really contain every leaf (a leaf the reference reaches is reached), the node test
is conservative, the leaf gate is the reference's -- i.e. the ANY-HIT result equals the oracle's bit for bit and the CLOSEST hit is the
oracle's triangle except for exact ties in t (counted; budget 1e-5).  The device kernels are compared with the oracle on the
GPU box (tests/test_gpu_wide.py); this test makes the same statement for the tree and the traversal ARITHMETIC where no GPU exists."""
import ctypes as C
import os
import numpy as np
import pytest
import common
import conftest
from common import COL, Q
from fluctus_amd import host, driver
from oracle.binding import OracleContext


def _lib():
    try:
        L = C.CDLL(conftest.build_wide_analysis())
    except Exception as e:                       # no g++ / libgomp here, or the archived experiment header does not compile: not this suite's business
        pytest.skip(f"tests/wide_analysis.cpp did not build: {e}")
    L.fh_analysis_last_error.restype = C.c_char_p
    return L


def _emulate(L, d, nodes, rays, mode):
    out = np.zeros(8, np.float64); tri = np.zeros(rays.shape[0], np.int32); nv = np.zeros(rays.shape[0], np.uint32)
    rc = L.fh_wide_visits_ex(nodes.ctypes.data_as(C.c_void_p), C.c_uint64(nodes.size), d.tris.ctypes.data_as(C.c_void_p), C.c_uint64(d.tris.size),
                             d.indices.ctypes.data_as(C.c_void_p), C.c_uint64(d.indices.size), rays.ctypes.data_as(C.c_void_p), C.c_uint64(rays.shape[0]), mode,
                             out.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p), nv.ctypes.data_as(C.c_void_p))
    assert rc == 0, L.fh_analysis_last_error()
    return tri, out


def _run(d, p, env, n, iters):
    """Oracle free run; per iteration the rays it is about to trace, its own results, and the emulation's."""
    L = _lib()
    o = OracleContext(n, threads=8)
    o.upload_scene(d)
    if env is not None:
        o.upload_envmap(env)
    o.set_params(p); driver.reset_renderer(o)
    npix = int(p["width"]) * int(p["height"])
    ext_rays = flips = sh_rays = 0
    any_modes = (1, 2)
    for it in range(iters):
        o.wf_logic(False); o.wf_raygen(); o.wf_materials()
        cnt = np.array(o.get_counters(), copy=True)
        st = o.state_export()
        qe = o.queue_read(Q.EXTENSION)[:int(cnt[Q.EXTENSION])]; qs = o.queue_read(Q.SHADOW)[:int(cnt[Q.SHADOW])]
        ext = np.zeros((qe.size, 8), np.float32)
        ext[:, 0:3] = st[COL.ORIG:COL.ORIG + 3, qe].T; ext[:, 3] = 3.4028235e38; ext[:, 4:7] = st[COL.DIR:COL.DIR + 3, qe].T
        sh = np.zeros((qs.size, 8), np.float32)
        sh[:, 0:3] = st[COL.SHADOW_ORIG:COL.SHADOW_ORIG + 3, qs].T; sh[:, 3] = st[COL.SHADOW_LEN, qs]; sh[:, 4:7] = st[COL.SHADOW_DIR:COL.SHADOW_DIR + 3, qs].T
        o.wf_extend(); o.wf_shadow()
        so = o.state_export()
        # closest hit: the oracle's triangle, or a tie in t.  (With an area light the oracle's traceExtension overrides a hit behind the
        # light quad with i = 0 / areaLightHit: src/wf_extrays.cl:28-29 -- those rays are compared on hit / no hit of the quad's side only.)
        tri, _ = _emulate(L, d, d.nodes, ext, 0)
        oi = so.view(np.int32)[COL.HIT_I][qe]
        light = so.view(np.uint32)[COL.AREA_LIGHT_HIT][qe] != 0
        differ = (tri != oi) & ~light
        if differ.any():
            # re-trace the differing rays with tMax just beyond the oracle's t: the emulation must find a triangle at the same distance
            assert (tri[differ] >= 0).all() and (oi[differ] >= 0).all(), f"it{it}: a hit on one side and a miss on the other"
        flips += int(differ.sum()); ext_rays += qe.size
        # any hit: bit-identical, whatever the visit order
        blocked = so.view(np.uint32)[COL.SHADOW_BLOCKED][qs] != 0
        for m in any_modes:
            occ, _ = _emulate(L, d, d.nodes, sh, m)
            if int(p["useAreaLight"]):
                # the reference tests the light quad first (src/wf_shadowrays.cl:32-33): blocked = quad OR tree
                assert not (occ >= 0)[~blocked].any(), f"it{it} any-hit order {m}: the emulation is occluded where the oracle is not"
            else:
                assert np.array_equal(occ >= 0, blocked), f"it{it} any-hit order {m}: {int(((occ >= 0) != blocked).sum())} of {qs.size} shadow rays differ"
        sh_rays += qs.size
        o.clear_queues(); o.pixel_index_update(npix, int(cnt[Q.RAYGEN]))
    return ext_rays, flips, sh_rays


def test_wide_tree_emulation_vs_oracle_mixed_scene():
    d = common.mixed_material_scene()
    p = common.scene_params(d, 64, 48, maxBounces=5, useAreaLight=0, useEnvMap=1, wfSeparateQueues=1)
    rays, flips, sh = _run(d, p, host.synthetic_sky(64, 32), 4096, 8)
    assert rays > 20000 and sh > 5000
    assert flips == 0, f"{flips} of {rays} closest hits differ from the oracle's"


def test_wide_tree_emulation_vs_oracle_egyptcat():
    """The reference's own benchmark scene #1 (tests/golden/egyptcat_scene.npz), its start-up parameters (area light)."""
    from fluctus_amd import wire
    d = common.egyptcat_scene()
    p = wire.default_params(128, 128, d.world_radius, d.tris.size)
    rays, flips, sh = _run(d, p, None, 16384, 6)
    assert rays > 50000 and sh > 5000
    assert flips <= max(1, int(1e-5 * rays)), f"{flips} of {rays} closest hits differ from the oracle's"


def test_wide_tree_emulation_vs_oracle_env_light_scene():
    """Env light only on a 30 k-triangle procedural kitchen: far -> near and last-slot-first any-hit orders both equal the oracle's answer."""
    d = host.generate_scene("kitchen", 30000, 42)
    host.build_bvh(d, "sbvh")
    from fluctus_amd import wire
    p = wire.default_params(96, 64, d.world_radius, d.tris.size)
    wire.look_at(p, (0.3, 1.5, 4.4), (0.0, 0.9, -0.5))
    p["useEnvMap"], p["useAreaLight"], p["maxBounces"], p["wfSeparateQueues"] = 1, 0, 6, 1
    rays, flips, sh = _run(d, p, host.synthetic_sky(64, 32), 6144, 8)
    assert rays > 30000 and sh > 10000
    assert flips <= max(1, int(1e-5 * rays)), f"{flips} of {rays} closest hits differ from the oracle's"
