"""How often would a per-path "last occluder" cache hit?  (round 3 verdict, next #7: test the triangle that blocked the path's previous shadow
ray first; any-hit is order-free, so shadowRayBlocked stays identical.)  The CPU oracle free-runs a workload at 65 536 paths; every iteration's
shadow rays are traced by the host emulation of the device's any-hit traversal (fh_wide_visits_ex), which also returns the occluder it met and
the wide-node visits of every ray; the cached triangle of the ray's path is tested first (numpy Moller-Trumbore).
  python scripts/exp_occluder_cache.py [workload] [iterations]"""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from fluctus_amd import host, driver  # noqa: E402
from fluctus_amd.wire import COL, Q  # noqa: E402
from oracle.binding import OracleContext  # noqa: E402


def analysis_lib():
    """tests/_build/libwide_analysis.so (tests/wide_analysis.cpp; built by tests/conftest.py: build_wide_analysis)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    L = C.CDLL(conftest.build_wide_analysis())
    L.fh_analysis_last_error.restype = C.c_char_p
    return L



def mt_hit(o, d, tmax, p0, p1, p2):
    s1, s2 = p1 - p0, p2 - p0
    pv = np.cross(d, s2); det = (s1 * pv).sum(1)
    ok = np.abs(det) >= 1e-12
    idet = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tv = o - p0
    u = (tv * pv).sum(1) * idet
    qv = np.cross(tv, s1)
    v = (d * qv).sum(1) * idet
    t = (s2 * qv).sum(1) * idet
    return ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0) & (t < tmax)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    d, p, env = bench.build_workload(name=wl)
    n = 1 << 16
    c = OracleContext(n, threads=os.cpu_count())
    c.upload_scene(d); c.upload_envmap(env); c.set_params(p); driver.reset_renderer(c)
    npix = int(p["width"]) * int(p["height"])
    P = np.stack([np.stack([d.tris[v]["p"][k] for k in "xyz"], -1) for v in ("v0", "v1", "v2")], 1).astype(np.float32)      # ntris x 3 x 3
    L = analysis_lib()
    mode = 2 if (p["useEnvMap"] and not p["useAreaLight"]) else 1
    cache = np.full(n, -1, np.int64)
    tot = dict(rays=0, occluded=0, cached=0, cache_hit=0, visits=0.0, visits_saved=0.0)
    for it in range(iters):
        c.wf_logic(False); c.wf_raygen(); c.wf_materials()
        cnt = np.array(c.get_counters(), copy=True)
        st = c.state_export()
        qs = c.queue_read(Q.SHADOW)[:int(cnt[Q.SHADOW])]
        regen = c.queue_read(Q.RAYGEN)[:int(cnt[Q.RAYGEN])]
        cache[regen] = -1                                  # a regenerated path starts without a cached occluder (it could keep it: same slot, other pixel)
        rays = np.zeros((qs.size, 8), np.float32)
        rays[:, 0:3] = st[COL.SHADOW_ORIG:COL.SHADOW_ORIG + 3, qs].T; rays[:, 3] = st[COL.SHADOW_LEN, qs]; rays[:, 4:7] = st[COL.SHADOW_DIR:COL.SHADOW_DIR + 3, qs].T
        out = np.zeros(8); tri = np.zeros(qs.size, np.int32); nvis = np.zeros(qs.size, np.uint32)
        rc = L.fh_wide_visits_ex(d.nodes.ctypes.data_as(C.c_void_p), C.c_uint64(d.nodes.size), d.tris.ctypes.data_as(C.c_void_p), C.c_uint64(d.tris.size),
                                 d.indices.ctypes.data_as(C.c_void_p), C.c_uint64(d.indices.size), rays.ctypes.data_as(C.c_void_p), C.c_uint64(qs.size), mode,
                                 out.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p), nvis.ctypes.data_as(C.c_void_p))
        assert rc == 0
        if it >= 8:                                        # steady state only
            have = cache[qs] >= 0
            ct = np.where(have, cache[qs], 0)
            hit = have & mt_hit(rays[:, 0:3].astype(np.float64), rays[:, 4:7].astype(np.float64), rays[:, 3].astype(np.float64),
                                P[ct, 0].astype(np.float64), P[ct, 1].astype(np.float64), P[ct, 2].astype(np.float64))
            tot["rays"] += qs.size; tot["occluded"] += int((tri >= 0).sum()); tot["cached"] += int(have.sum()); tot["cache_hit"] += int(hit.sum())
            tot["visits"] += float(nvis.sum()); tot["visits_saved"] += float(nvis[hit].sum())
        occ = tri >= 0
        cache[qs[occ]] = tri[occ]
        c.wf_extend(); c.wf_shadow(); c.clear_queues(); c.pixel_index_update(npix, int(cnt[Q.RAYGEN]))
    r = tot["rays"]
    print(f"{wl}: {r} shadow rays over {iters - 8} steady-state iterations: occluded {tot['occluded'] / r:.1%}; rays whose path has a cached occluder {tot['cached'] / r:.1%}; "
          f"cache hits {tot['cache_hit'] / r:.1%} of all rays = {tot['cache_hit'] / max(1, tot['occluded']):.1%} of the occluded ones; wide-node visits those rays would skip: "
          f"{tot['visits_saved'] / max(1.0, tot['visits']):.1%} of all any-hit node visits ({tot['visits'] / r:.2f} per ray)")


if __name__ == "__main__":
    main()
