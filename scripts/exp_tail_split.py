"""What would TAIL SPLITTING buy the thread-per-ray any-hit kernel?  (round 4 verdict, next #8.)  A wave of k_shadow4 lives as long as its
longest ray (20 of 64 lanes busy on the kitchen).  Variant: a ray that has made K node visits suspends -- stack + position to the spill area, its
queue index to a continuation list -- and a second (third, ...) launch finishes the suspended rays, compacted.  Model, from the per-ray wide-node
visit counts of the host emulation of the device traversal (tests/wide_analysis.cpp) on the oracle's steady-state shadow rays, rays in queue order,
64 consecutive rays = one wave, cost of a wave = the node-visit rounds it runs = max over its lanes (capped at the pass's budget):
    python scripts/exp_tail_split.py [workload] [log2 paths] [iterations]"""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from fluctus_amd import driver  # noqa: E402
from fluctus_amd.wire import COL, Q  # noqa: E402
from oracle.binding import OracleContext  # noqa: E402
from exp_occluder_cache import analysis_lib  # noqa: E402


def rounds(nv):
    """wave-rounds of a thread-per-ray launch over rays with nv visits each (queue order)"""
    pad = (-nv.size) % 64
    a = np.concatenate([nv, np.zeros(pad, nv.dtype)]).reshape(-1, 64)
    return int(a.max(1).sum()), a.shape[0]


def split(nv, budgets):
    rem = nv.astype(np.int64)
    total = 0; waves = 0; detail = []
    for j, K in enumerate(budgets + [None]):
        if rem.size == 0:
            break
        capped = rem if K is None else np.minimum(rem, K)
        r, w = rounds(capped)
        total += r; waves += w
        detail.append((K, rem.size, r))
        if K is None:
            break
        rem = rem[rem > K] - K
    return total, waves, detail


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
    n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    d, p, env = bench.build_workload(name=wl)
    c = OracleContext(n, threads=os.cpu_count())
    c.upload_scene(d); c.upload_envmap(env); c.set_params(p); driver.reset_renderer(c)
    npix = int(p["width"]) * int(p["height"])
    L = analysis_lib()
    mode = 2 if (p["useEnvMap"] and not p["useAreaLight"]) else 1
    for it in range(iters):
        c.wf_logic(False); c.wf_raygen(); c.wf_materials()
        cnt = np.array(c.get_counters(), copy=True)
        if it == iters - 1:
            st = c.state_export()
            qs = c.queue_read(Q.SHADOW)[:int(cnt[Q.SHADOW])]
            rays = np.zeros((qs.size, 8), np.float32)
            rays[:, 0:3] = st[COL.SHADOW_ORIG:COL.SHADOW_ORIG + 3, qs].T; rays[:, 3] = st[COL.SHADOW_LEN, qs]; rays[:, 4:7] = st[COL.SHADOW_DIR:COL.SHADOW_DIR + 3, qs].T
            out = np.zeros(8); tri = np.zeros(qs.size, np.int32); nv = np.zeros(qs.size, np.uint32)
            rc = L.fh_wide_visits_ex(d.nodes.ctypes.data_as(C.c_void_p), C.c_uint64(d.nodes.size), d.tris.ctypes.data_as(C.c_void_p), C.c_uint64(d.tris.size),
                                     d.indices.ctypes.data_as(C.c_void_p), C.c_uint64(d.indices.size), rays.ctypes.data_as(C.c_void_p), C.c_uint64(qs.size), mode,
                                     out.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p), nv.ctypes.data_as(C.c_void_p))
            assert rc == 0
        c.wf_extend(); c.wf_shadow(); c.clear_queues(); c.pixel_index_update(npix, int(cnt[Q.RAYGEN]))
    base, bw = rounds(nv)
    print(f"{wl}: {nv.size} shadow rays of one steady-state iteration at {n} paths; wide-node visits per ray: mean {nv.mean():.2f}, median {np.median(nv):.0f}, "
          f"p85 {np.percentile(nv, 85):.0f}, p95 {np.percentile(nv, 95):.0f}, p99 {np.percentile(nv, 99):.0f}, max {nv.max()}")
    print(f"thread-per-ray: {base} wave-rounds in {bw} waves = {base / bw:.1f} rounds per wave; lanes busy {nv.sum() / (64.0 * base):.1%} (device counters: 20 of 64 = 31 %)")
    for budgets in ([8], [12], [16], [24], [8, 8], [12, 12], [8, 16], [12, 24], [8, 8, 16], [6, 6, 12, 24]):
        t, w, det = split(nv, list(budgets))
        print(f"  budgets {str(budgets):16s}: {t} wave-rounds ({t / base - 1.0:+.1%}), {w} waves; lanes busy {nv.sum() / (64.0 * t):.1%}; passes: "
              + ", ".join(f"[K={k} rays={r} rounds={x}]" for k, r, x in det))


if __name__ == "__main__":
    main()
