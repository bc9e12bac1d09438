"""CPU experiment behind flx_wide_opt.h: node / leaf / triangle visits per ray of the 4-wide traversal (emulated on the host with the device's
arithmetic: host_capi.cpp fh_wide_visits) on steady-state rays of a bench workload, for the reference topology and after 1..P passes of
subtree reinsertion.  Rays come from the CPU oracle free-running the workload at 65 536 paths.
  python scripts/exp_tree_opt.py [workload] [passes...]"""
import ctypes as C
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from fluctus_amd import host, driver, wire  # noqa: E402
from fluctus_amd.wire import COL, Q  # noqa: E402
from oracle.binding import OracleContext  # noqa: E402


def analysis_lib():
    """tests/_build/libwide_analysis.so (tests/wide_analysis.cpp; built by tests/conftest.py: build_wide_analysis)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    L = C.CDLL(conftest.build_wide_analysis())
    L.fh_analysis_last_error.restype = C.c_char_p
    return L



def steady_rays(d, p, env, n=1 << 16, iters=14):
    c = OracleContext(n, threads=os.cpu_count())
    c.upload_scene(d); c.upload_envmap(env); c.set_params(p); driver.reset_renderer(c)
    npix = int(p["width"]) * int(p["height"])
    for _ in range(iters):
        driver.benchmark_iteration(c, npix)
    c.wf_logic(False); c.wf_raygen(); c.wf_materials()
    cnt = np.array(c.get_counters(), copy=True)
    st = c.state_export()
    qe = c.queue_read(Q.EXTENSION)[:int(cnt[Q.EXTENSION])]
    qs = c.queue_read(Q.SHADOW)[:int(cnt[Q.SHADOW])]
    ext = np.zeros((qe.size, 8), np.float32)
    ext[:, 0:3] = st[COL.ORIG:COL.ORIG + 3, qe].T; ext[:, 3] = 3.4028235e38; ext[:, 4:7] = st[COL.DIR:COL.DIR + 3, qe].T
    sh = np.zeros((qs.size, 8), np.float32)
    sh[:, 0:3] = st[COL.SHADOW_ORIG:COL.SHADOW_ORIG + 3, qs].T; sh[:, 3] = st[COL.SHADOW_LEN, qs]; sh[:, 4:7] = st[COL.SHADOW_DIR:COL.SHADOW_DIR + 3, qs].T
    return ext, sh


def visits(d, nodes, rays, mode):
    L = analysis_lib()
    out = np.zeros(8, np.float64)
    rc = L.fh_wide_visits(nodes.ctypes.data_as(C.c_void_p), C.c_uint64(nodes.size), d.tris.ctypes.data_as(C.c_void_p), C.c_uint64(d.tris.size),
                          d.indices.ctypes.data_as(C.c_void_p), C.c_uint64(d.indices.size), rays.ctypes.data_as(C.c_void_p), C.c_uint64(rays.shape[0]), mode,
                          out.ctypes.data_as(C.c_void_p))
    assert rc == 0, L.fh_analysis_last_error()
    return out


def optimise(nodes, passes):
    L = analysis_lib()
    out = np.zeros_like(nodes); st = np.zeros(8, np.float64)
    t0 = time.time()
    rc = L.fh_wide_optimise(nodes.ctypes.data_as(C.c_void_p), C.c_uint64(nodes.size), passes, out.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p))
    assert rc == 0, L.fh_analysis_last_error()
    return out, st, time.time() - t0


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
    passes = [int(x) for x in sys.argv[2:]] or [0, 1, 2, 4]
    d, p, env = bench.build_workload(name=wl)
    ext, sh = steady_rays(d, p, env)
    any_mode = 2 if (p["useEnvMap"] and not p["useAreaLight"]) else 1
    print(f"{wl}: {d.tris.size} triangles, {d.nodes.size} binary nodes; {ext.shape[0]} extension rays, {sh.shape[0]} shadow rays (any-hit order {any_mode})")
    base = None
    for ps in passes:
        nodes, st, dt = optimise(d.nodes, ps)
        e = visits(d, nodes, ext, 0); s = visits(d, nodes, sh, any_mode)
        ne, ns = ext.shape[0], sh.shape[0]
        row = (e[0] / ne, e[1] / ne, e[3] / ne, s[0] / ns, s[1] / ns, s[3] / ns)
        if base is None:
            base = row
        print(f"passes {ps}: SAH {st[0]:.2f} -> {st[1]:.2f}, depth {int(st[5])} -> {int(st[6])}, moved {int(st[2])}, {st[4] / max(1, st[3]):.0f} steps/search, {dt:.1f} s | wide nodes {int(e[6])} stack {int(e[5])}/{int(s[5])} | "
              f"closest: node {row[0]:.2f} ({row[0] / base[0] - 1:+.1%}) leaf {row[1]:.2f} tri {row[2]:.2f} hits {int(e[4])} chk {e[7]:.0f} | any: node {row[3]:.2f} ({row[3] / base[3] - 1:+.1%}) leaf {row[4]:.2f} tri {row[5]:.2f} occluded {int(s[4])}")


if __name__ == "__main__":
    main()
