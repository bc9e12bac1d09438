#!/usr/bin/env python3
"""Upper bound of what RAY ORDER in the extension / shadow queue is worth to the traversal kernels (round 3 experiment).

The reference's queues hold paths in whatever order its atomics hand out; any order is a valid queue.  This script takes a steady-state
iteration of a bench workload, rewrites the extension (and shadow) queue on the HOST in several orders -- as produced, path id, random,
origin Morton code, direction octant + origin Morton code -- and times the shipped traversal kernel alone on each (HIP events of the
profile hooks, serial schedule, 10 launches each; a re-launch traces the same rays).  The host-side sort is free here: the numbers bound
what an on-device sort could win BEFORE its own cost.

usage: python scripts/exp_ray_order.py [workload] [num_tasks]
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from fluctus_amd.device import HipContext       # noqa: E402
from fluctus_amd import driver                  # noqa: E402
from fluctus_amd.wire import COL, Q             # noqa: E402


def part1by2(v):
    v = v.astype(np.uint64) & 0x3FF
    v = (v | (v << 16)) & 0x30000FF
    v = (v | (v << 8)) & 0x300F00F
    v = (v | (v << 4)) & 0x30C30C3
    v = (v | (v << 2)) & 0x9249249
    return v


def morton(o, lo, hi):
    q = np.clip((o - lo) / np.maximum(hi - lo, 1e-20) * 1024.0, 0, 1023).astype(np.uint32)
    return part1by2(q[0]) | (part1by2(q[1]) << 1) | (part1by2(q[2]) << 2)


def octant(d):
    return ((d[0] < 0).astype(np.uint64) | ((d[1] < 0).astype(np.uint64) << 1) | ((d[2] < 0).astype(np.uint64) << 2))


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
    d, p, env = bench.build_workload(name=workload)
    npix = int(p["width"]) * int(p["height"])
    g = HipContext(n)
    g.set_option("overlap", 0)
    g.upload_scene(d); g.upload_envmap(env); g.set_params(p); driver.reset_renderer(g)
    for _ in range(30):
        driver.benchmark_iteration(g, npix)
    g.wf_logic(False); g.wf_raygen(); g.wf_materials()
    cnt = g.get_counters(); g.finish()                 # (the buffer is filled asynchronously: valid after finish)
    cnt = np.array(cnt, copy=True)
    st = g.state_export()
    rng = np.random.default_rng(1)

    def orders(qid, ocol, dcol):
        ne = int(cnt[qid])
        q0 = g.queue_read(qid)[:ne].copy()
        o, dd = st[ocol:ocol + 3][:, q0], st[dcol:dcol + 3][:, q0]
        fin = np.isfinite(o).all(axis=0)
        lo, hi = o[:, fin].min(axis=1, keepdims=True), o[:, fin].max(axis=1, keepdims=True)
        m = morton(np.nan_to_num(o), lo, hi)
        oc = octant(dd)
        dq = np.clip((dd * 0.5 + 0.5) * 4.0, 0, 3).astype(np.uint64)          # 2 bits per direction component
        dkey = dq[0] | (dq[1] << 2) | (dq[2] << 4)
        return q0, {
            "as produced": q0,
            "path id": np.sort(q0),
            "random": rng.permutation(q0),
            "origin morton (30 bit)": q0[np.argsort(m, kind="stable")],
            "origin morton (15 bit)": q0[np.argsort(m >> 15, kind="stable")],
            "octant, origin morton": q0[np.argsort((oc << 30) | m, kind="stable")],
            "origin morton (15 bit), direction 6 bit": q0[np.argsort(((m >> 15) << 6) | dkey, kind="stable")],
            "origin morton (30 bit) in 64 K-ray chunks": np.concatenate([c[np.argsort(morton(np.nan_to_num(st[ocol:ocol + 3][:, c]), lo, hi), kind="stable")]
                                                                         for c in np.array_split(q0, max(1, ne // 65536))]),
        }

    def time_kernel(which, launch):
        g.profile_reset(); g.profile_enable(1)
        for _ in range(10):
            launch()
        g.finish(); g.profile_enable(0)
        ms, k = g.profile_get()[which]
        return ms / max(1, k)

    print(f"{workload}: {n} paths, extension queue {int(cnt[Q.EXTENSION])}, shadow queue {int(cnt[Q.SHADOW])}", flush=True)
    q0, od = orders(Q.EXTENSION, COL.ORIG, COL.DIR)
    for name, q in od.items():
        g.queue_write(Q.EXTENSION, q)
        t = time_kernel("extend", g.wf_extend)
        print(f"  extension  {name:48s} {t:7.3f} ms", flush=True)
    g.queue_write(Q.EXTENSION, q0)
    g.wf_extend()
    q0, od = orders(Q.SHADOW, COL.SHADOW_ORIG, COL.SHADOW_DIR)
    for name, q in od.items():
        g.queue_write(Q.SHADOW, q)
        t = time_kernel("shadow", g.wf_shadow)
        print(f"  shadow     {name:48s} {t:7.3f} ms", flush=True)
    g.close()


if __name__ == "__main__":
    main()
