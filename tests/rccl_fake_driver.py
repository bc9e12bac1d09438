"""Driver of tests/test_gpu_rccl_fake.py, run in its own process with FLX_RCCL_LIB=tests/_build/libfake_rccl.so (the override is read when
libfluctus_hip.so first binds its RCCL entry points, and other tests of the suite bind the real library).

  python tests/rccl_fake_driver.py <nranks> <root> <mode>
    mode threads : one host thread per rank, each with its own context on device 0: flx_group_unique_id / flx_group_init / flx_gather
                   -- the one-process-per-GPU code path of api.hip (ncclCommInitRank, grouped ncclSend on the peers, ncclRecv on the root)
    mode local   : one thread, flx_group_init_local / flx_gather_local -- the single-process path (ncclCommInitAll, sends and receives
                   of all ranks inside one group)
Prints one JSON line: {"ok": bool, "errors": [...], "counters": [...], "info": [[count, rank], ...], "depth": group depth after the calls}.
The expected image is assembled from flx_read_pixels of every rank: global pixel p * R + r = tile r, pixel p."""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n, root, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import common
    from fluctus_amd import device, driver
    fake = C.CDLL(os.environ["FLX_RCCL_LIB"])
    d = common.mixed_material_scene()
    w, h, tasks = 61, 47, 2048                     # 2867 pixels: not a multiple of 3 or 8 -> ragged last tile
    p = common.scene_params(d, w, h, maxBounces=3, wfSeparateQueues=1)
    ctxs = []
    for r in range(n):
        c = device.HipContext(tasks)
        c.upload_scene(d)
        c.set_params(p)
        ctxs.append(c)
    out = {"ok": False, "errors": [], "info": [], "depth": None}
    images = [None] * n
    errors = [None] * n
    depth = [None] * n

    def render(c):
        driver.reset_renderer(c)
        for _ in range(4 + c_rank[id(c)]):          # a different number of iterations per rank: every tile is different
            driver.benchmark_iteration(c, c.local_pixels())
        c.finish()

    c_rank = {id(c): r for r, c in enumerate(ctxs)}
    if mode == "threads":
        uid = device.group_unique_id()

        def work(r):
            try:
                c = ctxs[r]
                c.group_init(r, n, uid)
                render(c)
                images[r] = c.gather(root)
            except Exception as e:                  # noqa: BLE001
                errors[r] = str(e)
            depth[r] = int(fake.fake_rccl_group_depth())
        th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
        if any(t.is_alive() for t in th):
            out["errors"].append("a rank thread is still blocked after 120 s")
            print(json.dumps(out), flush=True)
            os._exit(2)
        for r in range(n):
            if errors[r] is None:
                try:
                    out["info"].append(list(ctxs[r].group_info()))
                except Exception as e:              # noqa: BLE001  (an aborted communicator)
                    out["info"].append(str(e))
    else:
        try:
            device.group_init_local(ctxs)
            for c in ctxs:
                render(c)
            images[root] = device.gather_local(ctxs, root)
            out["info"] = [list(c.group_info()) for c in ctxs]
        except Exception as e:                      # noqa: BLE001
            errors[root] = str(e)
        depth[root] = int(fake.fake_rccl_group_depth())
    out["errors"] += [f"rank {r}: {e}" for r, e in enumerate(errors) if e]
    out["depth"] = [x for x in depth if x is not None]
    cnt = (C.c_uint64 * 8)()
    fake.fake_rccl_counters(cnt)
    out["counters"] = [int(x) for x in cnt]
    if not out["errors"]:
        full = np.zeros((w * h, 4), np.float32)
        for r, c in enumerate(ctxs):
            tile = c.read_pixels(0)
            assert tile.shape[0] == len(range(r, w * h, n))
            full[r::n] = tile
        got = images[root]
        out["ok"] = bool(got is not None and np.array_equal(got, full) and full[:, 3].sum() > 0)
        out["tiles_differ"] = bool(n == 1 or not np.array_equal(ctxs[0].read_pixels(0)[:8], ctxs[n - 1].read_pixels(0)[:8]))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
