"""Experiment: does the ORDER of the extension queue matter to k_extend4?  One steady-state iteration of the bench workload; the
extension kernel is timed on the queue as built, then on the same rays grouped by direction octant, by octant + origin cell, and
shuffled.  (Re-running the kernel on the same rays repeats the same traversal work; only pathLen moves on.)
usage: python scripts/exp_octant.py [workload]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from fluctus_amd import driver, wire
from fluctus_amd.device import HipContext
from fluctus_amd.wire import COL

name = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
d, p, env = bench.build_workload(None, None, name)
n = 1 << 22
c = HipContext(n)
c.upload_scene(d); c.upload_envmap(env); c.set_params(p)
driver.reset_renderer(c)
npix = int(p["width"]) * int(p["height"])
for _ in range(24):
    driver.benchmark_iteration(c, npix)
c.wf_logic(False); c.wf_raygen(); c.wf_materials()
cnt = np.array(c.get_counters(), copy=False); c.finish()
m = int(cnt[1])
q = c.queue_read(1)[:m].copy()
st = c.state_export()
dirs = st[COL.DIR:COL.DIR + 3, :][:, q]
orig = st[COL.ORIG:COL.ORIG + 3, :][:, q]
octant = ((dirs[0] < 0).astype(np.uint32) | ((dirs[1] < 0).astype(np.uint32) << 1) | ((dirs[2] < 0).astype(np.uint32) << 2))
lo, hi = orig.min(1, keepdims=True), orig.max(1, keepdims=True)
cell = np.clip(((orig - lo) / np.maximum(hi - lo, 1e-9) * 16).astype(np.uint32), 0, 15)
def part1by2(x):
    x = x & 0x3FF; x = (x | (x << 16)) & 0x30000FF; x = (x | (x << 8)) & 0x300F00F; x = (x | (x << 4)) & 0x30C30C3; x = (x | (x << 2)) & 0x9249249
    return x
morton = part1by2(cell[0]) | (part1by2(cell[1]) << 1) | (part1by2(cell[2]) << 2)
rng = np.random.RandomState(1)
orders = {
    "as built": np.arange(m),
    "octant (stable)": np.argsort(octant, kind="stable"),
    "origin cell 16^3 (stable)": np.argsort(morton, kind="stable"),
    "octant + origin cell": np.argsort(octant.astype(np.uint64) << 16 | morton, kind="stable"),
    "origin cell + octant": np.argsort(morton.astype(np.uint64) << 3 | octant, kind="stable"),
    "by path id (all)": np.argsort(q, kind="stable"),
    "regenerated first, then the rest by path id": np.concatenate([np.arange(int(cnt[0])), int(cnt[0]) + np.argsort(q[int(cnt[0]):], kind="stable")]),
    "shuffled": rng.permutation(m),
}
c.set_option("overlap", 0)
for label, o in orders.items():
    c.queue_write(1, q[o])
    c.set_counters(cnt)
    c.profile_reset(); c.profile_enable(3)
    for _ in range(3):
        c.wf_extend()
    c.finish(); c.profile_enable(0)
    ms, k = c.profile_get()["extend"]
    print("%-28s k_extend4 %.3f ms  (%d rays)" % (label, ms / k, m), flush=True)
