"""Where the persistent closest-hit kernel's wave instructions go (lab build -DFLX_LAB_RSTATS of trace4r.hip, selected with FLX_HIP_LIB):
descent rounds and leaf phases of a wave with the number of lanes that take part, triangle-loop iterations with theirs.
  python scripts/build_variants.py rstats:-DFLX_LAB_RSTATS && FLX_HIP_LIB=$PWD/variants/libfluctus_hip_rstats.so python scripts/exp_lane_use.py [workload]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluctus_amd import device, driver  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
d, p, env = bench.build_workload(name=wl)
g = device.HipContext(1 << 22)
g.upload_scene(d); g.upload_envmap(env); g.set_params(p); driver.reset_renderer(g)
for _ in range(24):
    bench.step_async(g)
g.finish(); g.reset_stats(); g.counter_totals(reset=True)
K = 8
for _ in range(K):
    bench.step_async(g)
g.finish()
tot = g.counter_totals(reset=True)
out = np.zeros(24, np.uint64)
g._chk(g.L.flx_trace_stats_get_all(g.h, out.ctypes.data_as(__import__("ctypes").c_void_p)))
rays = float(tot[1])
rounds, lanesN, phases, lanesL, tri_it, lanesT = [float(x) for x in out[18:24]]
print(f"{wl}: {rays / K:.0f} extension rays per launch; per ray: node-visit rounds {rounds / rays * 64:.1f} per wave-of-64-rays "
      f"({lanesN / max(1, rounds):.1f} of 64 lanes visit a node), leaf phases {phases / rays * 64:.1f} ({lanesL / max(1, phases):.1f} lanes on a leaf), "
      f"triangle iterations {tri_it / rays * 64:.1f} ({lanesT / max(1, tri_it):.1f} lanes test a triangle)")
nv, lf, tr = 140.0, 40.0, 55.0
wi = (rounds * nv + phases * lf + tri_it * tr) / rays
print(f"   estimated wave-instructions per ray at ~{nv:.0f} / {lf:.0f} / {tr:.0f} VALU per node visit / leaf entry / triangle test: {wi:.1f}  "
      f"= node {rounds * nv / rays:.1f} + leaf entry {phases * lf / rays:.1f} + triangles {tri_it * tr / rays:.1f}")
