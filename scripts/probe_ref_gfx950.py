"""Diagnostics for oracle/ref_gpu.py on the GPU box: python scripts/probe_ref_gfx950.py <opencl|hip> <hipfirst 0|1>"""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def say(*a):
    print(*a, file=sys.stderr, flush=True)


backend, hipfirst = sys.argv[1], int(sys.argv[2])
import common
from fluctus_amd import driver
from oracle import ref_gpu
d = common.simple_scene()
w, h, n = 40, 30, 2048
p = common.scene_params(d, w, h, maxBounces=6)
g = None
if hipfirst:
    from fluctus_amd.device import HipContext
    g = HipContext(n); g.upload_scene(d); g.set_params(p); driver.reset_renderer(g)
    say("hip context up")
r = ref_gpu.RefGpuContext(n, backend_name=backend)
say("ref ctx created on", r.B.device_name())
r.upload_scene(d); say("scene up")
r.set_params(p); say("params up")
r.wf_reset(); r.finish(); say("reset ran")
c = r.get_counters(); say("counters", c)
s = r.state_export(); say("state", s.shape, float(s[16].mean()))
r.wf_raygen(); r.finish(); say("raygen ran", r.get_counters())
r.wf_extend(); r.finish(); say("extend ran", r.get_counters())
s = r.state_export(); say("hits", int((s.view(np.int32)[61] >= 0).sum()))
r.clear_queues()
try:
    r.wf_logic(False); r.finish(); say("logic ran", r.get_counters())
except Exception as e:
    say("logic:", type(e).__name__, e)
r.wf_materials(); r.finish(); say("materials ran", r.get_counters())
r.wf_shadow(); r.finish(); say("shadow ran")
say("DONE")
