"""Soak: 150 free-running iterations of the mixed-material scene with every fuse / fuse_set / ext_order combination, three flag sets;
counters per iteration, final state (bit-exact) and framebuffer against the oracle.  GPU + oracle: python scripts/soak_fused.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from fluctus_amd import host, driver
from fluctus_amd.device import HipContext
from oracle.binding import OracleContext
d = common.mixed_material_scene()
w, h, n = 128, 96, 16384 + 101
env = host.synthetic_sky(64, 32)
for flags in (dict(useAreaLight=1, useEnvMap=1, wfSeparateQueues=1), dict(useAreaLight=0, useEnvMap=1, wfSeparateQueues=1, useRoulette=1),
              dict(useAreaLight=1, useEnvMap=0, wfSeparateQueues=0)):
    p = common.scene_params(d, w, h, maxBounces=7, envMapStrength=1.5, **flags)
    o = OracleContext(n, threads=8)
    o.upload_scene(d); o.upload_envmap(env); o.set_params(p); driver.reset_renderer(o)
    ocnt = []
    for it in range(150):
        ocnt.append(driver.benchmark_iteration(o, w * h).copy())
    so, po = o.state_export(), o.read_pixels(0)
    for cfg in ((0, 0, 0), (1, 1, 0), (1, 31, 1), (1, 1, 1), (1, 31, 0)):
        g = HipContext(n)
        g.set_option("extend_tree", 2); g.set_option("fuse", cfg[0])
        g.upload_scene(d); g.upload_envmap(env); g.set_params(p); driver.reset_renderer(g)
        if cfg[0]:
            g.set_option("fuse_set", cfg[1]); g.set_option("ext_order", cfg[2])
        for it in range(150):
            c = driver.benchmark_iteration(g, w * h)
            assert (c == ocnt[it]).all(), (flags, cfg, it, c, ocnt[it])
        fails = common.state_diff(g.state_export(), so, 0.0, 0.0)
        assert not fails, (flags, cfg, fails[:3])
        assert common.fb_close(g.read_pixels(0), po), (flags, cfg)
        print("ok", flags, cfg, flush=True)
