"""Soak of the fused logic + material pass: 150 free-running iterations of the mixed-material scene with every fuse / fuse_set /
ext_order combination -- counters per iteration, final state (bit-exact) and framebuffer against the oracle."""
import numpy as np
import pytest
import common
from fluctus_amd import host, driver

pytestmark = pytest.mark.gpu

ITERS = 150
CONFIGS = [(0, 0, 0), (1, 1, 0), (1, 31, 1), (1, 1, 1), (1, 31, 0), (1, 31, 2), (1, 1, 2)]          # (fuse, fuse_set, ext_order)


@pytest.mark.parametrize("flags", [dict(useAreaLight=1, useEnvMap=1, wfSeparateQueues=1),
                                   dict(useAreaLight=0, useEnvMap=1, wfSeparateQueues=1, useRoulette=1),
                                   dict(useAreaLight=1, useEnvMap=0, wfSeparateQueues=0)],
                         ids=["area+env", "env-roulette", "area-single-queue"])
def test_fused_configurations_free_running(flags):
    from fluctus_amd.device import HipContext
    from oracle.binding import OracleContext
    d = common.mixed_material_scene()
    w, h, n = 128, 96, 16384 + 101                                           # several paths per pixel, not a multiple of any block size
    env = host.synthetic_sky(64, 32)
    p = common.scene_params(d, w, h, maxBounces=7, envMapStrength=1.5, **flags)
    o = OracleContext(n, threads=8)
    o.upload_scene(d); o.upload_envmap(env); o.set_params(p); driver.reset_renderer(o)
    ocnt = [driver.benchmark_iteration(o, w * h).copy() for _ in range(ITERS)]
    so, po = o.state_export(), o.read_pixels(0)
    for fuse, fuse_set, ext_order in CONFIGS:
        g = HipContext(n)
        g.set_option("extend_tree", 2)                                       # the bit-exact closest hit
        g.set_option("fuse", fuse)
        g.upload_scene(d); g.upload_envmap(env); g.set_params(p); driver.reset_renderer(g)
        if fuse:
            g.set_option("fuse_set", fuse_set); g.set_option("ext_order", ext_order)
        for it in range(ITERS):
            c = driver.benchmark_iteration(g, w * h)
            assert (c == ocnt[it]).all(), f"{(fuse, fuse_set, ext_order)} iteration {it}: counters {c} vs {ocnt[it]}"
        fails = common.state_diff(g.state_export(), so, 0.0, 0.0)
        assert not fails, f"{(fuse, fuse_set, ext_order)}: " + "; ".join(fails[:3])
        assert common.fb_close(g.read_pixels(0), po), (fuse, fuse_set, ext_order)
