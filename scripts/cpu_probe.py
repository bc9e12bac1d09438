"""Probe how many host cores are really usable (cgroup quota vs nproc) and how the oracle scales."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
from fluctus_amd import host, wire, driver
from oracle.binding import OracleContext
d = host.generate_scene("kitchen", 60000, 42); host.build_bvh(d, "binned")
W, H = 640, 360
p = wire.default_params(W, H, d.world_radius, d.tris.size)
wire.look_at(p, (0.3, 1.5, 4.4), (0.0, 0.9, -0.5)); p["maxBounces"] = 8; p["useEnvMap"] = 1; p["useAreaLight"] = 0; p["wfSeparateQueues"] = 1
e = host.synthetic_sky(256, 128)
for thr in [1, 4, 8, 16, 32, 64, 128, 256]:
    if thr > (os.cpu_count() or 1): break
    c = OracleContext(1 << 16, threads=thr); c.upload_scene(d); c.upload_envmap(e); c.set_params(p); driver.reset_renderer(c)
    for _ in range(4): driver.benchmark_iteration(c, W * H)
    t = time.perf_counter(); rays = 0
    for _ in range(6):
        cnt = driver.benchmark_iteration(c, W * H); rays += int(cnt[1]) + int(cnt[2])
    dt = time.perf_counter() - t
    print("threads", thr, "Mrays/s %.3f" % (rays / dt / 1e6), flush=True)
    c.close()
