"""Throughput of the microkernel integrator (renderSingle passes) and cost of the denoiser feature buffers, kitchen-proc 1080p."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime first)
import bench
from fluctus_amd import device, driver

d, p, env = bench.build_workload()
npix = int(p["width"]) * int(p["height"])
for den in (0, 1):
    g = device.HipContext(npix)
    g.set_option("denoiser", den)
    g.upload_scene(d); g.upload_envmap(env)
    q = p.copy(); q["useRoulette"] = 0
    g.set_params(q); g.mk_reset(); g.mk_stats(reset=True)
    for _ in range(2):
        driver.render_single_pass(g, q["maxBounces"])
    g.finish(); g.mk_stats(reset=True)
    t0 = time.perf_counter(); passes = 8
    for _ in range(passes):
        driver.render_single_pass(g, q["maxBounces"])
    g.finish(); dt = time.perf_counter() - t0
    st = g.mk_stats()
    rays = float(st[0]) + float(st[1]) + float(st[2])
    print(f"MK denoiser={den}: {passes} spp in {dt*1e3:.1f} ms  {rays/dt/1e6:.0f} Mrays/s (primary+extension+shadow)  {st[3]/dt/1e6:.1f} Msamples/s")
    g.close()
# wavefront path with the feature buffers on
for den in (0, 1):
    g = device.HipContext(1 << 20)
    g.set_option("denoiser", den)
    g.upload_scene(d); g.upload_envmap(env); g.set_params(p); driver.reset_renderer(g)
    for _ in range(24):
        bench.step_async(g)
    g.finish(); g.counter_totals(reset=True)
    t0 = time.perf_counter()
    for _ in range(80):
        bench.step_async(g)
    g.finish(); dt = time.perf_counter() - t0
    tot = g.counter_totals(reset=True)
    print(f"WF denoiser={den}: {(float(tot[1])+float(tot[2]))/dt/1e6:.0f} Mrays/s  {dt/80*1e3:.3f} ms/step")
    g.close()
