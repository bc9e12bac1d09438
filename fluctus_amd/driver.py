"""Driver loops shared by tests and bench.py: they sequence the boundary calls exactly like the
reference's Tracer (reference: src/tracer.cpp:222-266 update() WF branch, :372-382 resetRenderer,
:431-470 runBenchmark body).  `ctx` is any object with the HipContext method set."""
import numpy as np


def reset_renderer(ctx):
    """runBenchmark's resetRenderer: updateParams happened already; reset + clear (src/tracer.cpp:372-382)."""
    ctx.pixel_index_reset()
    ctx.wf_reset()
    ctx.clear_queues()
    ctx.finish()


def benchmark_iteration(ctx, npix):
    """One pass of the runBenchmark WF body (src/tracer.cpp:433-439, :456-462). Returns the counters."""
    ctx.wf_logic(False)
    ctx.wf_raygen()
    ctx.wf_materials()
    cnt = ctx.get_counters()          # valid after finish() (async on the device path)
    ctx.wf_extend()
    ctx.wf_shadow()
    ctx.clear_queues()
    ctx.finish()
    cnt = np.array(cnt, copy=True)
    ctx.pixel_index_update(npix, int(cnt[0]))
    return cnt


def first_frame(ctx, params, npix):
    """Tracer::update with iteration == 0 (src/tracer.cpp:228-266): 2-bounce preview, N = 3."""
    p2 = params.copy()
    p2["maxBounces"] = min(2, int(params["maxBounces"]))
    ctx.set_params(p2)
    ctx.pixel_index_reset()
    ctx.wf_reset()
    ctx.wf_raygen()
    ctx.wf_extend()
    ctx.clear_queues()
    cnt = None
    for _ in range(3):
        ctx.wf_logic(True)
        ctx.wf_raygen()
        ctx.wf_materials()
        cnt = ctx.get_counters()
        ctx.wf_extend()
        ctx.wf_shadow()
        ctx.clear_queues()
    ctx.set_params(params)
    ctx.finish()
    cnt = np.array(cnt, copy=True)
    ctx.pixel_index_update(npix, int(cnt[0]))
    return cnt


def render_single_pass(ctx, max_bounces):
    """One sample per pixel with the microkernel integrator: the loop body of Tracer::renderSingle
    (reference: src/tracer.cpp:131-141)."""
    ctx.mk_raygen()
    for _ in range(int(max_bounces) + 1):
        ctx.mk_next_vertex()
        ctx.mk_sample_bsdf()
    ctx.mk_splat()


def render_single(ctx, params, spp):
    """Tracer::renderSingle (src/tracer.cpp:95-169): roulette off, reset, spp passes, post-process."""
    p = params.copy()
    p["useRoulette"] = 0
    ctx.set_params(p)
    ctx.mk_reset()
    for _ in range(spp):
        render_single_pass(ctx, p["maxBounces"])
    ctx.postprocess()
    ctx.finish()
    return p
