#!/usr/bin/env python3
"""Experiment (round 3): the memory-bound fused logic pass of one wavefront BESIDE the instruction-bound traversal of another.

Two wavefronts A, B on one GPU (pixel-interleaved partitions, as `bench.py --ctx-per-gpu 2`), three ways:
  free    -- both chains enqueued back to back, nothing orders them (what --ctx-per-gpu 2 does: they tend to run in phase)
  phased  -- A's [logic, genRays, materials] waits for B's and vice versa (HIP events between the two contexts' streams), so each
             wavefront's traversal runs while the other's logic pass does
  single  -- one wavefront with all the paths (the shipped configuration)

usage: python scripts/exp_phase.py [workload] [total paths]
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                    # noqa: E402
import bench                                    # noqa: E402
from fluctus_amd.device import HipContext       # noqa: E402
from fluctus_amd import driver                  # noqa: E402


def make(d, p, env, n, rank, nranks):
    g = HipContext(n)
    g.upload_scene(d); g.upload_envmap(env); g.set_partition(rank, nranks); g.set_params(p); driver.reset_renderer(g)
    return g


def l_phase(g):
    g.wf_logic(False); g.wf_raygen(); g.wf_materials()


def t_phase(g):
    g.wf_extend(); g.wf_shadow(); g.end_iteration_async()


def run(ctxs, streams, phased, steps, warmup):
    ev = [None] * len(ctxs)

    def step():
        for i, g in enumerate(ctxs):
            if phased:
                j = (i - 1) % len(ctxs)
                if ev[j] is not None:
                    streams[i].wait_event(ev[j])
            l_phase(g)
            if phased:
                e = torch.cuda.Event(); e.record(streams[i]); ev[i] = e
            t_phase(g)
    for _ in range(warmup):
        step()
    for g in ctxs:
        g.finish(); g.counter_totals(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for g in ctxs:
        g.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rays = 0.0
    for g in ctxs:
        tot = g.counter_totals(reset=True)
        rays += float(tot[1]) + float(tot[2])
    return rays / dt / 1e6, dt / steps * 1e3


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "kitchen"
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
    d, p, env = bench.build_workload(name=workload)
    torch.cuda.init()
    single = [make(d, p, env, total, 0, 1)]
    for rep in range(0 if os.environ.get("FLX_PHASE_SKIP_SINGLE") else 2):
        v, ms = run(single, None, False, 30, 24)
        print(f"{workload} {total} paths  single          {v:7.0f} Mrays/s  {ms:.3f} ms/step", flush=True)
    single[0].close()
    pair = [make(d, p, env, total // 2, i, 2) for i in range(2)]
    streams = [torch.cuda.ExternalStream(int(g.L.flx_stream(g.h))) for g in pair]
    for rep in range(2):
        for phased in (False, True):
            v, ms = run(pair, streams, phased, 30, 24)
            print(f"{workload} {total} paths  2 x {'phased' if phased else 'free  '}      {v:7.0f} Mrays/s  {ms:.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
