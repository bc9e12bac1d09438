#!/usr/bin/env python3
"""bench.py -- Mrays/s of the wavefront path-tracing hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): the kitchen-class scene at 1920x1080, 8 bounces, env-map MIS, separate
material queues.  Country-Kitchen.obj is a missing blob in the reference checkout, so the scene is the
deterministic procedural stand-in "kitchen-proc" (~0.5 M triangles, the real .mtl's material-type mix,
SURVEY 8(d)) under the reference's own environment map (assets/env_maps/night.hdr, tests/golden/night_env.npz); SBVH built by the host library.  NUM_TASKS = 16 777 216 paths in flight per
GPU (the reference's `wfBufferSize` setting, re-tuned for this chip -- see the comment at NUM_TASKS).

A step = one benchmark-style iteration of the reference's runBenchmark body (src/tracer.cpp:433-439):
logic -> raygen -> materials -> extension rays -> shadow rays -> end of iteration.
Metric: Mrays/s = (sum of extension-queue lengths + sum of shadow-queue lengths) / time.
Multi-GPU: every rank renders its own interleaved pixel subset with its own NUM_TASKS paths (weak scaling, no
collective in the timed region); the radiance tiles are gathered over RCCL afterwards (timed separately).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WIDTH, HEIGHT, BOUNCES = 1920, 1080, 8
# Paths in flight per GPU.  The reference's knob is the `wfBufferSize` setting (src/settings.cpp:20 "appropriate for dedicated GPU",
# 1e6 in settings_default.json, i.e. sized for a GTX-class part); an MI355X holds 458 k lanes at once, so 1 M paths is only 2.3 per
# lane and the traversal kernels spend much of their time in their tails.  Round 1 (kitchen-proc 1080p): 1 M 2762, 2 M 3027, 4 M 3188 Mrays/s.
# Round 4, same box, 4 M / 6 M / 8 M / 16 M (profiles/r04_num_tasks.txt): kitchen 5207 / 5417 / 5488 / 5297, conference 5146 / 5290 / 5388 /
# 5351, courtyard-1440p 2245 / 2368 / 2434 / 2480 -- with the persistent closest-hit kernel and the fused pass every launch has a fixed
# tail and seven dependent launches per iteration have their gaps, and 8 M paths amortise both (+5 % / +5 % / +8 %); 16 M loses again on the
# scenes whose tree lives in the Infinity Cache (3.3 GB of path state stream through it per iteration).  8 M paths = 1.6 GB of state +
# 0.3 GB of queues per GPU of 288 GB.
# Round 5, same box, 8 M / 12 M / 16 M / 24 M / 32 M (profiles/r05_num_tasks.txt): kitchen 6110-6260 / 6290 / 6340-6380 / 6400-6460 / 6470-6520, conference 5300 /
# 5400-5470 / 5510-5540, courtyard-1440p 2435 -> 2500 at 16 M, egyptcat 5000 -> 5330: the fused pass no longer loses per path when the state outgrows the
# Infinity Cache (its partial writes were what it paid for), so the fixed tails and launch gaps amortise further: 16 M paths (+2 ... +6 %) =
# 3.4 GB of state + 0.6 GB of queues per GPU of 288 GB; beyond that the gain per doubling is ~1 %.
NUM_TASKS = 1 << 24
TARGET_TRIS, SCENE_SEED = 500000, 42
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


# The other BASELINE.json configs (parity-test cases; selectable for extra measurements, never the default line)
WORKLOADS = {
    # name: (generator, target triangles, seed, bvh, width, height, bounces, useEnvMap, useAreaLight, camera pos, target)
    "kitchen": ("kitchen", TARGET_TRIS, SCENE_SEED, "sbvh", 1920, 1080, 8, 1, 0, (0.3, 1.5, 4.4), (0.0, 0.9, -0.5)),
    "conference": ("conference", 330000, 43, "sbvh", 1920, 1080, 8, 0, 1, (0.0, 1.2, 2.6), (0.0, 0.2, 0.0)),
    # (round 1 had to use the binned builder here: the serial SBVH build of 8.9 M triangles takes ~10 min; the parallel one ~1 min)
    "courtyard-1440p": ("courtyard", 10000000, 44, "sbvh", 2560, 1440, 12, 1, 0, (0.0, 3.0, 17.0), (0.0, 2.0, 0.0)),
    "courtyard-2160p": ("courtyard", 10000000, 44, "sbvh", 3840, 2160, 16, 1, 0, (0.0, 3.0, 17.0), (0.0, 2.0, 0.0)),
    # A REAL reference asset under the reference's own benchmark protocol: assets/egyptcat/egyptcat.obj is scene #1 of Tracer::runBenchmark
    # (src/tracer.cpp:384-389), rendered at 1024 x 1024 (:365-366) with the start-up parameters (:38-52, :760-797: default camera and area
    # light, no environment map, 10 bounces, ONE material queue) and the default wfBufferSize of 2^20 paths (src/settings.cpp:20; pass
    # --num-tasks 1048576 for the literal protocol).  The scene travels as data: tests/golden/egyptcat_scene.npz, written by
    # scripts/make_egyptcat_fixture.py from the OBJ / MTL / PNG through host/scene.cpp (/root/reference does not exist on the GPU box).
    "egyptcat": ("fixture:egyptcat_scene.npz", 16040, 0, "sbvh", 1024, 1024, 10, 0, 1, (0.0, 1.0, 3.5), (0.0, 1.0, 2.5)),
}
SINGLE_MATERIAL_QUEUE = {"egyptcat"}        # workloads that keep the reference's default wfSeparateQueues = 0


def build_workload(width=None, height=None, name="kitchen"):
    from fluctus_amd import host, wire
    gen, tris, seed, bvh, w, h, bounces, use_env, use_area, cam, target = WORKLOADS[name]
    width, height = width or w, height or h
    tris = int(os.environ.get("FLX_BENCH_TRIS", tris))          # experiments only (scripts/sweep.sh); the bench line uses the default
    if gen.startswith("fixture:"):
        z = np.load(os.path.join(ROOT, "tests", "golden", gen[len("fixture:"):]))
        d = host.SceneData()
        d.tris = z["tris"].view(wire.TRIANGLE).reshape(-1); d.materials = z["materials"].view(wire.MATERIAL).reshape(-1)
        d.texdesc = z["texdesc"].view(wire.TEXDESC).reshape(-1); d.texdata = z["texdata"]
    else:
        d = host.generate_scene(gen, tris, seed)
    # hierarchy cache (the reference's on-disk format, host/bvh.hpp) so that the back-to-back N = 1, 2, 4, 8 runs and the N ranks of
    # one run do not each spend 10-40 s in the SBVH builder; written atomically, keyed by the generator arguments
    cache_dir = os.environ.get("FLX_BVH_CACHE", "/tmp/flx_bvh_cache")
    cache = os.path.join(cache_dir, f"hierarchy_{gen.replace(':', '_')}_{tris}_{seed}_{bvh}_{d.tris.size}.bin")
    loaded = False
    if cache_dir and os.path.exists(cache):
        try:
            d.nodes, d.indices = host.bvh_import(cache)
            d.world_radius = host.bvh_import.world_radius
            loaded = True
        except Exception:
            loaded = False
    if not loaded:
        host.build_bvh(d, bvh)
        if cache_dir:
            try:
                os.makedirs(cache_dir, exist_ok=True)
                tmp = f"{cache}.{os.getpid()}.tmp"
                host.bvh_export_arrays(d, tmp)
                os.replace(tmp, cache)
            except Exception:
                pass
    p = wire.default_params(width, height, d.world_radius, d.tris.size)
    wire.look_at(p, cam, target, fov=60.0)
    p["maxBounces"], p["useEnvMap"], p["useAreaLight"], p["wfSeparateQueues"] = bounces, use_env, use_area, (0 if name in SINGLE_MATERIAL_QUEUE else 1)
    env = night_env()
    return d, p, env


_night = []


def night_env():
    """The reference's own environment map, assets/env_maps/night.hdr 512 x 256 (the map of the kitchen and courtyard configurations),
    from tests/golden/night_env.npz (scripts/make_envmap_fixture.py: pixels as read by the reference's rgbe.cpp); the alias / pdf tables are built by
    host/envmap.cpp.  Rounds 1-4 measured with host.synthetic_sky instead; a real night sky is far peakier (it changes the shadow-ray mix)."""
    if not _night:
        import hashlib
        from fluctus_amd import host
        z = np.load(os.path.join(ROOT, "tests", "golden", "night_env.npz"))
        e = host.envmap_from_rgb(int(z["w"]), int(z["h"]), z["rgb"])
        got = hashlib.sha256(e.prob.tobytes() + e.alias.tobytes() + e.pdf.tobytes()).hexdigest()
        assert got == str(z["tables_sha256"]), "night_env.npz: the sampling tables built here differ from the ones the fixture was made with"
        _night.append(e)
    return _night[0]


def step_async(ctx):
    ctx.wf_logic(False)
    ctx.wf_raygen()
    ctx.wf_materials()
    ctx.wf_extend()
    ctx.wf_shadow()
    ctx.end_iteration_async()


def algorithmic_bytes(stats):
    """extension ray 84 + 64*n_inner + 40*n_tri + 64*[hit] bytes; shadow ray 36 + 64*n_inner + 40*n_tri."""
    ext = 84 * stats["ext_rays"] + 64 * stats["ext_inner"] + 40 * stats["ext_tri"] + 64 * stats["ext_hits"]
    sh = 36 * stats["shadow_rays"] + 64 * stats["shadow_inner"] + 40 * stats["shadow_tri"]
    return ext, sh


def capture_key(args, ctx, p, C=1):
    """What a PMC capture of the extension kernel depends on (see main(): roofline)."""
    from fluctus_amd import build
    return {"workload": args.workload, "width": int(p["width"]), "height": int(p["height"]), "max_bounces": int(p["maxBounces"]),
            "num_tasks": args.num_tasks // C, "extend_tree": args.extend_tree, "shadow_tree": args.shadow_tree,
            "refill_extend": ctx.get_option("refill_extend"), "refill_shadow": ctx.get_option("refill_shadow"), "shadow_split": ctx.get_option("shadow_split"), "overlap": ctx.get_option("overlap"),
            "fuse": int(args.fuse), "fuse_set": ctx.get_option("fuse_set_now"), "ext_order": ctx.get_option("ext_order"),
            "shadow_split": ctx.get_option("shadow_split"), "regen": ctx.get_option("regen"), "regroup": ctx.get_option("regroup"), "regen_prep": ctx.get_option("regen_prep"),
            # which BINARY ran: the shipped library or an A/B variant (FLX_HIP_LIB, e.g. a -DFLX_LAB build of the same sources), and its compile flags
            "library": os.path.basename(os.environ.get("FLX_HIP_LIB") or "libfluctus_hip.so"), "build_flags": " ".join(build.HIP_FLAGS),
            "source_hash": build.source_hash()}


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box reports 256 logical CPUs but grants 16 through cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(d, p, env, budget_s=12.0, force_kind=None):
    """CPU baseline on the host cores, bounded sample: the same scene / camera / parameters with 65 536 paths in flight,
    16 warm-up iterations, then whole iterations until `budget_s` seconds have elapsed.

    kind "reference": oracle/_ref/libfluctus_ref.so -- the reference's OWN OpenCL kernels compiled for x86-64 in the build
    container (oracle/ref/Makefile), their NDRanges spread over the usable cores in chunks of 64 work-items the way a CPU
    OpenCL device schedules work-groups.  kind "port" (fallback when that library is absent): oracle/wf_oracle.cpp."""
    from fluctus_amd import driver
    from oracle import build as oracle_build
    oracle_build.build_all()
    from oracle.binding import OracleContext, RefContext, ref_available
    cores = usable_cores()
    n = 1 << 16
    kind = force_kind or ("reference" if ref_available() and os.environ.get("FLX_CPU_BASELINE", "") != "port" else "port")
    c = RefContext(n, threads=cores) if kind == "reference" else OracleContext(n, threads=cores)
    c.upload_scene(d)
    c.upload_envmap(env)
    c.set_params(p)
    driver.reset_renderer(c)
    npix = int(p["width"]) * int(p["height"])
    for _ in range(16):
        driver.benchmark_iteration(c, npix)
    rays, iters = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        cnt = driver.benchmark_iteration(c, npix)
        rays += int(cnt[1]) + int(cnt[2])
        iters += 1
    dt = time.perf_counter() - t0
    c.close()
    what = "the reference's wf_*.cl kernels built for x86-64 (oracle/_ref), NDRange over host threads" if kind == "reference" else "oracle/wf_oracle.cpp (OpenMP)"
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": kind,
            "sample": f"{what}; same scene/params, 65536 paths in flight, {iters} iterations after 16 warm-up, {dt:.1f} s"}


def ref_gpu_baseline(ctx, d, p, env, args, settle):
    """The reference's OWN wf_*.cl kernels on THIS GPU (untimed leg, rank 0 at N = 1; the product path is untouched): oracle/_ref/gfx950/fast/*.co =
    /root/reference/src/wf_*.cl compiled unmodified for gfx950 with the reference's own build flags (-cl-fast-relaxed-math, src/clcontext.cpp:145) and
    AMD's OpenCL built-in library (oracle/ref/Makefile, target gfx950), loaded by oracle/ref_gpu.py -- test infrastructure, imported here exactly like
    cpu_baseline imports oracle.binding.
      (a) `traversal`: the reference's traceExtension / traceShadow (src/wf_extrays.cl:5-36, src/wf_shadowrays.cl:6-38; NDRange = NUM_TASKS work-items,
          src/clcontext.cpp:815-850) on the SAME extension / shadow queues and path state as the product's steady state at --num-tasks, each alone on the
          machine, best of 3 -- beside the product's k_trace4r / k_shadow4 alone (serial schedule) on the same steady state;
      (b) `whole_loop`: the reference's complete runBenchmark iteration -- for a workload with an environment map with ONE builder-written piece in it: `logic`
          with USE_ENV_MAP samples an image, gfx950 has no image instructions, so its read_imagef / get_image_dim come from oracle/ref/gfx950_image_standin.cl
          (everything else of that kernel, and every other kernel, is AMD's compiler + built-in library on the reference's unmodified source;
          `whole_loop_image_standin` says which case it is) -- the reference's complete runBenchmark iteration (src/tracer.cpp:433-439, one finishQueue per iteration) at its own wfBufferSize 2^20 and at
          --num-tasks -> Mrays/s, beside the product at the same sizes."""
    from fluctus_amd import device, driver
    from oracle import ref_gpu
    out = {"kind": "the reference's wf_*.cl kernels compiled for gfx950 (oracle/_ref/gfx950), same GPU", "flavour": "fast"}
    if not ref_gpu.available("fast"):
        out["error"] = "oracle/_ref/gfx950/fast not built (make -C oracle/ref gfx950 needs /root/reference: build container only)"
        return out
    n = int(args.num_tasks)
    npix = int(p["width"]) * int(p["height"])
    Q_EXT, Q_SH = 1, 2
    # ---- (a) the two traversal kernels on the product's own steady-state queues
    try:
        ctx.set_option("overlap", 0)
        ctx.wf_logic(False); ctx.wf_raygen(); ctx.wf_materials()
        cnt = ctx.get_counters(); ctx.finish()
        cnt = np.array(cnt, copy=True)
        state = ctx.state_export()
        qe, qs = ctx.queue_read(Q_EXT), ctx.queue_read(Q_SH)
        r, backend = None, None
        for backend in ("opencl", "hip"):
            try:
                r = ref_gpu.RefGpuContext(n, backend_name=backend, flavour="fast")
                break
            except Exception as e:                      # (e.g. CL_DEVICE_MAX_MEM_ALLOC_SIZE below the 4.3 GB state of 16 M paths)
                out[f"{backend}_backend_error"] = str(e)[:200]
        if r is None:
            raise RuntimeError("no backend could hold the reference's path state")
        r.upload_scene(d); r.set_params(p)
        r.state_import(state); del state
        r.queue_write(Q_EXT, qe); r.queue_write(Q_SH, qs); r.set_counters(cnt)
        r.timed = True
        te, ts = [], []
        for _ in range(3):
            r.wf_extend(); r.wf_shadow(); r.finish()
            te.append(r.last_ms["traceExtension"]); ts.append(r.last_ms["traceShadow"])
        r.close()
        # the product's kernels alone on the same steady state (this iteration + 3 more, serial schedule, every launch event-timed)
        ctx.profile_reset(); ctx.profile_enable(1)
        ctx.wf_extend(); ctx.wf_shadow(); ctx.end_iteration_async()
        for _ in range(3):
            step_async(ctx)
        ctx.finish(); ctx.profile_enable(0)
        prof = ctx.profile_get()
        he, hs = prof["extend"][0] / max(1, prof["extend"][1]), prof["shadow"][0] / max(1, prof["shadow"][1])
        out["traversal"] = {"paths": n, "ext_rays": int(cnt[Q_EXT]), "shadow_rays": int(cnt[Q_SH]), "loader": backend,
                            "ref_traceExtension_ms": min(te), "ref_traceShadow_ms": min(ts),
                            "hip_extend_alone_ms": he, "hip_shadow_alone_ms": hs,
                            "speedup_extend": min(te) / he if he else None, "speedup_shadow": min(ts) / hs if hs else None,
                            "speedup_pair": (min(te) + min(ts)) / (he + hs) if (he + hs) else None}
    except Exception as e:
        out["traversal_error"] = f"{type(e).__name__}: {e}"[:300]
    finally:
        ctx.set_option("overlap", args.overlap)
        ctx.counter_totals(reset=True)
    # ---- (b) the reference's whole loop.  Without an environment map every kernel of it runs on gfx950 as AMD's compiler and built-in library made it; with one,
    # `logic` is the image stand-in build (oracle/ref/gfx950_image_standin.cl: read_imagef / get_image_dim restated, everything else AMD's) -- labelled.
    use_env = bool(int(p["useEnvMap"]))
    if (not use_env) or ref_gpu.available_env("fast"):
        loops = {}
        out["whole_loop_image_standin"] = use_env
        for size in sorted({1 << 20, n}):
            try:
                r = ref_gpu.RefGpuContext(size, backend_name="hip", flavour="fast")      # (`logic` carries an image2d_t argument: module loader, null descriptor)
                r.upload_scene(d); r.set_params(p)
                if use_env:
                    r.upload_envmap(env)
                driver.reset_renderer(r)
                for _ in range(settle):
                    driver.benchmark_iteration(r, npix)
                rays, iters, t0 = 0, 0, time.perf_counter()
                while iters < 8 or (time.perf_counter() - t0 < 3.0 and iters < 200):
                    c_ = driver.benchmark_iteration(r, npix)
                    rays += int(c_[Q_EXT]) + int(c_[Q_SH]); iters += 1
                dt = time.perf_counter() - t0
                r.close()
                g = device.HipContext(size)
                g.upload_scene(d)
                if use_env:
                    g.upload_envmap(env)
                g.set_params(p)
                driver.reset_renderer(g)
                for _ in range(settle):
                    step_async(g)
                g.finish(); g.counter_totals(reset=True)
                k = 20 if size > (1 << 22) else 60
                g0 = time.perf_counter()
                for _ in range(k):
                    step_async(g)
                g.finish()
                gdt = time.perf_counter() - g0
                tot = g.counter_totals(reset=True)
                g.close()
                loops[str(size)] = {"ref_Mrays_s": rays / dt / 1e6, "ref_ms_per_iteration": dt / iters * 1e3, "ref_iterations": iters,
                                    "hip_Mrays_s": (float(tot[1]) + float(tot[2])) / gdt / 1e6, "hip_ms_per_iteration": gdt / k * 1e3,
                                    "speedup": ((float(tot[1]) + float(tot[2])) / gdt) / (rays / dt) if rays else None}
            except Exception as e:
                loops[str(size)] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["whole_loop"] = loops
    else:
        out["whole_loop"] = None
        out["whole_loop_note"] = "logic with USE_ENV_MAP samples an image2d_t (read_imagef); gfx950 has no image support and the image stand-in builds (oracle/_ref/gfx950/fast/logic_v*_imgstandin.co) are absent"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--windows", type=int, default=5, help="back-to-back timed windows of --steps steps each; the headline is the MEDIAN window (all of them are in the line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-gpu-baseline", action="store_true", help="skip the untimed leg that runs the reference's own gfx950 kernels on this GPU (ref_gpu_baseline)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--workload", default="kitchen", choices=sorted(WORKLOADS))
    ap.add_argument("--num-tasks", type=int, default=NUM_TASKS)
    ap.add_argument("--xcd-remap", type=int, default=0)
    ap.add_argument("--extend-tree", type=int, default=4, choices=(2, 4), help="tree flx_wf_extend walks: 4-wide quantised (default) or the reference's binary tree (bit-exact)")
    ap.add_argument("--shadow-tree", type=int, default=4, choices=(2, 4))
    ap.add_argument("--overlap", type=int, default=-1, help="stream schedule of the two traversals: 0 serial, 1 shadow beside extension, 2 shadow right after logic, -1 = what flx_upload_scene picks for the scene")
    ap.add_argument("--fuse", type=int, default=1, choices=(0, 1), help="logic + material kernels as one fused pass (default) or the separate kernels")
    ap.add_argument("--ext-order", type=int, default=-1, choices=(-1, 0, 1, 2), help="fused pass: extension queue lists the continuing paths by path id (1) or in one segment per material queue (0); -1 = what flx_upload_scene chose")
    ap.add_argument("--fuse-set", type=int, default=0, choices=(0, 1, 31), help="BSDF types the fused pass inlines: 0 = what flx_upload_scene chose, 1 diffuse, 31 all")
    ap.add_argument("--refill-extend", type=int, default=-1, help="closest-hit traversal with persistent waves: refill when this many lanes are idle (0 = thread-per-ray kernel, -1 = library default)")
    ap.add_argument("--refill-shadow", type=int, default=-1, help="the same for the any-hit traversal")
    ap.add_argument("--regen", type=int, default=-1, help="in-kernel regeneration of terminating paths by the fused logic pass (option regen): 0 off (genRays kernel), 1 on, -1 = library default")
    ap.add_argument("--regen-prep", type=int, default=-1, help="prepared regeneration (option regen_prep): the seed-only half of genRays inside the fused RAW pass; 0 / 1, -1 = library default (1)")
    ap.add_argument("--regroup", type=int, default=-1, help="all-types fused pass with its material step sorted by BSDF type per block (option regroup): 0 / 1, -1 = what flx_upload_scene picks for the scene")
    ap.add_argument("--shadow-split", type=int, default=-1, help="tail splitting of the any-hit kernel: node-visit budget of the pass over the queue | budget of a second pass << 8; 0 = off, -1 = library default")
    ap.add_argument("--ctx-per-gpu", type=int, default=1, help="independent wavefronts per GPU (pixel-interleaved sub-partitions, paths split evenly)")
    ap.add_argument("--node-layout", type=int, default=1)
    ap.add_argument("--eager-bump", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"), help="multi-GPU: weak = every rank runs --num-tasks paths of its own (default, the driver's SCALE runs); strong = the paths in flight are fixed: --num-tasks // world per rank (the framebuffer is fixed either way: north_star's tile split)")
    ap.add_argument("--kernel-timing", type=int, default=4, help="HIP-event timing inside the timed region: 0 none, 1 every kernel, 2 the trace kernels + span, 3 the extension kernel only, 4 the three kernels of the roofline block (extension, logic, shadow)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    entry.build_product()             # the checker (oracle/) is built by cpu_baseline(), the only leg that uses it

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    num_tasks_total = args.num_tasks
    if args.scaling == "strong":
        # total work fixed as N grows: the same paths in flight over the whole job, split evenly (whole waves per rank); the pixel partition is unchanged
        args.num_tasks = max(64, (args.num_tasks // world) & ~63)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    # FLX_FORCE_DIST=1: take the RCCL code path (init, barrier, all-reduce, tile gather) even with one rank, so it can be
    # exercised on a 1-GPU box under `python -m torch.distributed.run --nproc-per-node 1`
    use_dist = world > 1 or os.environ.get("FLX_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from fluctus_amd import device, driver
    d, p, env = build_workload(args.width, args.height, args.workload)
    args.width, args.height = int(p["width"]), int(p["height"])
    # C independent wavefronts per GPU: the framebuffer partition is simply refined (rank*C + i of world*C), each wavefront
    # has its own path state (num_tasks / C paths), queues and streams; their kernels interleave on the device so one
    # wavefront's TA-bound traversal overlaps the other's HBM-bound logic / material / raygen phases.
    C = max(1, args.ctx_per_gpu)
    ctxs = []
    for i in range(C):
        c_ = device.HipContext(args.num_tasks // C, device_index=local_rank)
        c_.set_option("xcd_remap", args.xcd_remap)
        c_.set_option("extend_tree", args.extend_tree)
        c_.set_option("shadow_tree", args.shadow_tree)
        c_.set_option("overlap", args.overlap)
        c_.set_option("node_layout", args.node_layout)
        c_.set_option("eager_bump", args.eager_bump)
        c_.set_option("fuse", args.fuse)
        if args.refill_extend >= 0:
            c_.set_option("refill_extend", args.refill_extend)
        if args.refill_shadow >= 0:
            c_.set_option("refill_shadow", args.refill_shadow)
        if args.shadow_split >= 0:
            c_.set_option("shadow_split", args.shadow_split)
        if args.regen >= 0:
            c_.set_option("regen", args.regen)
        c_.set_option("regroup", args.regroup)
        if args.regen_prep >= 0:
            c_.set_option("regen_prep", args.regen_prep)
        c_.upload_scene(d)
        if args.fuse_set:
            c_.set_option("fuse_set", args.fuse_set)          # after the upload, which picks one for the scene
        if args.ext_order >= 0:
            c_.set_option("ext_order", args.ext_order)
        c_.upload_envmap(env)
        c_.set_partition(rank * C + i, world * C)
        c_.set_params(p)
        driver.reset_renderer(c_)
        ctxs.append(c_)
    ctx = ctxs[0]

    class _Group:
        """Fan the calls the measurement code makes out over the GPU's wavefronts."""
        def __getattr__(self, name):
            def call(*a, **k):
                out = [getattr(c_, name)(*a, **k) for c_ in ctxs]
                return out[0]
            return call

        def counter_totals(self, reset=False):
            return sum(c_.counter_totals(reset) for c_ in ctxs)

        def profile_get(self):
            acc = {}
            for c_ in ctxs:
                for k, (ms, n) in c_.profile_get().items():
                    a = acc.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += n
            return {k: (v[0], v[1]) for k, v in acc.items()}

        def stats(self):
            acc = {}
            for c_ in ctxs:
                for k, v in c_.stats().items():
                    acc[k] = acc.get(k, 0) + v
            return acc
    if C > 1:
        ctx = _Group()

    def barrier():
        if use_dist:
            dist.barrier()

    # BASELINE.md 2 / SURVEY 8(d): the timed window starts on a STATIONARY queue mix, i.e. after >= 2 x maxBounces iterations from reset
    # (before that no path has reached maxBounces and the wave of first terminations would fall inside the window).  --warmup is honoured
    # when it asks for more; the line reports what was run as `settle_iterations` and echoes the flag as `warmup`.
    settle = max(args.warmup, 2 * int(p["maxBounces"]) + 2)
    for _ in range(settle):
        step_async(ctx)
    ctx.finish()
    ctx.counter_totals(reset=True)
    ctx.profile_reset()
    ctx.profile_enable(args.kernel_timing)

    # `--windows` back-to-back windows of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides; a window is
    # ~0.1 s of GPU time, one sample of it moves by a few per cent from run to run, so the headline is the median window
    nwin = max(1, args.windows)
    win_elapsed, win_tot = [], []
    for _w in range(nwin):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_async(ctx)
        ctx.finish()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        win_elapsed.append(t1 - t0)
        win_tot.append(np.array(ctx.counter_totals(reset=True), dtype=np.float64))
    ctx.profile_enable(False)

    tot = np.sum(win_tot, axis=0)                     # all windows: what the event-timed kernel averages cover
    prof = ctx.profile_get()
    # kernels not timed inside the timed region (--kernel-timing 2 times only the trace kernels there, 0 none): averages from
    # an extra UNTIMED pass over the same steady state, so the JSON line still carries every kernel
    untimed = []
    if args.kernel_timing != 1:
        ctx.profile_reset(); ctx.profile_enable(1)
        for _ in range(16):
            step_async(ctx)
        ctx.finish(); ctx.profile_enable(0)
        extra = ctx.profile_get()
        for k, v in extra.items():
            if not prof.get(k, (0.0, 0))[1]:
                prof[k] = v; untimed.append(k)
        ctx.counter_totals(reset=True)
    # the extension kernel WITHOUT the concurrent shadow traversal (serial schedule), untimed: in the timed region the two share the
    # machine and the event-timed duration of either depends on how the hardware splits it between them
    ctx.set_option("overlap", 0)
    ctx.profile_reset(); ctx.profile_enable(3)
    for _ in range(12):
        step_async(ctx)
    ctx.finish(); ctx.profile_enable(0)
    alone_ms, alone_n = ctx.profile_get()["extend"]
    ctx.set_option("overlap", args.overlap)
    ctx.counter_totals(reset=True)
    # per window: elapsed = max over ranks, rays = sum over ranks
    wins = np.array([[e, t[0], t[1], t[2]] for e, t in zip(win_elapsed, win_tot)], dtype=np.float64)      # [window][elapsed, primary, extension, shadow]
    if use_dist:
        te = torch.tensor(wins[:, 0].copy(), dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        tr = torch.tensor(wins[:, 1:].copy(), dtype=torch.float64, device="cuda")
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        wins = np.concatenate([te.cpu().numpy()[:, None], tr.cpu().numpy()], axis=1)
    win_mrays = (wins[:, 2] + wins[:, 3]) / wins[:, 0] / 1e6
    order = np.argsort(win_mrays)
    med = int(order[len(order) // 2]) if len(order) % 2 else int(order[len(order) // 2 - 1])      # (even count: the lower middle window -- a measured one)
    elapsed = float(wins[med, 0])
    prim, ext, sh = float(wins[med, 1]), float(wins[med, 2]), float(wins[med, 3])
    rays_total = ext + sh
    ext_all = float(wins[:, 2].sum())                  # extension rays of all windows and ranks

    # ---- roofline of the dominant kernel (traceExtension): algorithmic bytes / HIP-event time
    # visit counts come from an UNTIMED pass of the counting kernel variants over the same steady state
    ctx.trace_stats_enable(True)
    ctx.reset_stats()
    for _ in range(4):
        step_async(ctx)
    ctx.finish()
    st = ctx.stats()
    own_leaf = ctx.leaf_stats() if (C == 1 and args.extend_tree == 4) else None
    st_own = dict(st)
    if args.extend_tree == 4 or args.shadow_tree == 4:
        # SURVEY 8(d) defines the algorithmic bytes through the visit counts of the REFERENCE traversal (binary tree, near child first) on
        # the same BVH: count them with the binary kernels on the same steady state (untimed); st_own keeps what the 4-wide kernels did
        ctx.set_option("extend_tree", 2); ctx.set_option("shadow_tree", 2)
        ctx.reset_stats()
        for _ in range(4):
            step_async(ctx)
        ctx.finish()
        st = ctx.stats()
        ctx.set_option("extend_tree", args.extend_tree); ctx.set_option("shadow_tree", args.shadow_tree)
    simd = None
    if C == 1:
        ws = ctx.wave_stats()
        def eff(lane, wave): return lane / (64.0 * wave) if wave else None
        simd = {"ext_inner": eff(st["ext_inner"], ws["ext"]["inner"]), "ext_tri": eff(st["ext_tri"], ws["ext"]["tri"]),
                "shadow_inner": eff(st["shadow_inner"], ws["shadow"]["inner"]), "shadow_tri": eff(st["shadow_tri"], ws["shadow"]["tri"]),
                "ext_outer_trips_per_wave": ws["ext"]["outer"] / max(1.0, st["ext_rays"] / 64.0),
                "shadow_outer_trips_per_wave": ws["shadow"]["outer"] / max(1.0, st["shadow_rays"] / 64.0),
                "ext_longest_ray_inner_per_wave": ws["ext_max_inner_sum"] / max(1.0, st["ext_rays"] / 64.0), "ext_wave_trips": ws["ext"], "shadow_wave_trips": ws["shadow"]}
    ctx.trace_stats_enable(False)
    ext_bytes, sh_bytes = algorithmic_bytes(st)
    bytes_per_ext_ray = ext_bytes / max(1, st["ext_rays"])
    # what the kernel that ran really touches per ray: 84 B of path state + one 64-B record per wide-node visit + 32 B per leaf header
    # + 48 B per triangle record + the 64-B shading record of a hit
    own_bytes_per_ray = None
    if own_leaf is not None:
        own_bytes_per_ray = (84 * st_own["ext_rays"] + 64 * st_own["ext_inner"] + 32 * own_leaf["ext_leaf"] + 48 * st_own["ext_tri"]
                             + 64 * st_own["ext_hits"]) / max(1, st_own["ext_rays"])
    ext_ms, ext_n = prof["extend"]
    rays_per_launch = ext_all / world / C / max(1, args.steps * nwin)
    achieved = (bytes_per_ext_ray * rays_per_launch) / (ext_ms / max(1, ext_n) * 1e-3) / 1e9 if ext_ms > 0 else 0.0
    # k_extend shares the machine with the concurrent k_shadow; the pair's span gives the combined traversal rate
    span_ms, span_n = prof.get("trace_span", (0.0, 0))
    bytes_per_sh_ray = sh_bytes / max(1, st["shadow_rays"])
    combined = None
    if span_n:
        combined_bytes = bytes_per_ext_ray * rays_per_launch + bytes_per_sh_ray * (float(wins[:, 3].sum()) / world / C / max(1, args.steps * nwin))
        combined = combined_bytes / (span_ms / span_n * 1e-3) / 1e9
    # fabric-side bytes of the extension kernel per launch, from the committed PMC capture of THIS workload (profiles/traffic_<workload>.json,
    # written by scripts/profile_round.sh: separate --pmc passes, request counters by size = 2 x FETCH_SIZE + WRITE_SIZE with the guide's gfx950
    # correction).  A capture is quoted only for the exact configuration AND kernel sources it was taken on: it carries `capture_key`
    # (workload, resolution, bounces, paths in flight, trees, refill / overlap / fused-pass settings and a hash of csrc/* + include/*.h);
    # any difference makes it stale and the line says which keys differ instead of quoting it.
    traffic = traffic_lines = traffic_lanes = traffic_valu = None
    traffic_stale = None
    passes = {}
    key = capture_key(args, ctx, p, C)
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            have = tj.get("capture_key") or {}
            diff = sorted(k for k in key if have.get(k) != key[k])
            if not diff:
                traffic = tj.get("extend_hbm_bytes_per_launch")
                traffic_lines = tj.get("extend_read_requests_128B")
                traffic_lanes = tj.get("extend_lanes_per_valu_instruction")
                traffic_valu = tj.get("extend_valu_instructions")
                passes = tj.get("passes") or {}
            else:
                traffic_stale = diff
        except Exception as e:
            traffic, traffic_stale = None, [f"unreadable: {e}"]

    # ---- multi-GPU: gather the radiance tiles over RCCL (outside the timed region)
    gather_ms = None
    native_hung = False
    native_err = native_ok = None
    if use_dist:
        lp = ctxs[0].local_pixels()
        maxlp = (args.width * args.height + world - 1) // world
        tile = torch.zeros((maxlp, 4), dtype=torch.float32, device="cuda")
        ctxs[0].copy_pixels_to_device(tile.data_ptr())     # (with --ctx-per-gpu > 1 only the first wavefront's tile is exercised here)
        ctx.finish()
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        from fluctus_amd import multi
        full = multi.gather_tiles(tile, args.width * args.height, rank, world)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        # the same gather through the library's own RCCL group (flx_group_init / flx_gather: what the C++ host uses), id shipped by torch
        # It runs under a watchdog thread: the bench line must come out even if this second RCCL communicator cannot be formed on some
        # node (nothing above N = 1 could be tested in the build environment); a failure is REPORTED in the line, never silently skipped.
        native_ms = native_ok = native_err = native_seen = None
        native_hung = False
        if C == 1 and os.environ.get("FLX_NATIVE_GATHER", "1") != "0":
            import threading
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            id_err = None
            if rank == 0:
                try:
                    idt.copy_(torch.frombuffer(bytearray(device.group_unique_id()), dtype=torch.uint8))
                except Exception as e:              # librccl not loadable, ...
                    id_err = str(e)
            dist.broadcast(idt, src=0)
            id_bytes = bytes(idt.cpu().numpy().tobytes())
            res = {}
            full_host = full.cpu().numpy() if rank == 0 else None

            def _native():
                try:
                    if id_err or not any(id_bytes):
                        raise RuntimeError(id_err or "no unique id")
                    ctxs[0].group_init(rank, world, id_bytes)
                    res["nranks_seen"] = ctxs[0].group_info()[0]          # ncclCommCount of the library's own communicator
                    n0 = time.perf_counter()
                    img = ctxs[0].gather(0)
                    res["ms"] = (time.perf_counter() - n0) * 1e3
                    if rank == 0:
                        res["ok"] = bool(np.array_equal(img, full_host))
                except Exception as e:
                    res["err"] = str(e)
            th = threading.Thread(target=_native, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("FLX_NATIVE_GATHER_TIMEOUT", "120")))
            if th.is_alive():
                native_hung = True
                native_err = "flx_group_init / flx_gather did not return within the watchdog timeout"
            else:
                native_ms, native_ok, native_err = res.get("ms"), res.get("ok"), res.get("err")
                native_seen = res.get("nranks_seen")
            if rank == 0 and native_ok is False:
                native_err = "flx_gather differs from torch.distributed.gather"
        gather_ok = None
        if rank == 0:
            assert torch.isfinite(full).all() and lp > 0
            # the gathered image holds rank r's tile at global pixels r, r + world, ...: rank 0's slice must be what
            # flx_read_pixels returns for its own framebuffer (bit for bit: nothing rendered in between)
            own = ctxs[0].read_pixels(0)
            gather_ok = bool(np.array_equal(full[0::world][:lp].cpu().numpy(), own))
            assert gather_ok, "gathered image differs from flx_read_pixels"

    launch_s = ext_ms / max(1, ext_n) * 1e-3
    alone_s = alone_ms / max(1, alone_n) * 1e-3
    own_bytes = own_bytes_per_ray * rays_per_launch if own_bytes_per_ray else None
    contract_bytes = bytes_per_ext_ray * rays_per_launch
    # `achieved` / `frac`: bytes that really cross the L2 <-> fabric boundary for this kernel (PMC capture of this workload AND these kernel
    # sources) over the launch time measured live -- the north-star's own definition ("rocprof-reported HBM bandwidth in the traversal
    # kernel").  Fabric-side: Infinity-Cache hits are included (the guide: FETCH_SIZE / the TCC_EA0 request counters cannot exclude them), so
    # it is an UPPER bound of HBM proper.  Without a valid capture `achieved` and `frac` are null -- never a different quantity under the same
    # name: the bytes the RUNNING kernel touches per ray (its own node / leaf / triangle counters, an upper bound of what can reach HBM)
    # are `frac_own` only.  SURVEY 8(d)'s contract figure -- bytes the REFERENCE's binary traversal would touch for the same rays -- is kept
    # as `contract_equivalent_GBps`: the 4-wide kernel reaches the same hits with half the visits, so that figure exceeds the peak and is a
    # speed-up measure, not a bandwidth.
    if traffic and launch_s > 0:
        ach, src = traffic / launch_s / 1e9, "counters"
    else:
        ach, src = None, None
    # ---- the instruction-issue roof.  gfx950's SIMD has two 16-lane VALU pipes (scripts/ubench/valu_*.hip, profiles/r03_ubench_*): plain mul / add / fmac / logic
    # instructions issue on either (2.1-2.4 SIMD-cycles per wave64 instruction), compares, selects, min / max, shifts and conversions on ONE of them (4.0-4.4).
    # A kernel cannot retire more than SIMDs x clock / c wave-instructions per second with c between those two:
    #     issue_frac     = instructions x 4 / (SIMDs x clock x launch time)   (all single-pipe: the TRAVERSAL kernels' mix -- 3.97 measured on k_extend4)
    #     issue_frac_min = instructions x 2 / ...                              (all dual-pipe: what even a pure-arithmetic mix cannot beat)
    # `bound` of a traversal kernel compares its HBM fraction with issue_frac; of the logic pass (arithmetic-heavy, mix not measured) with issue_frac_min, i.e.
    # it says "valu-issue" only when the instruction stream would saturate the SIMDs even at the dual-pipe rate.  (rocprof's derived VALUBusy = SQ_ACTIVE_INST_VALU
    # x 4 / SIMDs / cycles assumes 4 cycles for every instruction and exceeds 100 % on the all-types logic pass: not used.)
    prop = torch.cuda.get_device_properties(local_rank)
    simds = int(prop.multi_processor_count) * 4
    clock_hz = float(getattr(prop, "clock_rate", 2400000)) * 1e3
    CYC, CYC_MIN = 4.0, 2.0
    # Round 6: the cost of THIS kernel's instruction mix from its ISA (scripts/valu_roof.py -> profiles/valu_roof.json: opcode histogram of the hot loops of the
    # shipped code object x the per-class issue cost the micro-benchmarks measured, two-pipe model), quoted only while the file's source_hash is that of the
    # sources this library was built from.  Without it the line falls back to the round-5 bracket (4 = every instruction single-pipe, 2 = every one dual-pipe).
    isa_roof = {}
    try:
        from fluctus_amd import build as _b
        _vr = json.load(open(os.path.join(ROOT, "profiles", "valu_roof.json")))
        if _vr.get("source_hash") == _b.source_hash() and not os.environ.get("FLX_HIP_LIB"):
            isa_roof = _vr.get("kernels") or {}
    except Exception:
        isa_roof = {}
    any_order = 1 if (int(p["useEnvMap"]) and not int(p["useAreaLight"])) else 0
    k_ext = "k_trace4r<false, 0>" if ctx.get_option("refill_extend") else "k_extend4<false>"
    k_sh = None if ctx.get_option("shadow_split") else (f"k_trace4r<true, {any_order}>" if ctx.get_option("refill_shadow") else f"k_shadow4<false, {any_order}>")
    _raw = bool(ctx.get_option("refill_extend")) and args.extend_tree == 4
    _rg = _raw and ctx.get_option("fuse_set_now") == 31 and bool(ctx.get_option("regroup")) and (args.num_tasks // C) % 256 == 0
    k_lg = (f"k_logic<{ctx.get_option('fuse_set_now')}, {'true' if _raw else 'false'}, {'true' if _rg else 'false'}>") if args.fuse else "k_logic<0, false, false>"

    def valu_block(insts, lanes, secs, kname=None):
        if not insts or not secs:
            return None
        avail = simds * clock_hz * secs
        roof = isa_roof.get(kname) if kname else None
        if roof and roof.get("cycles_per_instruction"):
            cpi = float(roof["cycles_per_instruction"])
            return {"instructions_per_launch": insts, "cycles_per_instruction": cpi, "cycles_per_instruction_range": roof.get("range"),
                    "cycles_per_instruction_source": f"scripts/valu_roof.py: ISA of {kname} in the shipped code object (opcode histogram of its hot loops, trip-weighted) x measured per-class issue cost, two-pipe model; profiles/valu_roof.json",
                    "simds": simds, "clock_GHz": clock_hz / 1e9, "issue_frac": insts * cpi / avail,
                    "issue_frac_range": [insts * float(c_) / avail for c_ in (roof.get("range") or [cpi, cpi])],
                    "lanes_per_instruction": lanes, "useful_lane_frac": (lanes / 64.0) if lanes else None}
        return {"instructions_per_launch": insts, "cycles_per_instruction": {"single_pipe": CYC, "dual_pipe": CYC_MIN}, "cycles_per_instruction_source": "bracket (no profiles/valu_roof.json for these sources)",
                "simds": simds, "clock_GHz": clock_hz / 1e9,
                "issue_frac": insts * CYC / avail, "issue_frac_min": insts * CYC_MIN / avail,
                "lanes_per_instruction": lanes, "useful_lane_frac": (lanes / 64.0) if lanes else None}

    def pass_block(name, prof_keys, label, traversal, kname=None):
        pb = passes.get(name) or {}
        ms, n = 0.0, 0
        for k_ in prof_keys:
            if prof.get(k_, (0.0, 0))[1]:
                ms, n = prof[k_]; break
        secs = (ms / n * 1e-3) if n else None
        by = pb.get("hbm_bytes_per_launch")
        hb = (by / secs / 1e9) if (by and secs) else None
        vb = valu_block(pb.get("valu_instructions_per_launch"), pb.get("lanes_per_valu_instruction"), secs, kname)
        fr = {"hbm": (hb / HBM_PEAK_GBS) if hb else None, "valu-issue": (vb["issue_frac"] if (traversal or "issue_frac_min" not in vb) else vb["issue_frac_min"]) if vb else None}
        best = max((v, k_) for k_, v in fr.items() if v is not None)[1] if any(v is not None for v in fr.values()) else None
        return {"kernel": label, "launch_ms": (secs * 1e3) if secs else None, "timed": ("timed region" if prof_keys[0] not in untimed else "extra untimed pass"),
                "bound": best, "hbm": {"achieved": hb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr["hbm"], "traffic": by}, "valu": vb}
    other = {"logic": pass_block("logic", ["logic_fused", "logic"], "logic (+ the inlined material step) + queue scan + scatter: k_logic<FUSE, RAW>, k_queue_scan, k_queue_scatter", False, k_lg),
             "shadow": pass_block("shadow", ["shadow"], "traceShadow (k_shadow4: 4-wide quantised tree, thread per ray)" if not ctx.get_option("shadow_split") else "traceShadow (k_shadow4s: tail-split)", True, k_sh)}
    ext_valu = valu_block(traffic_valu, traffic_lanes, launch_s if launch_s > 0 else None, k_ext if args.extend_tree == 4 else None)
    ext_hbm_frac = (ach / HBM_PEAK_GBS) if ach is not None else None
    ext_bound = "hbm"
    if ext_valu and (ext_hbm_frac is None or ext_valu["issue_frac"] > ext_hbm_frac):
        ext_bound = "valu-issue"
    roofline = {"kernel": ("traceExtension (k_extend4: 4-wide quantised tree)" if not ctx.get_option("refill_extend") else "traceExtension (k_trace4r: 4-wide quantised tree, persistent waves with lane refill)") if args.extend_tree == 4 else "traceExtension (k_extend: binary tree)",
                # `bound`: whichever roof the kernel sits closer to -- "hbm" (achieved / peak / frac below, the contract's fields) or "valu-issue"
                # (roofline.valu: wave-instruction issue, the roof a divergent traversal on a cache-resident tree actually hits; MFMA has no work on this path)
                "bound": ext_bound, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ext_hbm_frac, "traffic": traffic,
                "valu": ext_valu, "other_kernels": other,
                "frac_source": src, "traffic_source": (f"committed capture profiles/traffic_{args.workload}.json (rocprofv3 --pmc passes of scripts/profile_round.sh on this configuration and these kernel sources: capture_key), not measured in this run" if traffic else None),
                "traffic_capture_stale_keys": traffic_stale, "capture_key": key,
                "definition": "achieved = fabric-side bytes (memory-side L2 requests; Infinity-Cache hits included = upper bound of HBM proper) of the extension kernel per launch (profiles/traffic_<workload>.json: rocprofv3 --pmc request counters by size, separate passes, captured on THIS configuration and THESE kernel sources: capture_key) / the kernel's average launch time measured live with HIP events on its stream inside the timed region.  null when no capture matches (traffic_capture_stale_keys says why); frac_own = bytes the running kernel itself touches per ray, an upper bound.  launch_ms: as it runs in the timed region beside the concurrent shadow traversal; launch_ms_alone / frac_alone: same kernel, same steady state, serial schedule (untimed extra pass).",
                "frac_alone": (ach * launch_s / alone_s / HBM_PEAK_GBS) if (ach is not None and alone_s > 0 and launch_s > 0) else None,
                "launch_ms": launch_s * 1e3, "launch_ms_alone": alone_s * 1e3,
                "own_bytes_per_ray": own_bytes_per_ray,
                # bytes the RUNNING kernel touches per ray (own node / leaf / triangle counters) over the launch time: L1 / L2 / Infinity-Cache hits
                # included, so it is a touch rate, not memory traffic -- it exceeds the HBM peak on a cache-resident tree (that is the point of the cache)
                "own_touched_GBps": (own_bytes / launch_s / 1e9) if (own_bytes and launch_s > 0) else None,
                "frac_own": (own_bytes / launch_s / 1e9 / HBM_PEAK_GBS) if (own_bytes and launch_s > 0) else None,
                "own_avg_wide_node_visits": (st_own["ext_inner"] / max(1, st_own["ext_rays"])) if own_leaf is not None else None,
                "own_avg_leaf_visits": (own_leaf["ext_leaf"] / max(1, st_own["ext_rays"])) if own_leaf is not None else None,
                "own_avg_tri_tests": (st_own["ext_tri"] / max(1, st_own["ext_rays"])) if own_leaf is not None else None,
                # the per-CU miss-handling ceiling (scripts/ubench/ta_cost.hip: ~67-70 G distinct lines/s chip-wide when the set is served
                # by the Infinity Cache): 128-byte fabric read requests per ray and per second
                "lines_per_ray": (traffic_lines / rays_per_launch) if (traffic_lines and rays_per_launch) else None,
                "G_lines_per_s": (traffic_lines / launch_s / 1e9) if (traffic_lines and launch_s > 0) else None,
                "contract_bytes_per_ray": bytes_per_ext_ray,
                "contract_equivalent_GBps": achieved,
                "contract_note": "SURVEY 8(d): 84 + 64 n_inner + 40 n_tri + 64 [hit] bytes per ray with the visit counts of the REFERENCE traversal (binary tree, near child first; counted on the same rays) x rays per launch / launch time.  Not a physical bandwidth: the kernel that runs makes half the visits",
                "avg_inner_visits": st["ext_inner"] / max(1, st["ext_rays"]),
                "avg_tri_tests": st["ext_tri"] / max(1, st["ext_rays"]),
                "hit_fraction": st["ext_hits"] / max(1, st["ext_rays"]),
                # from the same PMC capture: SQ_INSTS_VALU per launch and SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU of the kernel that runs
                "valu_instructions_per_launch": traffic_valu, "lanes_per_valu_instruction": traffic_lanes,
                # wave-level trip counts of the thread-per-ray BINARY kernels' counting variants (the reference traversal on the same rays)
                "simd_efficiency_reference_traversal": simd,
                "concurrent_traversal_span_ms": (span_ms / span_n) if span_n else None}
    if rank == 0:
        line = {
            "metric": "Mrays/s (primary+shadow) at 1080p, 8 bounces" if args.workload == "kitchen" else f"Mrays/s (primary+shadow), {args.workload}",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_iterations": settle,
            "ms_per_step": elapsed / args.steps * 1e3,
            "windows": {"count": nwin, "headline": "median window", "Mrays_s": [float(x) for x in win_mrays], "ms_per_step": [float(e / args.steps * 1e3) for e in wins[:, 0]],
                        "spread_pct": float((win_mrays.max() - win_mrays.min()) / win_mrays[med] * 100.0)},
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": ("real (reference asset egyptcat.obj + .mtl + texture)" if args.workload == "egyptcat" else
                                     "synthetic (procedural scene, deterministic); environment map: the reference's assets/env_maps/night.hdr" if int(p["useEnvMap"]) else "synthetic (procedural scene, deterministic)"),
            "config": {"workload": ("kitchen-proc (procedural stand-in for Country Kitchen OBJ), 1920x1080, 8 bounces, env-map MIS under night.hdr, "
                                    "separate material queues") if args.workload == "kitchen" else
                                   ("egyptcat.obj (REAL reference asset, reference benchmark protocol: 1024x1024, start-up parameters, single material queue)" if args.workload == "egyptcat" else args.workload + "-proc"),
                       "width": args.width, "height": args.height, "max_bounces": int(p["maxBounces"]), "triangles": int(d.tris.size),
                       "bvh": WORKLOADS[args.workload][3], "bvh_nodes": int(d.nodes.size), "num_tasks_per_gpu": args.num_tasks, "num_tasks_whole_job": args.num_tasks * world, "scaling_mode": ("weak: --num-tasks paths in flight per rank" if args.scaling == "weak" else f"strong: {num_tasks_total} paths in flight over the whole job, {args.num_tasks} per rank"), "wavefronts_per_gpu": C, "fused_logic_materials": bool(args.fuse), "fused_bsdf_set": ctx.get_option("fuse_set_now") if args.fuse else 0, "ext_order": ctx.get_option("ext_order") if args.fuse else 0,
                       "refill_extend": ctx.get_option("refill_extend"), "refill_shadow": ctx.get_option("refill_shadow"), "shadow_split": ctx.get_option("shadow_split"), "overlap": ctx.get_option("overlap"), "regroup": ctx.get_option("regroup"),
                       "parallelism": f"pixel-interleaved x{world}, no collective in the timed region"},
            "rays": {"primary": prim, "extension": ext, "shadow": sh,
                     "reference_style_total_Mrays_s": (prim + ext + sh) / elapsed / 1e6},
            "kernel_ms_avg": {k: (v[0] / v[1] if v[1] else None) for k, v in prof.items() if v[1]},
            "kernel_ms_avg_source": {"timed_region": sorted(k for k, v in prof.items() if v[1] and k not in untimed), "extra_untimed_pass": sorted(untimed)},
            "roofline": roofline,
        }
        if gather_ms is not None:
            line["gather_ms"] = gather_ms
            line["gather_matches_read_pixels"] = gather_ok
            line["gather_ms_native_rccl"] = native_ms
            line["gather_native_matches_torch"] = native_ok
            line["gather_native_nranks_seen"] = native_seen
            if native_err:
                line["gather_native_error"] = native_err
        if world == 1 and C == 1 and not args.no_ref_gpu_baseline:
            line["ref_gpu_baseline"] = ref_gpu_baseline(ctx, d, p, env, args, settle)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(d, p, env)
            if line["cpu_baseline"]["kind"] == "reference":      # the oracle port beside it (order-preserving appends instead of per-path atomics)
                line["cpu_baseline_port"] = cpu_baseline(d, p, env, budget_s=8.0, force_kind="port")
        print(json.dumps(line), flush=True)
    if use_dist:
        # The native gather (flx_group_init / flx_gather: the code the C++ host runs) failing is a FAILED run: the line above carries the
        # reason, and the exit status must not say "ok".  Hung: a thread is stuck inside RCCL, leave without the collective tear-down.
        native_failed = bool(native_hung or native_err or native_ok is False)
        if native_hung:
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(3)
        dist.barrier()
        dist.destroy_process_group()
        if native_failed:
            sys.stderr.write(f"[bench] rank {rank}: native RCCL gather failed: {native_err}\n")
            sys.exit(3)


if __name__ == "__main__":
    main()
