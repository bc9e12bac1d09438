/*
 * fluctus_hip.h -- C ABI of libfluctus_hip.so, the MI355X (gfx950) implementation of the
 * wavefront path-tracing hot path.
 *
 * This is the drop-in boundary: every entry point replaces one method of the reference's device
 * context `class CLContext` (reference: src/clcontext.hpp:26-211, src/clcontext.cpp), the only
 * object `Tracer` talks to for device work.  Plain pointers and sizes; no C++ or torch types.
 * All functions return 0 on success and non-zero on failure (message: flx_last_error()); nothing
 * throws or exits across the boundary (reference behaviour: clt::check() aborts,
 * src/clcontext.cpp:931-936).
 *
 * Threading/ordering (same contract as the reference's single in-order cl::CommandQueue,
 * src/clcontext.cpp:25-29): one host thread per context; every flx_wf_*, flx_clear_queues,
 * flx_get_counters_async, flx_set_params, flx_pixel_index_* call is ASYNCHRONOUS and executes in
 * issue order on the context's HIP stream; flx_finish() is the only synchronisation point
 * (uploads and the read-back helpers are blocking).  Inputs are copied; the caller keeps ownership.
 *
 * Wire formats: include/fluctus_wire.h.
 */
#ifndef FLUCTUS_HIP_H
#define FLUCTUS_HIP_H

#include <stdint.h>
#include <stddef.h>
#include "fluctus_wire.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct flx_ctx flx_ctx;

/* CLContext::CLContext + setup (src/clcontext.cpp:18-69): pick HIP device `device`, create the
 * stream, allocate path state for `num_tasks` paths (Settings wfBufferSize, src/settings.cpp:20)
 * and the 8 index queues + counters (src/clcontext.cpp:116-141). */
int flx_create(int device, uint32_t num_tasks, flx_ctx **out);
int flx_destroy(flx_ctx *ctx);
/* last error text of this context (or of flx_create when ctx == NULL) */
const char *flx_last_error(flx_ctx *ctx);

/* CLContext::uploadSceneData + packTextures (src/clcontext.cpp:522-611).  Takes the reference's
 * wire arrays (160-B triangles, u32 index list, 48-B nodes, 80-B materials, 12-B texture
 * descriptors + RGBA8 blob) and re-lays them out for CDNA4 on the device. */
int flx_upload_scene(flx_ctx *ctx, const void *tris, size_t ntris, const uint32_t *indices, size_t nidx,
                     const void *nodes, size_t nnodes, const void *materials, size_t nmat,
                     const void *texdesc, size_t ntex, const uint8_t *texdata, size_t texbytes);

/* CLContext::createEnvMap (src/clcontext.cpp:467-511): RGB float image + alias/prob/pdf tables. */
int flx_upload_envmap(flx_ctx *ctx, const float *rgb, int w, int h, const float *prob, const int *alias, const float *pdf);

/* CLContext::updateParams (src/clcontext.cpp:703-707), 240-byte RenderParams.  A change of
 * width*height re-allocates the framebuffers (CLContext::setupPixelStorage/resizeBuffers). */
int flx_set_params(flx_ctx *ctx, const void *render_params_240);

/* enqueueWfResetKernel / RaygenKernel / ExtRayKernel / ShadowRayKernel / LogicKernel /
 * MaterialKernels (src/clcontext.cpp:765-848).  Kernels: src/wf_reset.cl, wf_raygen.cl,
 * wf_extrays.cl, wf_shadowrays.cl, wf_logic.cl, wf_mat_*.cl.
 * All of them enqueue and return, as the reference's do.  With option "fuse" (default on, see flx_set_option) flx_wf_logic and a
 * flx_wf_raygen directly behind it are enqueued by the NEXT call on the context instead -- together with the material kernels as one
 * pass when that call is flx_wf_materials -- which no other call can tell from immediate launches except by timing. */
int flx_wf_reset(flx_ctx *ctx);
int flx_wf_raygen(flx_ctx *ctx);
int flx_wf_extend(flx_ctx *ctx);
int flx_wf_shadow(flx_ctx *ctx);
int flx_wf_logic(flx_ctx *ctx, int first_iteration);
int flx_wf_materials(flx_ctx *ctx);           /* dispatches on params.wfSeparateQueues */

/* enqueueClearWfQueues (src/clcontext.cpp:877-883) */
int flx_clear_queues(flx_ctx *ctx);
/* enqueueGetCounters (src/clcontext.cpp:668-671): asynchronous 32-byte read; *out32 (a
 * flx_queue_counters) is valid only after the next flx_finish(). */
int flx_get_counters_async(flx_ctx *ctx, void *out32);
/* finishQueue (src/clcontext.cpp:885-889) */
int flx_finish(flx_ctx *ctx);
/* updatePixelIndex / resetPixelIndex (src/clcontext.cpp:891-901) */
int flx_pixel_index_update(flx_ctx *ctx, uint32_t num_pixels, uint32_t num_new_paths);
int flx_pixel_index_reset(flx_ctx *ctx);
/* Device-side end of a benchmark-style iteration (no counterpart in the reference, which needs a
 * host round trip per iteration to move the cursor, src/tracer.cpp:456-462): in stream order,
 * (1) add the 8 queue counters to 64-bit running totals, (2) cursor = (cursor + raygenQueue) %
 * local pixels -- the value updatePixelIndex would write -- and (3) zero the counters
 * (enqueueClearWfQueues).  Lets a caller run K iterations with a single flx_finish(). */
int flx_end_iteration_async(flx_ctx *ctx);
/* running totals accumulated by flx_end_iteration_async (8 x u64, flx_queue_counters order); blocking */
int flx_counter_totals(flx_ctx *ctx, uint64_t *out8, int reset);
/* getNumTasks (src/clcontext.cpp:903-906) */
uint32_t flx_num_tasks(flx_ctx *ctx);

/* enqueuePostprocessKernel (src/clcontext.cpp:750-763; kernel src/mk_postprocess.cl) -- headless:
 * the preview buffer is a plain device buffer, no GL interop. */
int flx_postprocess(flx_ctx *ctx);
/* saveImage's read-back half (src/clcontext.cpp:386-465): which = 0 raw accumulation (rgb sum,
 * sample count), 1 = post-processed preview.  Blocking; out = float4 per (local) pixel.
 * With flx_set_option(ctx, "denoiser", 1) -- the reference's USE_OPTIX_DENOISER kernel build (recompileKernels,
 * src/clcontext.cpp:852-874; buffers :337-338) -- `logic` / the microkernels also accumulate the denoiser feature
 * buffers and flx_postprocess resolves them: which = 2 albedo of the first non-singular hit, 3 first-hit normal in
 * camera space (both as written to denoiserAlbedoGL / denoiserNormalGL, src/mk_postprocess.cl:49-54), 4 / 5 the raw
 * accumulators (sum, count). */
int flx_read_pixels(flx_ctx *ctx, int which, float *out_rgba);

/* ---- microkernel integrator (the reference's second integrator).  One path per pixel (needs
 * num_tasks >= width*height to cover the image), `phase` state machine, exactly one sample per pixel per pass.
 * enqueueResetKernel / RayGenKernel / NextVertexKernel / BsdfSampleKernel / SplatKernel / SplatPreviewKernel
 * (src/clcontext.cpp:709-750; kernels src/mk_reset.cl, mk_raygen.cl, mk_next_vertex.cl, mk_sample_bsdf.cl, mk_splat.cl,
 * mk_splat_preview.cl).  Single-GPU (the pixel partition applies to the wavefront path only). */
int flx_mk_reset(flx_ctx *ctx);
int flx_mk_raygen(flx_ctx *ctx);
int flx_mk_next_vertex(flx_ctx *ctx);
int flx_mk_sample_bsdf(flx_ctx *ctx);
int flx_mk_splat(flx_ctx *ctx);
int flx_mk_splat_preview(flx_ctx *ctx);
/* fetchStatsAsync / resetStats (src/clcontext.cpp:634-646): RenderStats {primaryRays, extensionRays, shadowRays, samples}
 * (4 x u32, src/geom.h:254-260) accumulated on the device by the microkernels; *out16 valid after flx_finish() */
int flx_mk_stats_async(flx_ctx *ctx, void *out16);
int flx_mk_stats_reset(flx_ctx *ctx);

/* ---- multi-GPU (no counterpart in the reference: one cl::CommandQueue, one device).
 * Rank r of R owns global pixels p*R + r; its framebuffer holds ceil((w*h - r)/R) local pixels. */
int flx_set_partition(flx_ctx *ctx, uint32_t rank, uint32_t nranks);
uint32_t flx_local_pixels(flx_ctx *ctx);
/* device-to-device copy of the raw accumulation buffer (local pixels, float4) into a caller-owned
 * device buffer (e.g. a torch tensor handed to an RCCL gather); asynchronous on the context stream */
int flx_copy_pixels_to_device(flx_ctx *ctx, void *dst_device_ptr);
/* the context's hipStream_t, for callers that order their own work after ours */
void *flx_stream(flx_ctx *ctx);

/* ---- multi-GPU group and gather over RCCL / xGMI (SURVEY 8(b) flx_create_group / flx_gather, 8(e)).  librccl.so.1 is bound with
 * dlopen at the first call, so single-GPU users never load it.  Two ways to form the group:
 *   one process per GPU (torchrun / MPI style): rank 0 calls flx_group_unique_id and ships the 128 bytes to the other ranks by
 *     whatever channel the launcher has; every rank then calls flx_group_init(ctx, rank, nranks, id) (= ncclCommInitRank +
 *     flx_set_partition), renders, and all ranks call flx_gather(ctx, root, out) -- a collective; `out` (width*height float4,
 *     host) is written on the root only.
 *   one process, N contexts on N devices (the reference's single-process Tracer driving a node): flx_group_init_local(ctxs, n)
 *     (= ncclCommInitAll + partitions 0..n-1) and flx_gather_local(ctxs, n, root, out).  If several of the contexts sit on the SAME
 *     device (a 1-GPU box standing in for N ranks) the tiles travel by device-to-device copies instead of RCCL, everything else
 *     (partition, staging, de-interleave) is the same code.
 * Each rank sends its compact float4[local pixels] accumulation tile (rgb sum, sample count) point-to-point to the root
 * (grouped ncclSend / ncclRecv); the root de-interleaves global pixel p*R + r <- tile r, pixel p.  Blocking. */
#define FLX_GROUP_ID_BYTES 128
int flx_group_unique_id(void *out128);
int flx_group_init(flx_ctx *ctx, uint32_t rank, uint32_t nranks, const void *id128);
int flx_group_init_local(flx_ctx **ctxs, uint32_t n);
int flx_gather(flx_ctx *ctx, uint32_t root, float *out_rgba_host);
int flx_gather_local(flx_ctx **ctxs, uint32_t n, uint32_t root, float *out_rgba_host);
int flx_group_destroy(flx_ctx *ctx);
/* what the communicator itself reports: out2 = {ncclCommCount, ncclCommUserRank} (a same-device local group reports its partition) */
int flx_group_info(flx_ctx *ctx, uint32_t *out2);

/* ---- measurement.  Per-kernel HIP-event timing on the context's stream (the reference attaches
 * cl::Events to the two trace kernels, src/clcontext.cpp:673-701,780,786).  kernel ids: */
enum { FLX_K_RESET = 0, FLX_K_RAYGEN = 1, FLX_K_EXTEND = 2, FLX_K_SHADOW = 3, FLX_K_LOGIC = 4, FLX_K_MATERIALS = 5,
       FLX_K_POSTPROCESS = 6,
       FLX_K_TRACE_SPAN = 7,   /* start of the extension kernel .. end of the (concurrent) shadow kernel */
       FLX_K_LOGIC_FUSED = 8,  /* logic + the inlined material step as one pass (option "fuse"); FLX_K_MATERIALS then covers the rest */
       FLX_K_COUNT = 9 };
/* on: 0 off | 1 time every kernel | 2 time only the two trace kernels (+ their span), as the reference does | 3 only the
 * extension kernel | 4 the three kernels bench.py prices against a roof: extension, logic (the fused pass incl. its queue scan + scatter), shadow.
 * Each event pair costs a few microseconds of stream time, which shows at ~11 launches per 0.7 ms
 * iteration: level 1 costs 7 % of the throughput, level 3 about 1.5 % (at 1 M paths; at the bench's 8 M a quarter of that). */
int flx_profile_enable(flx_ctx *ctx, int on);
/* after flx_finish(): accumulated milliseconds and launch count since the last reset */
int flx_profile_get(flx_ctx *ctx, int kernel, double *total_ms, uint64_t *launches);
int flx_profile_reset(flx_ctx *ctx);
/* traversal work counters (algorithmic-bytes model): when enabled the trace kernels
 * accumulate {ext rays, ext inner-node visits, ext triangle tests, ext hits, shadow inner,
 * shadow triangle tests, shadow rays}.  Counting launches are not used inside timed regions. */
int flx_trace_stats_enable(flx_ctx *ctx, int on);
int flx_trace_stats_get(flx_ctx *ctx, uint64_t *out7);
/* the same 7 counters + out16[8..11] / [12..15]: wave-level trip counts of the extension / shadow traversal
 * {outer iterations, inner-node branch executions, leaf branch executions, triangle-loop trips}; SIMD efficiency of
 * the traversal = lane-level visits / (64 x wave-level trips) */
int flx_trace_stats_get_ex(flx_ctx *ctx, uint64_t *out16);
/* the 16 above + out24[16] / [17]: leaf visits of the extension / shadow traversal ([18..23] reserved) */
#define FLX_NUM_TRACE_STATS 24
int flx_trace_stats_get_all(flx_ctx *ctx, uint64_t *out24);
int flx_trace_stats_reset(flx_ctx *ctx);
/* what flx_upload_scene built: out8 = {wide nodes, wide leaf data (16-byte units), wide traversal-stack bound, boxes nested (1/0),
 * binary tree depth, spill levels per lane, binary inner-node records, largest leaf} */
int flx_scene_info(flx_ctx *ctx, uint32_t *out8);

/* ---- test hooks: path state in the reference's GPUTaskState SoA layout (64 columns x num_tasks
 * words, src/geom.h:199-236), queues and counters.  Blocking. */
int flx_state_export(flx_ctx *ctx, float *out_64xN);
/* the arithmetic contract (include/flx_math.h) evaluated on the device over n operand pairs, results as bit patterns.  fn: 0 sin, 1 cos,
 * 2 tan, 3 atan2(a, b), 4 acos, 5 pow(a, b), 6 log, 7 exp, 8 asin, 9 atan, 10 fmin, 11 fmax, 12 a / b, 13 sqrt, 14 a * b, 15 a + b.
 * The oracle's orc_math_array is the CPU half: both sides must agree bit for bit, signed zeros included. */
int flx_math_probe(flx_ctx *ctx, int fn, const float *a, const float *b, uint32_t n, uint32_t *out_bits);
/* test hook: the per-texel light-sample table the library builds at flx_upload_envmap (8 floats per texel: {L.xyz, pdfW, Li.xyz, 0}).
 * The oracle's orc_env_sample_table computes the same values inline, as the reference's logic kernel does per path (src/wf_logic.cl:236-249). */
int flx_env_sample_table(flx_ctx *ctx, float *out_8_per_texel);
int flx_state_import(flx_ctx *ctx, const float *in_64xN);
int flx_queue_read(flx_ctx *ctx, int queue, uint32_t *out_N);
int flx_queue_write(flx_ctx *ctx, int queue, const uint32_t *in, uint32_t n);
int flx_set_counters(flx_ctx *ctx, const void *in32);
/* options.  Unknown names fail.
 *   shadow_tree       4 (default): flx_wf_shadow walks the 4-wide quantised tree built at upload over the reference tree's leaves
 *                     (csrc/flx_wide.h) -- bit-identical results to the binary tree (order-free query, conservative boxes, exact
 *                     leaf test) | 2: the reference's binary tree
 *   extend_tree       4 (default): flx_wf_extend on the 4-wide tree -- same closest hit except where the visit ORDER decides
 *                     (exact ties in t, box-vs-triangle rounding near-ties; SURVEY 8(c) budgets <= 1e-5 of rays, measured in
 *                     DESIGN.md 4.1) | 2: the reference's binary tree in the reference's visit order (bit-exact)
 *   overlap           0 serial | 1 flx_wf_shadow directly after flx_wf_extend runs concurrently with it on a second stream |
 *                     2 as 1, and it starts as soon as `logic` is done when only raygen / materials / extend
 *                     were enqueued since flx_wf_logic (see flx_wf_shadow in api.hip) | -1 (default) = 2; flx_get_option returns the
 *                     effective value
 *   refill_extend     the closest-hit query on the 4-wide tree as a PERSISTENT kernel (csrc/trace4r.hip): value = refillMin | waitMax << 8
 *                     -- lanes that finished take new rays when refillMin lanes are idle; a descent round ends when waitMax lanes
 *                     stand on a leaf.  Default 16 | 32 << 8; 0 = the thread-per-ray kernel.  Bit-identical results.  The kernel leaves
 *                     RAW hit records; they are committed by the next fused logic pass, or by a separate pass as soon as any call that
 *                     could observe a hit record is made -- no call sees a raw one
 *   refill_shadow     the same for the any-hit query; -1 (default) = 0: two persistent kernels cannot share the machine, so the any-hit
 *                     kernel stays thread-per-ray and fills the slots the persistent closest-hit kernel's waves leave (api.hip: pickSchedule)
 *   fuse              1 (default): flx_wf_logic is DEFERRED -- launched by the next call on this context; when that call is
 *                     flx_wf_materials (a flx_wf_raygen between the two is deferred along and launched right after), logic and the
 *                     material step of the most common BSDF types run as ONE pass over the path state (logic.hip: k_logic<FUSED>),
 *                     the other types through their queues as usual; when it is anything else, the plain kernel runs first.  No call
 *                     can observe a state, queue or counter the separate kernels would not have produced (but see ext_order for
 *                     the ORDER of the extension queue); an error raised by a deferred launch is reported by the call that launched it.  Needs the
 *                     material queues empty (flx_clear_queues / flx_end_iteration_async since the last logic) and wfSeparateQueues
 *                     (or a build that inlines every type) -- otherwise, and with 0, every call launches its own kernels at once
 *   fuse_set          BSDF types the fused pass inlines: 1 diffuse only | 31 all six.  flx_upload_scene picks it from the scene
 *                     (diffuse surfaces >= 3/4 of the triangle area: 1, else 31); set it after the upload to override.  OVERRIDDEN while
 *                     params.wfSeparateQueues == 0: with a single material queue every BSDF type sits in the diffuse list and only the pass
 *                     that inlines all types can serve it, so that pass runs whatever fuse_set says -- flx_get_option("fuse_set") still returns
 *                     the stored value, the read-only "fuse_set_now" the set the next fused pass will really inline (an A/B of fuse_set under a
 *                     single material queue compares the all-types pass with itself)
 *   regen_prep        1 (default) | 0: inside the fused chain logic -> genRays -> materials the RAW logic pass computes, for the paths it terminates, the half
 *                     of genRays that is a function of the path's seed alone (jitter, thin-lens origin, throughput + the seed genRays leaves,
 *                     shadowRayBlocked, lastLightPickProb) and stores it with its full-line stores; the genRays launch of the same chain then stores
 *                     only the direction and the pixel (logic.hip / misc.hip: PREPARED REGENERATION).  Bit-identical results; no call can run
 *                     between the two launches
 *   regroup           the all-types fused pass hands its material step through LDS sorted by BSDF type, so that a wave runs one type
 *                     (logic.hip: LOGIC_REGROUP; 96 VGPRs + 20 KB of LDS per block): 0 | 1 | -1 (default) = on; flx_get_option returns the
 *                     effective value.
 *                     Bit-identical results.  Takes effect when the path count is a multiple of 256 (whole blocks)
 *   ext_order         how the fused pass lists the traced paths in the extension queue: 0 one segment per material queue, in the
 *                     separate kernels' order | 1 all continuing paths by path id | 2 continuing AND regenerated paths merged into one
 *                     list by path id (genRays then does not append; needs genRays between logic and the material kernels, else as 1).
 *                     The same SET of paths either way; the reference's order is whatever its atomic_inc produces.  flx_upload_scene picks
 *                     it with fuse_set (1 with 31, 2 with 1); set it afterwards to override
 *   node_layout       1 (default) sibling-pair record numbering of the binary tree | 0 DFS numbering; takes effect at the next flx_upload_scene
 *   denoiser          1: accumulate the denoiser feature buffers (see flx_read_pixels); default 0
 *   xcd_remap, eager_bump: A/B knobs of the binary kernels */
int flx_set_option(flx_ctx *ctx, const char *name, int value);
/* current value of an option above, or of the read-only "fused_queue_mask" (bit q set = the fused pass inlines the material step of
 * queue q's paths, flx_queue_counters order), or of the read-only "phase": the state of the call-sequence state machine behind the
 * deferred / fused / early-started kernels (api.hip: enum Phase) -- bits 0-2: 0 idle, 1 flx_wf_logic deferred, 2 flx_wf_logic +
 * flx_wf_raygen deferred, 3 only genRays / material kernels enqueued since logic, 4 ... and the extension kernel last, 5 the extension
 * kernel last with the chain since logic broken; bit 3: the hit records of the last extension launch are still RAW; bit 4: the material
 * queues are known to be empty.  Never changes any state (tests/test_gpu_fuzz.py reports its coverage with it). */
int flx_get_option(flx_ctx *ctx, const char *name, int *value);

#ifdef __cplusplus
}
#endif
#endif /* FLUCTUS_HIP_H */
