// flx_device.h -- device-side data layout of libfluctus_hip.so (gfx950 / CDNA4 only).
//
// PATH STATE.  The reference keeps 64 four-byte SoA columns per path (256 B, 11 of them dead
// padding; src/geom.h:199-236) and gathers them one dword at a time through the index queues.
// Here the state is twelve 16-byte records per path ("SoA of float4") + two scalar columns =
// 200 B/path, grouped by which kernel reads/writes them together, so that every access --
// direct (logic: gid = thread) or indirect (queue[gid]) -- is one dwordx4 per lane:
//
//   ORIG   {orig.xyz, lastPdfW}          DIR    {dir.xyz, pathLen}
//   HITP   {P.xyz, t}                    HITN   {N.xyz, flags: bit0 areaLightHit, bit1 backfaceHit}
//   HITUV  {u, v, i, matId}              THR    {T.xyz, seed}
//   EI     {Ei.xyz, pixelIndex}          SHO    {shadowOrig.xyz, shadowRayLen}
//   SHD    {shadowDir.xyz, lastPdfDirect}  LBSDF {lastBsdf.xyz, lastPdfImplicit}
//   LEMIT  {lastEmission.xyz, lastCosTh} LT     {lastT.xyz, lastSpecular}
//   scalars: shadowRayBlocked (u32), lastLightPickProb (f32), firstDiffuseHit (u32, export only)
//
// BVH.  Uploaded in the reference's wire format (48-B nodes with one box each, 160-B triangles),
// re-laid out for traversal: one 64-B record per INNER node holding BOTH child boxes and both
// child references (one cache line per visit instead of 48 B + 2 x 32 B), 48-B position-only
// triangle records in leaf order (leaf = contiguous run, count stored in the run's first record),
// and a 64-B shading record per triangle (normals, uvs, matId) read once per ray after traversal.
// Tree topology, child order and leaf order are the reference's, so ties resolve identically.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fluctus_wire.h"
#include "../../include/flx_math.h"

namespace flxd {

using namespace flx;

enum { S_ORIG = 0, S_DIR, S_HITP, S_HITN, S_HITUV, S_THR, S_EI, S_SHO, S_SHD, S_LBSDF, S_LEMIT, S_LT, S_NUM_REC };

struct State {
    float4 *rec[S_NUM_REC];       // record r of path gid: rec[r] + gid  (pure SoA of float4; a layout of 64-byte lines with four records each
                                  // was measured in round 1 and lost: the coalesced kernels turn into 16-byte accesses at a 64-byte stride)
    __host__ __device__ __forceinline__ float4 *at(int r, uint32_t gid) const { return rec[r] + (size_t)gid; }
    uint32_t *blocked;            // shadowRayBlocked
    float *pickProb;              // lastLightPickProb
    uint32_t *firstDiffuse;       // carried for export parity only
    uint32_t *phase;              // PathPhase of the microkernel integrator (src/geom.h:183-192); untouched by the wavefront path
    uint32_t numTasks;
};

// Block cursors of the persistent traversal kernels (trace4r.hip): one per XCD and kernel, [0..7] closest hit, [8..15] any hit, FLX_CURSOR_STRIDE
// words apart (every cursor is a hot atomic: each gets a cache line and L2 channel of its own).  A wave takes its next 64-ray block from the list of
// its own XCD (blocks x, x + 8, x + 16, ...) and from the next XCD's list when its own is exhausted.  A persistent launch needs them at zero:
// k_end_iteration zeroes them with the counters, flx_clear_queues does when a launch used them since (flx_ctx::cursorDirty), and the
// launch sites (flx_wf_extend / flx_wf_shadow) zero them first whenever cursorDirty says neither happened in between.
#define FLX_NUM_BLOCK_CURSORS 16
#define FLX_CURSOR_STRIDE 64

struct Queues {
    uint32_t *q[FLX_NUM_QUEUES];
    uint32_t *counters;           // 8 x u32 (flx_queue_counters)
    uint32_t *cursors;            // FLX_NUM_BLOCK_CURSORS x FLX_CURSOR_STRIDE u32
    uint32_t extPend;             // lazy extension counter: bit q set = counters[q] entries were appended to the extension queue
                                  // but counters[EXTENSION] has not been bumped yet (see ext_len)
};

#define FLX_LEAF_BIT 0x80000000u

struct BNode {                    // 64 B, 64-B aligned
    float lmin[3], lmax[3];       // left child box
    float rmin[3], rmax[3];       // right child box
    uint32_t left, right;         // inner: BNode index; leaf: FLX_LEAF_BIT | first triangle slot
    uint32_t pad[2];
};

struct TriRec {                   // 48 B: three float4
    float4 a;                     // v0.xyz, triangle index (int bits)
    float4 b;                     // v1.xyz, leaf count in the first record of a leaf run (int bits)
    float4 c;                     // v2.xyz, unused
};

struct ShadeRec {                 // 64 B: four float4
    float4 a;                     // n0.xyz, uv0.x
    float4 b;                     // n1.xyz, uv0.y
    float4 c;                     // n2.xyz, uv1.x
    float4 d;                     // uv1.y, uv2.x, uv2.y, matId (int bits)
};

struct Scene {
    const BNode *bnodes;
    const TriRec *trirecs;
    const ShadeRec *shade;
    const flx_triangle *tris;     // reference-layout triangles (tangent frames for normal maps only)
    const flx_material *materials;
    const flx_texdesc *texdesc;
    const uint8_t *texdata;
    uint32_t rootRef;             // BNode 0 (inner) -- tiny scenes get a synthetic root
    // 4-wide quantised tree over the same leaves (flx_wide.h): 64-B nodes, leaf blocks {exact box, count, triangles}
    const void *wnodes;
    const float4 *wleaf;
    uint32_t wrootRef;
    float wideClamp;              // bound of |1 / dir| in the wide node test: 2^100, or 2^64 for scenes beyond +-2^26 (flx_trace4.h: WRay::setup)
    // environment map
    const float4 *envRGBA;
    const float *pdfTable;
    const float2 *aliasRec;       // {probTable[i], aliasTable[i] (int bits)}: the alias step of sample_env_alias is ONE 8-byte gather at the random index
                                  // instead of two dependent ones (round 5: the light sample's gathers were 18 % of the fused logic pass,
                                  // profiles/r05_logic_probes_ab.txt); built at flx_upload_envmap, values copied bit for bit
    // everything next-event estimation computes for an importance-sampled texel is a function of the texel alone -- the direction of its centre (two
    // sincos + normalize), its solid-angle pdf (a sine, a division), the radiance looked up in that direction (atan2, acos, four texels, the bilinear
    // blend): {L.xyz, pdfW} {Li.xyz, 0} per texel, filled at flx_upload_envmap by a kernel that runs the very device functions the inline code ran
    // (flx_shading.h: env_sample_compute), so the values are the same bit for bit.  Round 5: that code was 16 % of the fused logic pass.
    const float4 *neeRec;
    int envW, envH;
};

struct Frame {                    // framebuffers + cursor + partition
    float *pixels;                // float4 per local pixel (rgb sum, sample count)
    float *preview;
    // denoiser feature buffers (reference: USE_OPTIX_DENOISER build; src/clcontext.cpp:337-338): float4 per local pixel,
    // accumulators + the resolved outputs of `process`; nullptr while the option is off
    float *aovAlbedo, *aovNormal, *aovAlbedoOut, *aovNormalOut;
    uint32_t *currPixelIdx;       // device copy of the pixel cursor
    uint32_t rank, nranks;
    uint32_t localPixels;
};

__device__ __forceinline__ f3 ld3(const float4 &v) { return mk3(v.x, v.y, v.z); }
__device__ __forceinline__ float4 mk4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ float4 mk4u(f3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
__device__ __forceinline__ f3 V(const flx_vec3 &v) { return mk3(v.x, v.y, v.z); }
// first-hit normal in camera space: rotation rows right, up, -dir (src/wf_logic.cl:189-196, src/mk_next_vertex.cl:63-67)
__device__ __forceinline__ f3 camera_space_normal(const flx_render_params &p, f3 N)
{
    const f3 r1 = V(p.camera.right), r2 = V(p.camera.up), r3 = V(p.camera.dir) * -1.0f;
    return mk3(dot(r1, N), dot(r2, N), dot(r3, N));
}

// Path-state accessors.  Every kernel streams the state exactly once per launch, while the BVH is re-read constantly;
// FLX_NT marks state loads (bit 0) / stores (bit 1) non-temporal so they do not evict the tree from L2 / Infinity Cache.
// Round 1 measured nothing from it; on the round-2 kernels (4 M paths = 0.8 GB of state per iteration against a 256 MB Infinity Cache)
// stores alone give +2.4 % Mrays/s on the kitchen, loads + stores +4 % (k_extend4 0.857 -> 0.833 ms, the fused logic pass 0.42 -> 0.40),
// +1 % on conference and courtyard.  The kernels that read what another kernel has JUST written through a queue (genRays' seed, the
// material kernels' records) lose with non-temporal loads and use rd4t.
#ifndef FLX_NT
#define FLX_NT 3
#endif
typedef float flx_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 rd4(const float4 *p)
{
#if FLX_NT & 1
    flx_v4f v = __builtin_nontemporal_load(reinterpret_cast<const flx_v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ float4 rd4t(const float4 *p) { return *p; }          // temporal: the line was written moments ago and is wanted from L2
__device__ __forceinline__ void wr4(float4 *p, float4 v)
{
#if FLX_NT & 2
    flx_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<flx_v4f *>(p));
#else
    *p = v;
#endif
}

__device__ __forceinline__ void wr4t(float4 *p, float4 v) { *p = v; }             // temporal: re-read soon (by this thread or the next kernel)

// REGENERATED PATHS WITHOUT DEAD STORES.  genRays re-initialises the whole path state (src/wf_raygen.cl:77-96): 13 records per
// regenerated path here, 9 of them values that nothing reads before a later kernel overwrites them (the hit record until
// traceExtension; lastBsdf / lastSpecular until the material kernel; lastEmission / lastPdfDirect / shadowRayLen until logic's NEE),
// three of those as 16-byte read-modify-writes for one member.  k_raygen writes only the four records that are live (orig, dir,
// Ei + pixel, T + seed) and the three scalars, and marks the path with two flag bits carried in words it writes anyway:
//   FLX_FRESH in pathLen   (DIR.w)  "no material kernel since regeneration": cleared by the material kernels when they write DIR
//   FLX_FRESH in pixelIndex (EI.w)  "no NEE sample since regeneration":     cleared by logic when it writes a light sample
// and pathLen == 0 means "not extended since regeneration".  flx_state_export substitutes the reference's reset values for the
// members those flags cover, so exported states equal the reference's bit for bit after every kernel (the lockstep tests);
// no kernel ever reads a covered member while its flag is set (each is read only behind shadowRayBlocked == 0 / pathLen > 1).
#define FLX_FRESH 0x80000000u

// wave64 helpers
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// EXTENSION-QUEUE APPENDS WITHOUT ATOMICS.  The reference appends to the extension queue with one
// atomic_inc per work-item (src/wf_raygen.cl:69-70, src/wf_mat_diffuse.cl:64-65; src/utils.cl:328-358).
// A single hot counter saturates at ~88 atomics/us on MI355X, so even one atomic per WAVE cost
// ~150 us per iteration here (13 k waves).  Every appender consumes a compacted source queue whose
// length is already known, so the slot is computed instead: slot = extBase + (lengths of the source
// queues appended before this one in the same call) + index in the own source queue.  extBase is the
// extension counter as left by the previous call; a 1-thread kernel adds the appended total
// afterwards (stream order).  The extension queue thus is the concatenation
// [raygen | diffuse | glossy | ggxRefl | ggxRefr | delta] in source order: deterministic.  (The fused logic + material pass writes the
// material part itself, in these segments or as one list by path id: logic.hip, k_queue_scatter.)
// The bump itself is LAZY: the host remembers which source queues were appended (Queues::extPend) and every kernel
// that needs the extension-queue length adds those lengths on the fly (ext_len); the 1-thread bump kernel
// (k_bump_extension, misc.hip) only runs when someone outside these kernels looks at the counter
// (flx_get_counters_async, flx_wf_logic without a preceding clear, ...).  The steady-state loop never launches it.

__device__ __forceinline__ uint32_t ext_len(const Queues &qs)
{
    uint32_t n = qs.counters[FLX_Q_EXTENSION];
    if (qs.extPend) for (int q = 0; q < FLX_NUM_QUEUES; q++) if (qs.extPend & (1u << q)) n += qs.counters[q];
    return n;
}

} // namespace flxd
