// trace.hip -- closest-hit (traceExtension) and any-hit (traceShadow) BVH traversal for gfx950.
//
// Replaces reference kernels traceExtension (src/wf_extrays.cl:5-36 -> bvh_intersect,
// src/bvh.cl:234-310) and traceShadow (src/wf_shadowrays.cl:6-38 -> bvh_occluded,
// src/bvh.cl:312-373), with intersectAABB / intersectTriangle / intersectLight of
// src/intersect.cl:41-93,124-155.
//
// Design (not a translation of the OpenCL kernels):
//  * one 64-B BNode per inner visit carries both child boxes (see flx_device.h);
//  * the traversal stack lives in LDS, laid out [level][lane] so a wave's push/pop is one
//    conflict-free ds_write_b32/ds_read_b32 (the reference's `uint stack[64]` is private memory,
//    i.e. scratch on a wave64 machine); levels >= LDS_LEVELS spill to a global side buffer;
//  * the current node is kept in a register ("push farther, continue with closer"), which is the
//    same visit order as the reference's push-both-pop-one;
//  * triangles are 48-B position-only records in leaf order; normals/uvs/matId are fetched once
//    per ray after traversal from the 64-B shading record of the winning triangle.  The reference
//    re-interpolates at every commit; only the last commit is observable, so results are equal.
//  * arithmetic is the contract of include/flx_math.h: slab test as (box - orig) * (1/dir),
//    Moller-Trumbore with EPSILON 1e-12, no FMA contraction -> bit-identical to the oracle.
#include "flx_trace.h"

namespace flxd {

template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK, TRACE_MIN_WAVES) void k_extend(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux, int xcdRemap)
{
    __shared__ uint32_t s_stack[LDS_LEVELS * TRACE_BLOCK];
    const uint32_t qlen = ext_len(qs);
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x, xcdRemap);
    const uint32_t idx = blk * TRACE_BLOCK + threadIdx.x;
    if (idx >= qlen) return;
    const uint32_t gid = qs.q[FLX_Q_EXTENSION][idx];

    const float4 o4 = rd4(st.at(S_ORIG, gid));
    const float4 d4 = rd4(st.at(S_DIR, gid));
    const f3 orig = ld3(o4), dir = ld3(d4);

    Stack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.spill = aux.spill + (blockIdx.x * TRACE_BLOCK + threadIdx.x);

    float t = FLX_FLT_MAX, u = 0.0f, v = 0.0f;
    int tri = -1;
    uint32_t nInner = 0, nTri = 0;
    traverse<false, STATS>(sc, stk, orig, dir, t, u, v, tri, nInner, nTri, STATS ? aux.stats + 8 : nullptr);

    uint32_t flags; int matId;
    commit_hit(st, sc, p, gid, orig, dir, d4.w, t, u, v, tri, flags, matId);

    if (STATS) {
        bool hitGeom = matId >= 0 && !(flags & 1u);
        unsigned long long a = nInner, b = nTri, c = hitGeom ? 1ull : 0ull;
        uint32_t mx = nInner;                                           // longest ray of the wave (lower bound of the wave's trip count)
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64)); }
        uint64_t act = __ballot(true);
        if (lane_id() == (uint32_t)__ffsll((long long)act) - 1u) {
            atomicAdd(&aux.stats[7], (unsigned long long)mx);
            atomicAdd(&aux.stats[0], (unsigned long long)__popcll(act));
            atomicAdd(&aux.stats[1], a); atomicAdd(&aux.stats[2], b); atomicAdd(&aux.stats[3], c);
        }
    }
}

template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK, SHADOW_MIN_WAVES) void k_shadow(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux, int xcdRemap)
{
    __shared__ uint32_t s_stack[LDS_LEVELS * TRACE_BLOCK];
    const uint32_t qlen = qs.counters[FLX_Q_SHADOW];
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x, xcdRemap);
    const uint32_t idx = blk * TRACE_BLOCK + threadIdx.x;
    if (idx >= qlen) return;
    const uint32_t gid = qs.q[FLX_Q_SHADOW][idx];

    const float4 o4 = rd4(st.at(S_SHO, gid));
    const float4 d4 = rd4(st.at(S_SHD, gid));
    const f3 orig = ld3(o4), dir = ld3(d4);
    float lenL = o4.w;

    Stack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.spill = aux.spill + (blockIdx.x * TRACE_BLOCK + threadIdx.x);

    // the area-light quad itself blocks first (reference: src/wf_shadowrays.cl:32-33)
    bool occluded = false;
    uint32_t nInner = 0, nTri = 0;
    if (p.useAreaLight) { float tl = lenL; occluded = light_quad(p.areaLight, orig, dir, &tl); }
    if (!occluded) {
        float t = lenL, u, v; int tri;
        occluded = traverse<true, STATS>(sc, stk, orig, dir, t, u, v, tri, nInner, nTri, STATS ? aux.stats + 12 : nullptr);
    }
    st.blocked[gid] = occluded ? 1u : 0u;

    if (STATS) {
        unsigned long long a = nInner, b = nTri;
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        uint64_t act = __ballot(true);
        if (lane_id() == (uint32_t)__ffsll((long long)act) - 1u) {
            atomicAdd(&aux.stats[4], a); atomicAdd(&aux.stats[5], b);
            atomicAdd(&aux.stats[6], (unsigned long long)__popcll(act));
        }
    }
}

void launch_extend(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p,
                   uint32_t *spill, unsigned long long *stats, int xcdRemap)
{
    uint32_t blocks = (st.numTasks + TRACE_BLOCK - 1) / TRACE_BLOCK;
    TraceAux aux{spill, blocks * TRACE_BLOCK, stats};
    if (stats) hipLaunchKernelGGL(k_extend<true>, dim3(blocks), dim3(TRACE_BLOCK), 0, s, st, qs, sc, p, aux, xcdRemap);
    else hipLaunchKernelGGL(k_extend<false>, dim3(blocks), dim3(TRACE_BLOCK), 0, s, st, qs, sc, p, aux, xcdRemap);
}

void launch_shadow(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p,
                   uint32_t *spill, unsigned long long *stats, int xcdRemap)
{
    uint32_t blocks = (st.numTasks + TRACE_BLOCK - 1) / TRACE_BLOCK;
    TraceAux aux{spill, blocks * TRACE_BLOCK, stats};
    if (stats) hipLaunchKernelGGL(k_shadow<true>, dim3(blocks), dim3(TRACE_BLOCK), 0, s, st, qs, sc, p, aux, xcdRemap);
    else hipLaunchKernelGGL(k_shadow<false>, dim3(blocks), dim3(TRACE_BLOCK), 0, s, st, qs, sc, p, aux, xcdRemap);
}

} // namespace flxd
