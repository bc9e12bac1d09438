// trace4.hip -- closest-hit and any-hit kernels on the 4-wide quantised tree (flx_wide.h, flx_trace4.h).
//
// Same contract as trace.hip's k_extend / k_shadow (reference kernels traceExtension, src/wf_extrays.cl:5-36, and
// traceShadow, src/wf_shadowrays.cl:6-38): one ray per lane from the extension / shadow queue, results into the hit
// record / shadowRayBlocked.  The tree has the reference's leaves under collapsed inner levels:
//  * k_shadow4 is bit-identical to the reference's bvh_occluded (order-free query; the argument is in flx_wide.h);
//  * k_extend4 visits the leaves in a different order than bvh_intersect, which can only matter for exact ties in t and
//    for box-vs-triangle rounding near-ties; tests/test_gpu_wide.py counts those flips against the oracle.
// One wave per block, traversal stack in LDS [level][lane] as in trace.hip.
#include "flx_trace4.h"

namespace flxd {

#ifndef WIDE_MIN_WAVES
#define WIDE_MIN_WAVES 1
#endif

template <bool STATS>
__global__ __launch_bounds__(WIDE_BLOCK, WIDE_MIN_WAVES) void k_extend4(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux)
{
    __shared__ uint32_t s_stack[WIDE_LDS_LEVELS * WIDE_BLOCK];
    const uint32_t qlen = ext_len(qs);
    const uint32_t idx = blockIdx.x * WIDE_BLOCK + threadIdx.x;
    if (idx >= qlen) return;
    const uint32_t gid = qs.q[FLX_Q_EXTENSION][idx];

    const float4 o4 = rd4(st.at(S_ORIG, gid));
    const float4 d4 = rd4(st.at(S_DIR, gid));
    const f3 orig = ld3(o4), dir = ld3(d4);

    WStack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.spill = aux.spill + idx;
    stk.base = 0;

    float t = FLX_FLT_MAX, u = 0.0f, v = 0.0f;
    int tri = -1;
    uint32_t nInner = 0, nTri = 0, nLeaf = 0;
    traverse4<false, STATS>(sc, stk, orig, dir, t, u, v, tri, nInner, nTri, nLeaf, STATS ? aux.stats + 8 : nullptr);

    uint32_t flags; int matId;
    commit_hit(st, sc, p, gid, orig, dir, d4.w, t, u, v, tri, flags, matId);

    if (STATS) {
        bool hitGeom = matId >= 0 && !(flags & 1u);
        unsigned long long a = nInner, b = nTri, c = hitGeom ? 1ull : 0ull, l = nLeaf;
        uint32_t mx = nInner;
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); l += __shfl_xor(l, o, 64); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64)); }
        uint64_t act = __ballot(true);
        if (lane_id() == (uint32_t)__ffsll((long long)act) - 1u) {
            atomicAdd(&aux.stats[7], (unsigned long long)mx);
            atomicAdd(&aux.stats[0], (unsigned long long)__popcll(act));
            atomicAdd(&aux.stats[1], a); atomicAdd(&aux.stats[2], b); atomicAdd(&aux.stats[3], c);
            atomicAdd(&aux.stats[16], l);
        }
    }
}

template <bool STATS, int ANY_ORDER>
__global__ __launch_bounds__(WIDE_BLOCK, WIDE_MIN_WAVES) void k_shadow4(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux)
{
    __shared__ uint32_t s_stack[WIDE_LDS_LEVELS * WIDE_BLOCK];
    const uint32_t qlen = qs.counters[FLX_Q_SHADOW];
    // (the shadow queue holds ~3/4 of numTasks, so ~15 k of the 65 k waves exit here at once; a capped grid whose waves stride over the queue
    //  was measured instead -- 0.315 -> 0.387 ms, two blocks in a row per wave lengthen the tail -- profiles/r03_shadow_grid_cap_ab.txt)
    const uint32_t idx = blockIdx.x * WIDE_BLOCK + threadIdx.x;
    if (idx >= qlen) return;
    const uint32_t gid = qs.q[FLX_Q_SHADOW][idx];

    const float4 o4 = rd4(st.at(S_SHO, gid));
    const float4 d4 = rd4(st.at(S_SHD, gid));
    const f3 orig = ld3(o4), dir = ld3(d4);
    const float lenL = o4.w;

    WStack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.spill = aux.spill + idx;
    stk.base = 0;

    // the area-light quad itself blocks first (reference: src/wf_shadowrays.cl:32-33)
    bool occluded = false;
    uint32_t nInner = 0, nTri = 0, nLeaf = 0;
    if (p.useAreaLight) { float tl = lenL; occluded = light_quad(p.areaLight, orig, dir, &tl); }
    if (!occluded) {
        float t = lenL, u, v; int tri;
        occluded = traverse4<true, STATS, ANY_ORDER>(sc, stk, orig, dir, t, u, v, tri, nInner, nTri, nLeaf, STATS ? aux.stats + 12 : nullptr);
    }
    st.blocked[gid] = occluded ? 1u : 0u;

    if (STATS) {
        unsigned long long a = nInner, b = nTri, l = nLeaf;
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); l += __shfl_xor(l, o, 64); }
        uint64_t act = __ballot(true);
        if (lane_id() == (uint32_t)__ffsll((long long)act) - 1u) {
            atomicAdd(&aux.stats[4], a); atomicAdd(&aux.stats[5], b);
            atomicAdd(&aux.stats[6], (unsigned long long)__popcll(act));
            atomicAdd(&aux.stats[17], l);
        }
    }
}

void launch_extend4(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill, unsigned long long *stats)
{
    uint32_t blocks = (st.numTasks + WIDE_BLOCK - 1) / WIDE_BLOCK;
    TraceAux aux{spill, blocks * WIDE_BLOCK, stats};
    if (stats) hipLaunchKernelGGL(k_extend4<true>, dim3(blocks), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux);
    else hipLaunchKernelGGL(k_extend4<false>, dim3(blocks), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux);
}

void launch_shadow4(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill, unsigned long long *stats)
{
    uint32_t blocks = (st.numTasks + WIDE_BLOCK - 1) / WIDE_BLOCK;
    TraceAux aux{spill, blocks * WIDE_BLOCK, stats};
    // visit order of the any-hit traversal (flx_trace4.h: ANY_ORDER): far -> near when every shadow ray runs toward the environment light
    const bool farFirst = p.useEnvMap && !p.useAreaLight;
    if (stats) {
        if (farFirst) hipLaunchKernelGGL((k_shadow4<true, 1>), dim3(blocks), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux);
        else hipLaunchKernelGGL((k_shadow4<true, 0>), dim3(blocks), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux);
    } else {
        if (farFirst) hipLaunchKernelGGL((k_shadow4<false, 1>), dim3(blocks), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux);
        else hipLaunchKernelGGL((k_shadow4<false, 0>), dim3(blocks), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux);
    }
}

} // namespace flxd
