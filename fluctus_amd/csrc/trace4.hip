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

// ---- tail splitting of the any-hit query (flx_trace4.h: traverse4_any_budget).  FIRST: one lane per shadow-queue entry, as k_shadow4, with the
// first budget.  !FIRST: one lane per continuation record of the previous pass.  A continuation record is 64 B: {queue index, sp, FLX_SPLIT_KEEP stack
// words}.  Suspended rays are appended with ONE atomic per wave -- but not to one counter: a single hot address serves ~88 atomics / us on this chip
// (flx_device.h), and with nearly every wave of the first pass suspending a ray that alone was 1 ms per launch (first version, round 5).  The record
// array is cut into FLX_SPLIT_LISTS sub-lists with a counter each, 64 B apart; wave w appends to list w mod FLX_SPLIT_LISTS, the next pass runs one
// lane per record SLOT of every sub-list and the lanes beyond a list's count leave at once.  Without `out` (last pass) the budget is unlimited.
#define FLX_SPLIT_LISTS 256
#define FLX_SPLIT_COUNT_STRIDE 16          // words between two sub-list counters
struct SplitAux {
    const uint32_t *inCount; const uint4 *inRec; uint32_t inSubCap, inLimit;      // !FIRST: records to resume: FLX_SPLIT_LISTS sub-lists of inSubCap slots, inLimit of them usable
    uint32_t *outCount; uint4 *outRec; uint32_t outSubCap, outLimit;     // where suspended rays go (nullptr: none may suspend); outLimit <= outSubCap slots per list are used
    int budget;
    uint32_t *zero;                                                       // FIRST: the counter set of the NEXT launch, zeroed here (launch_shadow4_split)
};

template <int ANY_ORDER, bool FIRST>
__global__ __launch_bounds__(WIDE_BLOCK, WIDE_MIN_WAVES) void k_shadow4s(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux, SplitAux sa)
{
    __shared__ uint32_t s_stack[WIDE_LDS_LEVELS * WIDE_BLOCK];
    const uint32_t i = blockIdx.x * WIDE_BLOCK + threadIdx.x;
    uint32_t idx;                                                     // position in the shadow queue
    int sp = 0;
    uint32_t cur = sc.wrootRef;
    WStack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.base = 0;
    if (FIRST) {
        if (i < 2u * FLX_SPLIT_LISTS) for (uint32_t k = i; k < 2u * FLX_SPLIT_LISTS; k += gridDim.x * WIDE_BLOCK) sa.zero[k * FLX_SPLIT_COUNT_STRIDE] = 0u;      // (a grid of fewer than 8 waves: strided)
        if (i >= qs.counters[FLX_Q_SHADOW]) return;
        idx = i;
    } else {
        const uint32_t list = i / sa.inSubCap, within = i - list * sa.inSubCap;      // (inSubCap is a multiple of the wave size: one list per wave)
        uint32_t n = sa.inCount[list * FLX_SPLIT_COUNT_STRIDE]; n = n < sa.inLimit ? n : sa.inLimit;
        if (within >= n) return;
        const uint4 *rec = sa.inRec + (size_t)i * 4;
        const uint4 h = rec[0];
        idx = h.x; sp = (int)h.y;
        // the stack: 14 words behind the two header words; unconditional 16-byte loads, the entries beyond sp are never popped
        const uint4 a = rec[1], b = rec[2], c = rec[3];
        stk.slot(0) = h.z; stk.slot(1) = h.w;
        stk.slot(2) = a.x; stk.slot(3) = a.y; stk.slot(4) = a.z; stk.slot(5) = a.w;
        stk.slot(6) = b.x; stk.slot(7) = b.y; stk.slot(8) = b.z; stk.slot(9) = b.w;
        stk.slot(10) = c.x; stk.slot(11) = c.y; stk.slot(12) = c.z; stk.slot(13) = c.w;
        cur = stk.pop(sp);                                            // the node the ray stood on when it suspended
    }
    stk.spill = aux.spill + idx;                                      // the ray's own spill column, whichever lane traces it
    const uint32_t gid = qs.q[FLX_Q_SHADOW][idx];
    const float4 o4 = rd4(st.at(S_SHO, gid));
    const float4 d4 = rd4(st.at(S_SHD, gid));
    const f3 orig = ld3(o4), dir = ld3(d4);
    const float lenL = o4.w;

    int status = 0;
    // the area-light quad itself blocks first (reference: src/wf_shadowrays.cl:32-33)
    if (FIRST && p.useAreaLight) { float tl = lenL; if (light_quad(p.areaLight, orig, dir, &tl)) status = 1; }
    WRay r;
    if (status == 0) { r.setup(orig, dir, sc.wideClamp); status = traverse4_any_budget<ANY_ORDER>(sc, stk, r, lenL, sp, cur, sa.outRec ? sa.budget : 0x7fffffff); }

    // suspended rays: one slot each, one atomic per wave (the lanes still alive here) on the wave's sub-list counter
    const uint64_t m = __ballot(status == 2);
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t list = blockIdx.x & (FLX_SPLIT_LISTS - 1);
        uint32_t base = 0;
        if ((int)lane_id() == leader) base = atomicAdd(sa.outCount + list * FLX_SPLIT_COUNT_STRIDE, (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
        if (status == 2) {
            const uint32_t slot = base + mbcnt(m);
            if (slot < sa.outLimit) {
                uint4 *rec = sa.outRec + ((size_t)list * sa.outSubCap + slot) * 4;
                rec[0] = make_uint4(idx, (uint32_t)sp, stk.slot(0), stk.slot(1));
                rec[1] = make_uint4(stk.slot(2), stk.slot(3), stk.slot(4), stk.slot(5));
                rec[2] = make_uint4(stk.slot(6), stk.slot(7), stk.slot(8), stk.slot(9));
                rec[3] = make_uint4(stk.slot(10), stk.slot(11), stk.slot(12), stk.slot(13));
            } else {                                                  // the sub-list is full: finish here
                cur = stk.pop(sp);
                status = traverse4_any_budget<ANY_ORDER>(sc, stk, r, lenL, sp, cur, 0x7fffffff);
            }
        }
    }
    if (status != 2) st.blocked[gid] = status == 1 ? 1u : 0u;
}

// budgets: k1 for the pass over the queue, k2 for a second pass over the suspended rays (0: that pass finishes them), a last pass without budget.
// counts: 2 SETS of 2 x FLX_SPLIT_LISTS counters, FLX_SPLIT_COUNT_STRIDE words apart; recA / recB: FLX_SPLIT_LISTS x subCapA / subCapB records of 64 B.
// The launches alternate between the two counter sets (`parity`): the first pass zeroes the OTHER set -- the previous launch's, complete in stream
// order -- for the next launch.  (A hipMemsetAsync in front of the first pass was a fill kernel of 0.12 ms on the shadow stream beside genRays.)
void launch_shadow4_split(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill,
                          uint32_t *counts, uint32_t parity, uint4 *recA, uint32_t subCapA, uint4 *recB, uint32_t subCapB, int k1, int k2, uint32_t limit)
{
    const uint32_t blocks = (st.numTasks + WIDE_BLOCK - 1) / WIDE_BLOCK;
    TraceAux aux{spill, blocks * WIDE_BLOCK, nullptr};
    const bool farFirst = p.useEnvMap && !p.useAreaLight;
    const uint32_t setWords = 2u * FLX_SPLIT_LISTS * FLX_SPLIT_COUNT_STRIDE;
    uint32_t *cA = counts + (parity & 1u) * setWords, *cB = cA + FLX_SPLIT_LISTS * FLX_SPLIT_COUNT_STRIDE, *other = counts + ((parity & 1u) ^ 1u) * setWords;
    const uint32_t limA = limit && limit < subCapA ? limit : subCapA, limB = limit && limit < subCapB ? limit : subCapB;      // (test hook: force the full-list path)
    const SplitAux a1{nullptr, nullptr, 0u, 0u, cA, recA, subCapA, limA, k1, other};
    const SplitAux a2{cA, recA, subCapA, limA, k2 > 0 ? cB : nullptr, k2 > 0 ? recB : nullptr, subCapB, limB, k2, nullptr};
    const SplitAux a3{cB, recB, subCapB, limB, nullptr, nullptr, 0u, 0u, 0, nullptr};
    const dim3 b(WIDE_BLOCK), g1(blocks), g2(FLX_SPLIT_LISTS * (subCapA / WIDE_BLOCK)), g3(FLX_SPLIT_LISTS * (subCapB / WIDE_BLOCK));
    if (farFirst) {
        hipLaunchKernelGGL((k_shadow4s<1, true>), g1, b, 0, s, st, qs, sc, p, aux, a1);
        hipLaunchKernelGGL((k_shadow4s<1, false>), g2, b, 0, s, st, qs, sc, p, aux, a2);
        if (k2 > 0) hipLaunchKernelGGL((k_shadow4s<1, false>), g3, b, 0, s, st, qs, sc, p, aux, a3);
    } else {
        hipLaunchKernelGGL((k_shadow4s<0, true>), g1, b, 0, s, st, qs, sc, p, aux, a1);
        hipLaunchKernelGGL((k_shadow4s<0, false>), g2, b, 0, s, st, qs, sc, p, aux, a2);
        if (k2 > 0) hipLaunchKernelGGL((k_shadow4s<0, false>), g3, b, 0, s, st, qs, sc, p, aux, a3);
    }
}
uint32_t shadow_split_lists() { return FLX_SPLIT_LISTS; }
uint32_t shadow_split_count_words() { return 2u * 2u * FLX_SPLIT_LISTS * FLX_SPLIT_COUNT_STRIDE; }

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
