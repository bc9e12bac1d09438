// trace_persistent.hip -- persistent-wave BVH traversal for gfx950 (closest hit and any hit).
//
// Same per-ray arithmetic and visit order as trace.hip (so results are bit-identical); what changes is
// how a 64-lane wave spends its issue slots:
//
//  * persistent waves: the grid is sized to the machine (CUs x resident blocks), each wave pulls rays
//    from the queue with ONE atomic per refill (ballot + prefix ranks), from 8 shard counters -- one per
//    XCD (block b runs on XCD b % 8), each covering a contiguous eighth of the queue -- and steals from
//    the other shards when its own is drained;
//  * while-while scheduling: all lanes descend inner nodes together until every lane sits on a leaf
//    (or is finished), then all lanes intersect their leaves together.  The thread-per-ray kernel pays
//    box-test + triangle-loop cost in nearly every iteration because some lane is always at a leaf
//    (1.2 leaf visits vs 20 inner visits per ray);
//  * refill on under-occupancy: when fewer than `thresh` lanes still own a live ray, finished lanes
//    commit their results and fetch new rays, instead of idling until the slowest lane of a fixed
//    64-ray batch is done.
#include "flx_trace.h"

namespace flxd {

#ifndef PERSIST_WHILE_WHILE
#define PERSIST_WHILE_WHILE 0     // 1: descend-all-then-intersect-all; 0: one node per lane per iteration
#endif
#define RAY_DONE 0xFFFFFFFFu
__device__ __forceinline__ bool is_inner(uint32_t c) { return !(c & FLX_LEAF_BIT); }
__device__ __forceinline__ bool is_leaf(uint32_t c) { return (c & FLX_LEAF_BIT) && c != RAY_DONE; }

struct PersistAux {
    uint32_t *fetch;          // 8 shard counters, zeroed before the launch
    int thresh;               // refill when fewer live lanes than this
};

template <bool ANY_HIT, bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK, TRACE_MIN_WAVES) void k_trace_persistent(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux, PersistAux pa)
{
    __shared__ uint32_t s_stack[LDS_LEVELS * TRACE_BLOCK];
    const int QID = ANY_HIT ? FLX_Q_SHADOW : FLX_Q_EXTENSION;
    const uint32_t qlen = ANY_HIT ? qs.counters[QID] : ext_len(qs);
    const uint32_t *queue = qs.q[QID];
    const uint32_t shardLen = ((qlen + 7u) / 8u + 63u) & ~63u;       // contiguous eighth, wave-aligned

    Stack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.spill = aux.spill + (blockIdx.x * TRACE_BLOCK + threadIdx.x);

    // per-lane ray state
    uint32_t gid = 0, cur = RAY_DONE;
    bool hasRay = false;
    int sp = 0;
    f3 orig = mk3(0.0f), dir = mk3(0.0f), dinv = mk3(0.0f);
    float tbest = 0.0f, ubest = 0.0f, vbest = 0.0f, dirw = 0.0f;
    int tribest = -1;
    bool occluded = false;
    uint32_t nInner = 0, nTri = 0, nHit = 0, nRays = 0;

    // wave-uniform fetch state
    uint32_t curShard = blockIdx.x & 7u;
    int tried = 0;
    int thresh = pa.thresh;

    for (;;) {
        // ---------------------------------------------------------------- commit finished rays
        if (cur == RAY_DONE && hasRay) {
            if (!ANY_HIT) {
                f3 P = mk3(0.0f), N = mk3(0.0f);
                float tu = 0.0f, tv = 0.0f, t = tbest;
                int tri = tribest, matId = -1;
                uint32_t flags = 0;
                if (tri >= 0) {                                      // src/bvh.cl:271-279
                    const float4 *sp4 = reinterpret_cast<const float4 *>(sc.shade + tri);
                    float4 a = sp4[0], b = sp4[1], c = sp4[2], d = sp4[3];
                    P = orig + t * dir;
                    N = normalize(bary(ubest, vbest, ld3(a), ld3(b), ld3(c)));
                    f3 uv = bary(ubest, vbest, mk3(a.w, b.w, 0.0f), mk3(c.w, d.x, 0.0f), mk3(d.y, d.z, 0.0f));
                    tu = uv.x; tv = uv.y;
                    matId = __float_as_int(d.w);
                    if (STATS) nHit++;
                }
                if (p.sampleImpl && p.useAreaLight) {                // src/wf_extrays.cl:28-29
                    if (light_quad(p.areaLight, orig, dir, &t)) {
                        flags = 1u; P = orig + t * dir; N = V(p.areaLight.N); tri = 0; matId = 0;
                    }
                }
                const uint32_t keep = __float_as_uint(reinterpret_cast<const float *>(&st.rec[S_HITN][gid])[3]) & 2u;
                wr4(st.rec[S_DIR] + gid, mk4u(dir, __float_as_uint(dirw) + 1u));
                wr4(st.rec[S_HITP] + gid, mk4(P, t));
                wr4(st.rec[S_HITN] + gid, mk4u(N, flags | keep));
                wr4(st.rec[S_HITUV] + gid, make_float4(tu, tv, __int_as_float(tri), __int_as_float(matId)));
            } else {
                st.blocked[gid] = occluded ? 1u : 0u;
            }
            hasRay = false;
        }
        // ---------------------------------------------------------------- refill idle lanes (wave-uniform loop)
        for (;;) {
            const bool want = !hasRay;
            const uint64_t m = __ballot(want);
            if (m == 0ull || tried >= 8) break;
            const uint32_t n = (uint32_t)__popcll(m);
            const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
            uint32_t base = 0;
            if (lane_id() == leader) base = atomicAdd(&pa.fetch[curShard], n);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
            const uint32_t off = base + mbcnt(m);
            const uint32_t idx = curShard * shardLen + off;
            if (want && off < shardLen && idx < qlen) {
                gid = queue[idx];
                float4 o4, d4;
                if (!ANY_HIT) { o4 = rd4(st.rec[S_ORIG] + gid); d4 = rd4(st.rec[S_DIR] + gid); }
                else { o4 = rd4(st.rec[S_SHO] + gid); d4 = rd4(st.rec[S_SHD] + gid); }
                orig = ld3(o4); dir = ld3(d4); dirw = d4.w;
                dinv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                tbest = ANY_HIT ? o4.w : FLX_FLT_MAX;
                ubest = vbest = 0.0f; tribest = -1; occluded = false;
                sp = 0; cur = sc.rootRef; hasRay = true;
                if (STATS) nRays++;
                if (ANY_HIT && p.useAreaLight) {                     // the light quad itself blocks first (src/wf_shadowrays.cl:32-33)
                    float tl = tbest;
                    if (light_quad(p.areaLight, orig, dir, &tl)) { occluded = true; cur = RAY_DONE; }
                }
            }
            if (base + n >= shardLen || curShard * shardLen + base + n >= qlen) { curShard = (curShard + 1u) & 7u; tried++; }
        }
        if (tried >= 8) thresh = 1;                                  // queue drained: no point leaving the loop early
        if (!__any(hasRay)) break;

        // ---------------------------------------------------------------- traverse
        do {
#if PERSIST_WHILE_WHILE
            while (__any(is_inner(cur))) {
#else
            {
#endif
                if (is_inner(cur)) {
                    const float4 *np = reinterpret_cast<const float4 *>(sc.bnodes + cur);
                    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                    if (STATS) nInner++;
                    const float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
                    const float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
                    const uint32_t left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
                    float lnear, rnear;
                    const bool lh = slab(lmin, lmax, orig, dinv, tbest, &lnear);
                    const bool rh = slab(rmin, rmax, orig, dinv, tbest, &rnear);
                    if (lh && rh) {
                        uint32_t closer = left, farther = right;
                        if (rnear < lnear) { closer = right; farther = left; }
                        stk.push(sp++, farther);
                        cur = closer;
                    } else if (lh) cur = left;
                    else if (rh) cur = right;
                    else cur = (sp == 0) ? RAY_DONE : stk.pop(--sp);
#if !PERSIST_WHILE_WHILE
                    continue;        // single-loop scheduling: one node (inner OR leaf) per lane per iteration
#endif
                }
            }
            if (is_leaf(cur)) {
                const uint32_t slot = cur & ~FLX_LEAF_BIT;
                const float4 *tp = reinterpret_cast<const float4 *>(sc.trirecs + slot);
                float4 a = tp[0], b = tp[1], c = tp[2];
                const int count = __float_as_int(b.w);
                bool stop = false;
                for (int k = 0;;) {
                    if (STATS) nTri++;
                    float t, u, v;
                    if (moller_trumbore(orig, dir, ld3(a), ld3(b), ld3(c), &t, &u, &v) && t > 0.0f && t < tbest) {
                        if (ANY_HIT) { occluded = true; stop = true; break; }
                        tbest = t; ubest = u; vbest = v; tribest = __float_as_int(a.w);
                    }
                    if (++k >= count) break;
                    tp += 3;
                    a = tp[0]; b = tp[1]; c = tp[2];
                }
                cur = (stop || sp == 0) ? RAY_DONE : stk.pop(--sp);
            }
        } while ((int)__popcll(__ballot(cur != RAY_DONE)) >= thresh);
    }

    if (STATS) {
        unsigned long long a = nInner, b = nTri, c = nHit, d = nRays;
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); d += __shfl_xor(d, o, 64); }
        if (lane_id() == 0u) {
            if (!ANY_HIT) { atomicAdd(&aux.stats[0], d); atomicAdd(&aux.stats[1], a); atomicAdd(&aux.stats[2], b); atomicAdd(&aux.stats[3], c); }
            else { atomicAdd(&aux.stats[4], a); atomicAdd(&aux.stats[5], b); atomicAdd(&aux.stats[6], d); }
        }
    }
}

static int g_blocksPerCU[4] = {0, 0, 0, 0};

template <bool ANY_HIT, bool STATS>
static void launch_p(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p,
                     uint32_t *spill, unsigned long long *stats, uint32_t *fetch, int thresh, int numCUs, uint32_t maxBlocks)
{
    int &bpc = g_blocksPerCU[(ANY_HIT ? 2 : 0) + (STATS ? 1 : 0)];
    if (bpc == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_persistent<ANY_HIT, STATS>, TRACE_BLOCK, 0) != hipSuccess || n < 1) n = 4;
        bpc = n;
    }
    uint32_t blocks = (uint32_t)numCUs * (uint32_t)bpc;
    uint32_t need = (st.numTasks + TRACE_BLOCK - 1) / TRACE_BLOCK;
    if (blocks > need) blocks = need;
    if (blocks > maxBlocks) blocks = maxBlocks;
    blocks = (blocks + 7u) & ~7u;                                   // whole shards
    if (blocks > maxBlocks) blocks = maxBlocks & ~7u;
    if (blocks == 0) blocks = maxBlocks < 8 ? maxBlocks : 8;
    (void)hipMemsetAsync(fetch, 0, 8 * sizeof(uint32_t), s);
    TraceAux aux{spill, blocks * TRACE_BLOCK, stats};
    PersistAux pa{fetch, thresh};
    hipLaunchKernelGGL((k_trace_persistent<ANY_HIT, STATS>), dim3(blocks), dim3(TRACE_BLOCK), 0, s, st, qs, sc, p, aux, pa);
}

void launch_extend_persistent(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p,
                              uint32_t *spill, unsigned long long *stats, uint32_t *fetch, int thresh, int numCUs, uint32_t maxBlocks)
{
    if (stats) launch_p<false, true>(s, st, qs, sc, p, spill, stats, fetch, thresh, numCUs, maxBlocks);
    else launch_p<false, false>(s, st, qs, sc, p, spill, stats, fetch, thresh, numCUs, maxBlocks);
}
void launch_shadow_persistent(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p,
                              uint32_t *spill, unsigned long long *stats, uint32_t *fetch, int thresh, int numCUs, uint32_t maxBlocks)
{
    if (stats) launch_p<true, true>(s, st, qs, sc, p, spill, stats, fetch, thresh, numCUs, maxBlocks);
    else launch_p<true, false>(s, st, qs, sc, p, spill, stats, fetch, thresh, numCUs, maxBlocks);
}

} // namespace flxd
