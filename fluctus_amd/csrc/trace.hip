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
#include "flx_device.h"

namespace flxd {

#define TRACE_BLOCK 256
#define LDS_LEVELS 32
#define MAX_LEVELS 64

struct TraceAux {
    uint32_t *spill;        // (MAX_LEVELS - LDS_LEVELS) x totalThreads
    uint32_t totalThreads;
    unsigned long long *stats;   // 7 counters or nullptr
};

__device__ __forceinline__ bool slab(const float *bmin, const float *bmax, f3 orig, f3 dinv, float tMaxPrev, float *tnear)
{
    f3 tmp = (mk3(bmin[0], bmin[1], bmin[2]) - orig) * dinv;
    f3 tmaxv = (mk3(bmax[0], bmax[1], bmax[2]) - orig) * dinv;
    f3 tminv = min3(tmp, tmaxv);
    tmaxv = max3(tmp, tmaxv);
    float tmin = fmaxf_(fmaxf_(tminv.x, tminv.y), tminv.z);
    float tmax = fminf_(fminf_(tmaxv.x, tmaxv.y), tmaxv.z);
    *tnear = tmin;
    if (tmax < 0.0f) return false;
    if (tmin > tmax) return false;
    return tmin < tMaxPrev;
}

__device__ __forceinline__ bool moller_trumbore(f3 orig, f3 dir, f3 p0, f3 p1, f3 p2, float *tret, float *uret, float *vret)
{
    f3 s1 = p1 - p0;
    f3 s2 = p2 - p0;
    f3 pvec = cross(dir, s2);
    float det = dot(s1, pvec);
    if (absf(det) < 1e-12f) return false;
    float iDet = 1.0f / det;
    f3 tvec = orig - p0;
    float u = dot(tvec, pvec) * iDet;
    if (u < 0.0f || u > 1.0f) return false;
    f3 qvec = cross(tvec, s1);
    float v = dot(dir, qvec) * iDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(s2, qvec) * iDet;
    if (t < 0.0f) return false;
    *tret = t; *uret = u; *vret = v;
    return true;
}

// area-light quad as two triangles; updates *t (reference: src/intersect.cl:96-155)
__device__ __forceinline__ bool light_quad(const flx_arealight &L, f3 orig, f3 dir, float *t)
{
    if (dot(dir, V(L.N)) > 0.0f) return false;
    f3 pos = V(L.pos), right = V(L.right), up = V(L.up);
    f3 tl = pos + L.size.x * right + L.size.y * up;
    f3 tr = pos - L.size.x * right + L.size.y * up;
    f3 bl = pos + L.size.x * right - L.size.y * up;
    f3 br = pos - L.size.x * right - L.size.y * up;
    bool hit = false;
    float tt, u, v;
    if (moller_trumbore(orig, dir, tl, bl, br, &tt, &u, &v) && !(tt > *t)) { *t = tt; hit = true; }
    if (moller_trumbore(orig, dir, tl, br, tr, &tt, &u, &v) && !(tt > *t)) { *t = tt; hit = true; }
    return hit;
}

struct Stack {
    uint32_t *lds;          // this thread's column: lds[level * TRACE_BLOCK]
    uint32_t *spill;        // this thread's column: spill[(level - LDS_LEVELS) * totalThreads]
    uint32_t stride;
    __device__ __forceinline__ void push(int level, uint32_t v)
    {
        if (level < LDS_LEVELS) lds[level * TRACE_BLOCK] = v;
        else spill[(size_t)(level - LDS_LEVELS) * stride] = v;
    }
    __device__ __forceinline__ uint32_t pop(int level)
    {
        return level < LDS_LEVELS ? lds[level * TRACE_BLOCK] : spill[(size_t)(level - LDS_LEVELS) * stride];
    }
};

template <bool ANY_HIT, bool STATS>
__device__ __forceinline__ bool traverse(const Scene &sc, Stack &stk, f3 orig, f3 dir, float &tbest, float &ubest, float &vbest,
                                         int &tribest, uint32_t &nInner, uint32_t &nTri)
{
    const f3 dinv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    int sp = 0;
    uint32_t cur = sc.rootRef;
    for (;;) {
        if (!(cur & FLX_LEAF_BIT)) {
            const float4 *np = reinterpret_cast<const float4 *>(sc.bnodes + cur);
            float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            if (STATS) nInner++;
            float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
            float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
            uint32_t left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
            float lnear, rnear;
            bool lh = slab(lmin, lmax, orig, dinv, tbest, &lnear);
            bool rh = slab(rmin, rmax, orig, dinv, tbest, &rnear);
            if (lh && rh) {
                uint32_t closer = left, farther = right;
                if (rnear < lnear) { closer = right; farther = left; }
                stk.push(sp++, farther);
                cur = closer;
                continue;
            } else if (lh) { cur = left; continue; }
            else if (rh) { cur = right; continue; }
        } else {
            uint32_t slot = cur & ~FLX_LEAF_BIT;
            const float4 *tp = reinterpret_cast<const float4 *>(sc.trirecs + slot);
            float4 a = tp[0], b = tp[1], c = tp[2];
            int count = __float_as_int(b.w);
            for (int k = 0;;) {
                if (STATS) nTri++;
                float t, u, v;
                if (moller_trumbore(orig, dir, ld3(a), ld3(b), ld3(c), &t, &u, &v) && t > 0.0f && t < tbest) {
                    if (ANY_HIT) return true;
                    tbest = t; ubest = u; vbest = v; tribest = __float_as_int(a.w);
                }
                if (++k >= count) break;
                tp += 3;
                a = tp[0]; b = tp[1]; c = tp[2];
            }
        }
        if (sp == 0) break;
        cur = stk.pop(--sp);
    }
    return false;
}

// XCD-aware block -> queue-chunk mapping: consecutive blocks are dispatched round-robin over the 8
// XCDs (each with its own 4 MiB L2); give every XCD a CONTIGUOUS 1/8th of the ray queue so rays that
// are neighbours in the queue (neighbouring pixels / paths) share one L2's view of the BVH.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblocks, int enable)
{
    if (!enable) return b;
    uint32_t per = nblocks >> 3;
    if (per == 0 || b >= (per << 3)) return b;
    return (b & 7u) * per + (b >> 3);
}

template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK) void k_extend(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux, int xcdRemap)
{
    __shared__ uint32_t s_stack[LDS_LEVELS * TRACE_BLOCK];
    const uint32_t qlen = qs.counters[FLX_Q_EXTENSION];
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x, xcdRemap);
    const uint32_t idx = blk * TRACE_BLOCK + threadIdx.x;
    if (idx >= qlen) return;
    const uint32_t gid = qs.q[FLX_Q_EXTENSION][idx];

    const float4 o4 = st.rec[S_ORIG][gid];
    const float4 d4 = st.rec[S_DIR][gid];
    const f3 orig = ld3(o4), dir = ld3(d4);

    Stack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.spill = aux.spill + (blockIdx.x * TRACE_BLOCK + threadIdx.x);

    float t = FLX_FLT_MAX, u = 0.0f, v = 0.0f;
    int tri = -1;
    uint32_t nInner = 0, nTri = 0;
    traverse<false, STATS>(sc, stk, orig, dir, t, u, v, tri, nInner, nTri);

    // commit: shading attributes of the winning triangle (reference: src/bvh.cl:271-279)
    f3 P = mk3(0.0f), N = mk3(0.0f);
    float tu = 0.0f, tv = 0.0f;
    int matId = -1;
    uint32_t flags = 0;
    if (tri >= 0) {
        const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
        float4 a = sp[0], b = sp[1], c = sp[2], d = sp[3];
        P = orig + t * dir;
        N = normalize(bary(u, v, ld3(a), ld3(b), ld3(c)));
        f3 uv = bary(u, v, mk3(a.w, b.w, 0.0f), mk3(c.w, d.x, 0.0f), mk3(d.y, d.z, 0.0f));
        tu = uv.x; tv = uv.y;
        matId = __float_as_int(d.w);
    }
    // implicit area-light hit (reference: src/wf_extrays.cl:28-29, src/intersect.cl:124-155)
    if (p.sampleImpl && p.useAreaLight) {
        if (light_quad(p.areaLight, orig, dir, &t)) {
            flags = 1u;
            P = orig + t * dir;
            N = V(p.areaLight.N);
            tri = 0; matId = 0;
        }
    }
    st.rec[S_DIR][gid] = mk4u(dir, __float_as_uint(d4.w) + 1u);          // pathLen += 1
    st.rec[S_HITP][gid] = mk4(P, t);
    // backfaceHit (bit 1) belongs to `logic`; the reference's kernel leaves it untouched
    const uint32_t keep = __float_as_uint(reinterpret_cast<const float *>(&st.rec[S_HITN][gid])[3]) & 2u;
    st.rec[S_HITN][gid] = mk4u(N, flags | keep);
    st.rec[S_HITUV][gid] = make_float4(tu, tv, __int_as_float(tri), __int_as_float(matId));

    if (STATS) {
        bool hitGeom = matId >= 0 && !(flags & 1u);
        unsigned long long a = nInner, b = nTri, c = hitGeom ? 1ull : 0ull;
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
        uint64_t act = __ballot(true);
        if (lane_id() == (uint32_t)__ffsll((long long)act) - 1u) {
            atomicAdd(&aux.stats[0], (unsigned long long)__popcll(act));
            atomicAdd(&aux.stats[1], a); atomicAdd(&aux.stats[2], b); atomicAdd(&aux.stats[3], c);
        }
    }
}

template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK) void k_shadow(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux, int xcdRemap)
{
    __shared__ uint32_t s_stack[LDS_LEVELS * TRACE_BLOCK];
    const uint32_t qlen = qs.counters[FLX_Q_SHADOW];
    const uint32_t blk = xcd_remap(blockIdx.x, gridDim.x, xcdRemap);
    const uint32_t idx = blk * TRACE_BLOCK + threadIdx.x;
    if (idx >= qlen) return;
    const uint32_t gid = qs.q[FLX_Q_SHADOW][idx];

    const float4 o4 = st.rec[S_SHO][gid];
    const float4 d4 = st.rec[S_SHD][gid];
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
        occluded = traverse<true, STATS>(sc, stk, orig, dir, t, u, v, tri, nInner, nTri);
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
