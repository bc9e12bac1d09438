// flx_trace.h -- ray/box, ray/triangle and traversal-stack primitives shared by the trace kernels.
// Arithmetic: include/flx_math.h contract (bit-identical to the oracle).
#pragma once
#include "flx_device.h"

namespace flxd {

// tunables (overridable with -D for A/B builds; defaults are the measured best)
#ifndef TRACE_BLOCK
#define TRACE_BLOCK 64           // threads per block of the trace kernels (1 wave: LDS is released wave by wave)
#endif
#ifndef LDS_LEVELS
#define LDS_LEVELS 16            // traversal-stack levels kept in LDS (4 KiB/wave -> 7 waves/SIMD at 72 VGPRs); deeper levels spill to global memory
#endif
#ifndef TRACE_MIN_WAVES
#define TRACE_MIN_WAVES 1        // __launch_bounds__ 2nd argument (min waves per SIMD -> VGPR cap)
#endif
#ifndef SHADOW_MIN_WAVES
#define SHADOW_MIN_WAVES 1         // 8 fits (64 VGPRs, no spill) but measures the same as 7
#endif
#define MAX_LEVELS 64

struct TraceAux {
    uint32_t *spill;        // (MAX_LEVELS - LDS_LEVELS) x totalThreads
    uint32_t totalThreads;
    unsigned long long *stats;   // 16 counters (7 ray-level, [8..11] / [12..15] wave-level trips of k_extend / k_shadow) or nullptr
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

// What traceExtension writes for one ray once the closest triangle is known (reference: src/wf_extrays.cl:22-35): the
// shading attributes of the winning triangle interpolated at (u, v) (src/bvh.cl:271-279; the reference re-interpolates at
// every commit, only the last one is observable), then the implicit area-light hit (src/wf_extrays.cl:28-29,
// src/intersect.cl:124-155), pathLen += 1 and the 12 hit columns.  Shared by the binary and the 4-wide kernels.
// hit_values computes, commit_hit stores.
struct HitVals { f3 P, N; float tu, tv, t; int tri, matId; uint32_t flags; };

template <bool LIGHT_QUAD = true>
__device__ __forceinline__ HitVals hit_values(const Scene &sc, const flx_render_params &p, f3 orig, f3 dir, float t, float u, float v, int tri)
{
    HitVals h;
    h.P = mk3(0.0f); h.N = mk3(0.0f);
    h.tu = 0.0f; h.tv = 0.0f;
    h.matId = -1;
    h.flags = 0;
    if (tri >= 0) {
        const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
        float4 a = sp[0], b = sp[1], c = sp[2], d = sp[3];
        h.P = orig + t * dir;
        h.N = normalize(bary(u, v, ld3(a), ld3(b), ld3(c)));
        f3 uv = bary(u, v, mk3(a.w, b.w, 0.0f), mk3(c.w, d.x, 0.0f), mk3(d.y, d.z, 0.0f));
        h.tu = uv.x; h.tv = uv.y;
        h.matId = __float_as_int(d.w);
    }
    if (LIGHT_QUAD && p.sampleImpl && p.useAreaLight) {          // (LIGHT_QUAD false: the caller applies the quad itself, or the scene has none)
        if (light_quad(p.areaLight, orig, dir, &t)) {
            h.flags = 1u;
            h.P = orig + t * dir;
            h.N = V(p.areaLight.N);
            tri = 0; h.matId = 0;
        }
    }
    h.t = t; h.tri = tri;
    return h;
}

// the two bits of HITN.w: bit 0 areaLightHit comes from the commit; bit 1 backfaceHit belongs to `logic` and the reference's
// traceExtension leaves it untouched -- except that genRays clears it: a regenerated path (pathLen still 0) must not inherit the bit of
// the slot's previous path (flx_device.h)
__device__ __forceinline__ uint32_t hit_keep_flags(float pathLenBits, uint32_t oldHitNw)
{
    const bool first = (__float_as_uint(pathLenBits) & ~FLX_FRESH) == 0u && (__float_as_uint(pathLenBits) & FLX_FRESH) != 0u;
    return first ? 0u : (oldHitNw & 2u);
}

template <bool LIGHT_QUAD = true>
__device__ __forceinline__ void commit_hit(const State &st, const Scene &sc, const flx_render_params &p, uint32_t gid, f3 orig, f3 dir, float pathLenBits,
                                           float t, float u, float v, int tri, uint32_t &flags, int &matId)
{
    const HitVals h = hit_values<LIGHT_QUAD>(sc, p, orig, dir, t, u, v, tri);
    flags = h.flags; matId = h.matId;
    wr4(st.at(S_DIR, gid), mk4u(dir, __float_as_uint(pathLenBits) + 1u));          // pathLen += 1
    wr4(st.at(S_HITP, gid), mk4(h.P, h.t));
    const uint32_t keep = hit_keep_flags(pathLenBits, __float_as_uint(reinterpret_cast<const float *>(st.at(S_HITN, gid))[3]));
    wr4(st.at(S_HITN, gid), mk4u(h.N, h.flags | keep));
    wr4(st.at(S_HITUV, gid), make_float4(h.tu, h.tv, __int_as_float(h.tri), __int_as_float(h.matId)));
}

// RAW HIT RECORDS (round 3).  The persistent-wave closest-hit kernel (trace4r.hip) does not commit: a finished lane stores
//     HITUV = {u, v, FLX_RAW | (triangle + 1), t}
// and the commit -- implicit area-light quad included -- happens where the record is consumed anyway: in the fused logic pass of the next
// iteration (logic.hip: k_logic<FUSE, RAW>), which has the ray in registers and needs the shading attributes next -- or, when anything else
// wants to look first (a read-back, the separate kernels, the microkernels ...), in k_materialise (api.hip: settle).  A committed record
// holds the hit index there (>= -1), so bits 31:30 == 01 marks a raw one unambiguously for scenes below 2^30 triangles.
#define FLX_RAW       0x40000000u
#define FLX_RAW_TRI   0x3FFFFFFFu
__device__ __forceinline__ bool hit_is_raw(uint32_t z) { return (z & 0xC0000000u) == FLX_RAW; }
__device__ __forceinline__ HitVals hit_values_raw(const Scene &sc, const flx_render_params &p, f3 orig, f3 dir, float4 raw)
{
    return hit_values<true>(sc, p, orig, dir, raw.w, raw.x, raw.y, (int)(__float_as_uint(raw.z) & FLX_RAW_TRI) - 1);
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

// One ray through the tree, reference visit order (near child first, leaf triangles in index order).
//
// Loop shape ("while-while"): the inner `while` descends AND pops, so a lane leaves it only when it stands on a leaf or
// has finished; the wave then intersects all pending leaves together.  The obvious single loop (if inner / else leaf /
// pop at the bottom) is structurised by the compiler into descend-only inner loops with the pop in the outer loop:
// every lane that dead-ends waits for the deepest descent of the wave -- measured 95 inner trips per wave against 76
// for this shape and 46.5 for the wave's longest ray (bench.py roofline.simd_efficiency).
#define FLX_RAY_DONE 0xFFFFFFFFu                  // leaf bit set: never taken for an inner node

template <bool ANY_HIT, bool STATS>
__device__ __forceinline__ bool traverse(const Scene &sc, Stack &stk, f3 orig, f3 dir, float &tbest, float &ubest, float &vbest,
                                         int &tribest, uint32_t &nInner, uint32_t &nTri, unsigned long long *wstats = nullptr)
{
    // STATS builds only: wave-level trip counts (SIMD efficiency = lane-level work / (64 x wave-level trips)):
    // wstats[0] outer iterations, [1] executions of the inner-node branch, [2] of the leaf branch, [3] triangle-loop trips
#define FLX_WAVE_TICK(k) do { if (STATS && wstats) { const uint64_t m_ = __ballot(true); \
        if (lane_id() == (uint32_t)__ffsll((long long)m_) - 1u) atomicAdd(&wstats[k], 1ull); } } while (0)
    const f3 dinv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    int sp = 0;
    uint32_t cur = sc.rootRef;
    for (;;) {
        FLX_WAVE_TICK(0);
        while (!(cur & FLX_LEAF_BIT)) {
            FLX_WAVE_TICK(1);
            const float4 *np = reinterpret_cast<const float4 *>(sc.bnodes + cur);
            const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            const float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
            const float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
            const uint32_t left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
            if (STATS) nInner++;
            float lnear, rnear;
            const bool lh = slab(lmin, lmax, orig, dinv, tbest, &lnear);
            const bool rh = slab(rmin, rmax, orig, dinv, tbest, &rnear);
            if (lh && rh) {
                const bool goRight = rnear < lnear;
                stk.push(sp++, goRight ? left : right);
                cur = goRight ? right : left;
            } else if (lh || rh) {
                cur = lh ? left : right;
            } else {
                cur = (sp == 0) ? FLX_RAY_DONE : stk.pop(sp - 1);
                sp = (sp == 0) ? 0 : sp - 1;
            }
        }
        if (cur == FLX_RAY_DONE) break;
        {
            const uint32_t slot = cur & ~FLX_LEAF_BIT;
            const float4 *tp = reinterpret_cast<const float4 *>(sc.trirecs + slot);
            float4 a = tp[0], b = tp[1], c = tp[2];
            const int count = __float_as_int(b.w);
            FLX_WAVE_TICK(2);
            for (int k = 0;;) {
                FLX_WAVE_TICK(3);
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
#undef FLX_WAVE_TICK
}


} // namespace flxd
