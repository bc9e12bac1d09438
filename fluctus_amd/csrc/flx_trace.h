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
#ifndef TRACE_COMPACT
#define TRACE_COMPACT 0           // 1: lossless 32-byte compact node records for nodes entered straight from their parent (2 loads instead
                                 // of 4).  Bit-exact, but measured 5 % SLOWER (78 VGPRs -> 6 waves/SIMD, decode ALU, two load paths); A/B only
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
// Loop shape ("while-while", TRACE_LOOP 1): the inner `while` descends AND pops, so a lane leaves it only when it stands
// on a leaf or has finished; the wave then intersects all pending leaves together.  The obvious single loop
// (TRACE_LOOP 0: if inner / else leaf / pop at the bottom) is structurised by the compiler into descend-only inner
// loops with the pop in the outer loop: every lane that dead-ends waits for the deepest descent of the wave --
// measured 95 inner trips per wave against 46.5 for the wave's longest ray (bench.py roofline.simd_efficiency).
#ifndef TRACE_LOOP
#define TRACE_LOOP 1
#endif
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
#if TRACE_LOOP == 1
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
#else
#if TRACE_COMPACT
    bool boxKnown = false;
    float pmin[3] = {0.0f, 0.0f, 0.0f}, pmax[3] = {0.0f, 0.0f, 0.0f};
#endif
    for (;;) {
        FLX_WAVE_TICK(0);
        if (!(cur & FLX_LEAF_BIT)) {
            FLX_WAVE_TICK(1);
            float lmin[3], lmax[3], rmin[3], rmax[3];
            uint32_t left, right;
#if TRACE_COMPACT
            if (boxKnown) {
                // Reached straight from the parent, whose test just produced this node's own box (pmin, pmax): a 32-byte
                // record suffices.  The union of the two child boxes IS the parent box, so on every one of the six faces at
                // least one child carries the parent's plane; the record stores only the other child's plane per face plus
                // two ownership bits (in the spare high bits of the child references).  Lossless: the 12 decoded floats are
                // bit-identical to the full record's, 2 vector loads instead of 4.
                const float4 *cp = reinterpret_cast<const float4 *>(sc.cnodes + cur);
                const float4 c0 = cp[0], c1 = cp[1];
                const uint32_t lr = __float_as_uint(c1.z), rr = __float_as_uint(c1.w);
                const float in[6] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
                const float par[6] = {pmin[0], pmin[1], pmin[2], pmax[0], pmax[1], pmax[2]};
                float lp[6], rp[6];
#pragma unroll
                for (int f = 0; f < 6; f++) {
                    lp[f] = (lr & (1u << (CREF_FLAG_SHIFT + f))) ? par[f] : in[f];
                    rp[f] = (rr & (1u << (CREF_FLAG_SHIFT + f))) ? par[f] : in[f];
                }
                lmin[0] = lp[0]; lmin[1] = lp[1]; lmin[2] = lp[2]; lmax[0] = lp[3]; lmax[1] = lp[4]; lmax[2] = lp[5];
                rmin[0] = rp[0]; rmin[1] = rp[1]; rmin[2] = rp[2]; rmax[0] = rp[3]; rmax[1] = rp[4]; rmax[2] = rp[5];
                left = (lr & FLX_LEAF_BIT) | (lr & CREF_INDEX_MASK); right = (rr & FLX_LEAF_BIT) | (rr & CREF_INDEX_MASK);
            } else
#endif
            {
                const float4 *np = reinterpret_cast<const float4 *>(sc.bnodes + cur);
                const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                lmin[0] = n0.x; lmin[1] = n0.y; lmin[2] = n0.z; lmax[0] = n0.w; lmax[1] = n1.x; lmax[2] = n1.y;
                rmin[0] = n1.z; rmin[1] = n1.w; rmin[2] = n2.x; rmax[0] = n2.y; rmax[1] = n2.z; rmax[2] = n2.w;
                left = __float_as_uint(n3.x); right = __float_as_uint(n3.y);
            }
            if (STATS) nInner++;
            float lnear, rnear;
            bool lh = slab(lmin, lmax, orig, dinv, tbest, &lnear);
            bool rh = slab(rmin, rmax, orig, dinv, tbest, &rnear);
            if (lh && rh) {
                uint32_t closer = left, farther = right;
                bool goRight = rnear < lnear;
                if (goRight) { closer = right; farther = left; }
                stk.push(sp++, farther);
                cur = closer;
#if TRACE_COMPACT
                for (int k = 0; k < 3; k++) { pmin[k] = goRight ? rmin[k] : lmin[k]; pmax[k] = goRight ? rmax[k] : lmax[k]; }
                boxKnown = sc.cnodes != nullptr;
#endif
                continue;
            } else if (lh || rh) {
                cur = lh ? left : right;
#if TRACE_COMPACT
                for (int k = 0; k < 3; k++) { pmin[k] = lh ? lmin[k] : rmin[k]; pmax[k] = lh ? lmax[k] : rmax[k]; }
                boxKnown = sc.cnodes != nullptr;
#endif
                continue;
            }
        } else {
            uint32_t slot = cur & ~FLX_LEAF_BIT;
            const float4 *tp = reinterpret_cast<const float4 *>(sc.trirecs + slot);
            float4 a = tp[0], b = tp[1], c = tp[2];
            int count = __float_as_int(b.w);
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
#if TRACE_COMPACT
        boxKnown = false;                      // a popped node's own box is not at hand: full record
#endif
    }
    return false;
#endif
#undef FLX_WAVE_TICK
}


} // namespace flxd
