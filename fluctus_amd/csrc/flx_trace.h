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

} // namespace flxd
