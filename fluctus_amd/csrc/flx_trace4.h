// flx_trace4.h -- traversal of the 4-wide quantised tree (flx_wide.h) for gfx950.
//
// Node test.  With sd = s * dinv and od = (o - orig) * dinv per axis, plane q of a child has
//     t(q) = fma(q, sd, od)                                   -- one v_cvt_f32_ubyteN + one v_fma_f32 per plane.
// The reference's test of the child's EXACT box is T(b) = fl(fl(b - orig) * dinv) (src/intersect.cl:41-60).  The builder
// guarantees (in real arithmetic) o + qlo s <= bmin and o + qhi s >= bmax, and with u = 2^-24, M = max |t| over the node's
// grid on that axis, |t(q) - real| <= 3uM and |T(b) - real| <= 2uM, so shifting the near planes by -e and the far planes by
// +e with e = 2^-21 (|od| + 256 |sd|) >= 8uM makes [tnear, tfar] a superset of the reference's interval on every axis: the
// wide test passes whenever the reference's passes (same three conditions: tfar >= 0, tnear <= tfar, tnear < tMax).
// dinv is clamped to +-2^100 for this test only (a zero direction component gives +-inf and 0 * inf = NaN planes that would
// switch the axis off and let an axis-parallel ray wander through the whole slab of the scene); e, which grows with |sd|,
// covers the boundary cases of that substitution.  Near / far planes are picked per axis by the sign of the direction with
// one v_cndmask on the packed bytes of all four children.
//
// Leaf test: the leaf's exact fp32 box with the reference's arithmetic (slab(), the UNclamped 1/dir), then its triangles in
// index-list order with the same Moller-Trumbore as the binary path.
#pragma once
#include "flx_trace.h"
#include "flx_wide.h"

namespace flxd {

#ifndef WIDE_BLOCK
#define WIDE_BLOCK 64
#endif
#ifndef WIDE_LDS_LEVELS
#define WIDE_LDS_LEVELS 16          // stack entries kept in LDS per lane (4 KiB per wave); deeper entries spill to global memory
#endif

// Traversal stack: a RING of WIDE_LDS_LEVELS entries per lane in LDS ([slot][lane], conflict-free), holding stack levels
// [base, base + WIDE_LDS_LEVELS); level L lives in slot L & (WIDE_LDS_LEVELS - 1).  Pushes are UNCONDITIONAL ds_writes followed by
// sp += valid (no exec-mask branch per child); when a visit could overflow the window, the oldest 8 levels are paged out to the
// global spill area, and paged back in when the window runs empty -- both rare, out of the hot path.
// (the two paging routines are free functions with by-value arguments so that the stack descriptor stays in registers)
__device__ __noinline__ void wstack_page_out(uint32_t *lds, uint32_t *spill, uint32_t stride, int base)
{
    for (int k = 0; k < 8; k++) spill[(size_t)(base + k) * stride] = lds[((base + k) & (WIDE_LDS_LEVELS - 1)) * WIDE_BLOCK];
}
__device__ __noinline__ void wstack_page_in(uint32_t *lds, const uint32_t *spill, uint32_t stride, int base)
{
    for (int k = 0; k < 8; k++) lds[((base + k) & (WIDE_LDS_LEVELS - 1)) * WIDE_BLOCK] = spill[(size_t)(base + k) * stride];
}
struct WStack {
    uint32_t *lds;          // this lane's column: lds[slot * WIDE_BLOCK]
    uint32_t *spill;        // this lane's column: spill[level * stride]
    uint32_t stride;
    int base;               // levels [0, base) live in the spill area; multiple of 8
    __device__ __forceinline__ uint32_t &slot(int level) { return lds[(level & (WIDE_LDS_LEVELS - 1)) * WIDE_BLOCK]; }
    __device__ __forceinline__ void put(int &sp, uint32_t v, bool valid) { slot(sp) = v; sp += valid ? 1 : 0; }
    // room for 4 more entries
    __device__ __forceinline__ void reserve(int sp) { if (sp - base > WIDE_LDS_LEVELS - 4) { wstack_page_out(lds, spill, stride, base); base += 8; } }
    // pop; FLX_RAY_DONE when the stack is empty
    __device__ __forceinline__ uint32_t pop(int &sp)
    {
        if (sp == base) { if (base == 0) return FLX_RAY_DONE; base -= 8; wstack_page_in(lds, spill, stride, base); }
        return slot(--sp);
    }
};

__device__ __forceinline__ float ub0(uint32_t v) { return (float)(v & 0xFFu); }
__device__ __forceinline__ float ub1(uint32_t v) { return (float)((v >> 8) & 0xFFu); }
__device__ __forceinline__ float ub2(uint32_t v) { return (float)((v >> 16) & 0xFFu); }
__device__ __forceinline__ float ub3(uint32_t v) { return (float)(v >> 24); }
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float max3_(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float min3_(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

#define FLX_WIDE_DINV_MAX 1.2676506e30f          // 2^100
#define FLX_WIDE_DINV_FAR 1.8446744e19f          // 2^64: the clamp of a ray whose origin lies beyond +-2^26

// compare-exchange of (key, ref) pairs, ascending key
#define FLX_CE(ka, ra, kb, rb) do { const bool sw_ = kb < ka; const float tk_ = sw_ ? kb : ka; const uint32_t tr_ = sw_ ? rb : ra; \
        kb = sw_ ? ka : kb; rb = sw_ ? ra : rb; ka = tk_; ra = tr_; } while (0)

// Per-ray constants of the wide traversal
struct WRay {
    f3 orig, dir, dinv;            // dinv = the reference's native_recip(dir) (exact, may be +-inf): leaf boxes
    float dwx, dwy, dwz;           // dinv clamped to +-2^100: node test
    bool negx, negy, negz;
    __device__ __forceinline__ void setup(f3 o, f3 d, float clampNear)
    {
        orig = o; dir = d;
        dinv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        // od = (o - orig) * dw must stay finite: an infinite od makes od - e a NaN, every comparison false and the child culled, and
        // the any-hit sort uses +inf as its miss sentinel.  clampNear is 2^100 when every node coordinate lies inside +-2^26 (every
        // real scene; flx_upload_scene) and the origin does too: |o - orig| < 2^27, the product stays below 2^127.  Otherwise -- node
        // coordinates are bounded by 2^62 at upload (flx_wide.h) -- the clamp is 2^64: |o - orig| < 2^63.  (The clamp only has to
        // switch an axis off for a ray that runs parallel to it INSIDE the node's slab: e >= 2^-13 s |dw| must exceed every entry
        // distance of the other axes, header comment; 2^64 still does for nodes larger than 2^-37 of the scene.)
        const float far = fmaxf_(fmaxf_(absf(o.x), absf(o.y)), absf(o.z));
        const float lim = far < 67108864.0f ? clampNear : FLX_WIDE_DINV_FAR;
        dwx = fminf_(fmaxf_(dinv.x, -lim), lim);
        dwy = fminf_(fmaxf_(dinv.y, -lim), lim);
        dwz = fminf_(fmaxf_(dinv.z, -lim), lim);
        negx = (__float_as_uint(dwx) >> 31) != 0u; negy = (__float_as_uint(dwy) >> 31) != 0u; negz = (__float_as_uint(dwz) >> 31) != 0u;
    }
};

// One inner-node visit of one ray: test the four children, push what has to wait, choose where to go next.
// (v_pk_fma_f32 was tried for the 24 plane evaluations: 12 % fewer instructions, same time -- packed f32 ops take two issue
// slots on CDNA4's 32-wide SIMDs.)
// ANY_ORDER (any-hit only): which hit child a shadow ray descends into first.  The answer is the same whatever the order; the number of
// nodes visited before the first occluder is not.  0 = the LAST hit slot, earlier hits pushed (no sort; pops come back in reverse slot
// order); 1 = FARTHEST entry first, the others pushed sorted so that pops go far -> near.  Measured (k_shadow4, 4 M paths, MI355X): rays toward
// an environment light -- unbounded, leaving through the whole scene -- kitchen 0.358 -> 0.319 ms, courtyard 0.374 -> 0.353 with 1; rays toward
// an area light (conference) 0.256 -> 0.299, so 0 there.  Also tried: nearest first 0.466 (kitchen), longest overlap [max(tn,0), min(tf,tmax)]
// first 0.41, farthest / longest first with the rest in slot order 0.41 / 0.39 -- all worse than either.
template <bool ANY_HIT, int ANY_ORDER>
__device__ __forceinline__ void wide_node_visit(const float4 *wn, WStack &stk, const WRay &r, float tbest, int &sp, uint32_t &cur)
{
    const float4 *np = wn + (size_t)cur * 4;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    // per-axis slope / offset of t(q) = q * sd + od, and the conservative shift e (header comment)
    const float sdx = n0.w * r.dwx, sdy = n1.x * r.dwy, sdz = n1.y * r.dwz;
    const float odx = (n0.x - r.orig.x) * r.dwx, ody = (n0.y - r.orig.y) * r.dwy, odz = (n0.z - r.orig.z) * r.dwz;
    const float ex = 4.76837158e-7f * fma_(absf(sdx), 256.0f, absf(odx));      // 2^-21 (|od| + 256 |sd|)
    const float ey = 4.76837158e-7f * fma_(absf(sdy), 256.0f, absf(ody));
    const float ez = 4.76837158e-7f * fma_(absf(sdz), 256.0f, absf(odz));
    const float onx = odx - ex, ofx = odx + ex, ony = ody - ey, ofy = ody + ey, onz = odz - ez, ofz = odz + ez;
    const uint32_t qlox = __float_as_uint(n2.z), qloy = __float_as_uint(n2.w), qloz = __float_as_uint(n3.x);
    const uint32_t qhix = __float_as_uint(n3.y), qhiy = __float_as_uint(n3.z), qhiz = __float_as_uint(n3.w);
    const uint32_t qnx = r.negx ? qhix : qlox, qfx = r.negx ? qlox : qhix;
    const uint32_t qny = r.negy ? qhiy : qloy, qfy = r.negy ? qloy : qhiy;
    const uint32_t qnz = r.negz ? qhiz : qloz, qfz = r.negz ? qloz : qhiz;
    uint32_t r0 = __float_as_uint(n1.z), r1 = __float_as_uint(n1.w), r2 = __float_as_uint(n2.x), r3 = __float_as_uint(n2.y);
    // tnear <= tfar, tfar >= 0, tnear < tMax: the reference's three conditions (src/intersect.cl:55-59).  Unused child slots point
    // at a dummy leaf that cannot be hit (flx_wide.h), so no validity test is needed here.
#define FLX_CHILD(UB, KEY, HIT) \
    { const float tn = max3_(fma_(UB(qnx), sdx, onx), fma_(UB(qny), sdy, ony), fma_(UB(qnz), sdz, onz)); \
      const float tf = min3_(fma_(UB(qfx), sdx, ofx), fma_(UB(qfy), sdy, ofy), fma_(UB(qfz), sdz, ofz)); \
      HIT = (tn <= tf) && (tf >= 0.0f) && (tn < tbest); KEY = tn; }
    float k0, k1, k2, k3; bool h0, h1, h2, h3;
    FLX_CHILD(ub0, k0, h0)
    FLX_CHILD(ub1, k1, h1)
    FLX_CHILD(ub2, k2, h2)
    FLX_CHILD(ub3, k3, h3)
#undef FLX_CHILD
    stk.reserve(sp);
    if (ANY_HIT && ANY_ORDER == 1) {
        const float INF = __builtin_huge_valf();
        float w0 = h0 ? -k0 : INF, w1 = h1 ? -k1 : INF, w2 = h2 ? -k2 : INF, w3 = h3 ? -k3 : INF;
        FLX_CE(w0, r0, w1, r1); FLX_CE(w2, r2, w3, r3); FLX_CE(w0, r0, w2, r2); FLX_CE(w1, r1, w3, r3); FLX_CE(w1, r1, w2, r2);
        stk.put(sp, r3, w3 < INF); stk.put(sp, r2, w2 < INF); stk.put(sp, r1, w1 < INF);
        if (w0 < INF) cur = r0;
        else cur = stk.pop(sp);
    } else if (ANY_HIT) {
        // order-free: continue with the LAST hit child, push the earlier ones
        const bool p0 = h0 && (h1 || h2 || h3), p1 = h1 && (h2 || h3), p2 = h2 && h3;
        stk.put(sp, r0, p0); stk.put(sp, r1, p1); stk.put(sp, r2, p2);
        if (h0 || h1 || h2 || h3) cur = h3 ? r3 : (h2 ? r2 : (h1 ? r1 : r0));
        else cur = stk.pop(sp);
    } else {
        // nearest first: sort the four (entry distance, ref) pairs, misses at +inf
        const float INF = __builtin_huge_valf();
        k0 = h0 ? k0 : INF; k1 = h1 ? k1 : INF; k2 = h2 ? k2 : INF; k3 = h3 ? k3 : INF;
        FLX_CE(k0, r0, k1, r1); FLX_CE(k2, r2, k3, r3); FLX_CE(k0, r0, k2, r2); FLX_CE(k1, r1, k3, r3); FLX_CE(k1, r1, k2, r2);
        stk.put(sp, r3, k3 < INF); stk.put(sp, r2, k2 < INF); stk.put(sp, r1, k1 < INF);
        if (k0 < INF) cur = r0;
        else cur = stk.pop(sp);
    }
}

// One leaf visit: the leaf node's exact box with the reference's test, then its triangles in index-list order.
// Returns true as soon as ANY_HIT finds an occluder.
template <bool ANY_HIT, bool STATS>
__device__ __forceinline__ bool wide_leaf_visit(const float4 *wleaf, const WRay &r, uint32_t cur, float &tbest, float &ubest, float &vbest, int &tribest,
                                                uint32_t &nTri, unsigned long long *wstats)
{
    const float4 *lp = wleaf + (cur & FLX_WIDE_OFF_MASK);
    const float4 b0 = lp[0], b1 = lp[1];
    const float bmin[3] = {b0.x, b0.y, b0.z}, bmax[3] = {b1.x, b1.y, b1.z};
    float tnear;
    if (slab(bmin, bmax, r.orig, r.dinv, tbest, &tnear)) {         // the reference's own test of the leaf node's box
        const int count = __float_as_int(b0.w);
        const float4 *tp = lp + 2;
        float4 a = tp[0], b = tp[1], c = tp[2];
        for (int k = 0;;) {
            if (STATS && wstats) { const uint64_t m_ = __ballot(true); if (lane_id() == (uint32_t)__ffsll((long long)m_) - 1u) atomicAdd(&wstats[3], 1ull); }
            if (STATS) nTri++;
            float t, u, v;
            if (moller_trumbore(r.orig, r.dir, ld3(a), ld3(b), ld3(c), &t, &u, &v) && t > 0.0f && t < tbest) {
                if (ANY_HIT) return true;
                tbest = t; ubest = u; vbest = v; tribest = __float_as_int(a.w);
            }
            if (++k >= count) break;
            tp += 3;
            a = tp[0]; b = tp[1]; c = tp[2];
        }
    }
    return false;
}

template <bool ANY_HIT, bool STATS, int ANY_ORDER = 0>
__device__ __forceinline__ bool traverse4(const Scene &sc, WStack &stk, f3 orig, f3 dir, float &tbest, float &ubest, float &vbest,
                                          int &tribest, uint32_t &nInner, uint32_t &nTri, uint32_t &nLeaf, unsigned long long *wstats = nullptr)
{
#define FLX_WAVE_TICK(k) do { if (STATS && wstats) { const uint64_t m_ = __ballot(true); \
        if (lane_id() == (uint32_t)__ffsll((long long)m_) - 1u) atomicAdd(&wstats[k], 1ull); } } while (0)
    WRay r; r.setup(orig, dir, sc.wideClamp);
    const float4 *wn = reinterpret_cast<const float4 *>(sc.wnodes);
    int sp = 0;
    uint32_t cur = sc.wrootRef;
    for (;;) {
        FLX_WAVE_TICK(0);
        // (ending the descent round early -- when N lanes stand on a leaf -- is what halves the persistent kernel's instruction count
        //  (trace4r.hip); WITHOUT lane refill it loses: k_shadow4 0.315 -> 0.35-0.38 ms for N = 32 ... 8, profiles/r03_shadow_wait_ab.txt)
        while (!(cur & FLX_WIDE_LEAF_BIT)) {
            FLX_WAVE_TICK(1);
            if (STATS) nInner++;
            wide_node_visit<ANY_HIT, ANY_ORDER>(wn, stk, r, tbest, sp, cur);
        }
        if (cur == FLX_RAY_DONE) break;
        FLX_WAVE_TICK(2);
        if (STATS) nLeaf++;
        if (wide_leaf_visit<ANY_HIT, STATS>(sc.wleaf, r, cur, tbest, ubest, vbest, tribest, nTri, wstats)) return true;
        cur = stk.pop(sp);
        if (cur == FLX_RAY_DONE) break;
    }
    return false;
#undef FLX_WAVE_TICK
}

// ---- any-hit traversal with a node-visit BUDGET, resumable (trace4.hip: k_shadow4s, "tail splitting").
// A thread-per-ray wave lives as long as its longest ray: on the kitchen's shadow rays (6.8 wide-node visits in the mean, p95 12, max 58) a wave of
// k_shadow4 runs 18 rounds with 20 of its 64 lanes busy.  Here a ray that has used up its budget in front of an INNER node SUSPENDS: it leaves the loop
// with the node it stood on pushed on top of its stack; the kernel writes {queue index, sp, stack} to a continuation record and a later launch -- its rays
// compacted, a new budget -- goes on from there.  The ray's own sequence of node and leaf visits is exactly k_shadow4's (same order, same arithmetic, the
// any-hit t bound never shrinks), only cut across launches, so shadowRayBlocked is bit-identical by construction.
// Returns 0 not occluded | 1 occluded | 2 suspended (sp entries on the stack, the next node on top).  maxKeep = entries a continuation record holds;
// a ray whose stack is deeper (or partly paged out) simply keeps going.
#define FLX_SPLIT_KEEP 14
template <int ANY_ORDER>
__device__ __forceinline__ int traverse4_any_budget(const Scene &sc, WStack &stk, const WRay &r, float tmax, int &sp, uint32_t cur, int budget)
{
    const float4 *wn = reinterpret_cast<const float4 *>(sc.wnodes);
    float u, v; int tri; uint32_t nTri = 0;
    for (;;) {
        while (!(cur & FLX_WIDE_LEAF_BIT) && budget > 0) { wide_node_visit<true, ANY_ORDER>(wn, stk, r, tmax, sp, cur); budget--; }
        if (!(cur & FLX_WIDE_LEAF_BIT)) {                         // out of budget in front of an inner node
            if (stk.base == 0 && sp < FLX_SPLIT_KEEP) { stk.slot(sp) = cur; sp++; return 2; }
            budget = 0x7fffffff;                                  // too deep for a record: finish here
            continue;
        }
        if (cur == FLX_RAY_DONE) return 0;
        if (wide_leaf_visit<true, false>(sc.wleaf, r, cur, tmax, u, v, tri, nTri, nullptr)) return 1;
        cur = stk.pop(sp);
        if (cur == FLX_RAY_DONE) return 0;
    }
}

} // namespace flxd
