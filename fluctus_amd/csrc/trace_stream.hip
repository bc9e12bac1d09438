// trace_stream.hip -- "streaming" BVH traversal for gfx950: every wave owns a contiguous CHUNK of the ray queue and
// keeps its 64 lanes busy by handing a finished lane the next ray of the chunk (trace_mode 2).
//
// Why: in the thread-per-ray kernels (trace.hip) a wave lasts as long as its longest ray and its lanes wait for each
// other at every leaf; measured SIMD efficiency of the descent loop is 22-27 % (bench.py roofline.simd_efficiency) and the
// kernels are bound by trip latency, not by bytes.  What this kernel changes:
//   * grid = what the machine holds at once (CUs x resident waves), chunk = ceil(queue length / grid) rays (>= 64);
//     the chunk is STATIC, so a refill costs no atomic (a single hot counter sustains only ~88 atomics/us on MI355X)
//     -- the wave's cursor is one SGPR;
//   * a refill happens at the top of the outer iteration when at least `refillMin` lanes are idle: idle lanes take
//     consecutive queue entries (ballot + mbcnt ranks), load their ray and join the next descent at the root;
//   * results leave the loop through a tiny store (closest hit: one float4 {t,u,v,tri} per ray, indexed by queue slot;
//     any hit: shadowRayBlocked[gid]) so the loop's live state stays at 16 VGPRs + node/triangle temporaries;
//     the closest-hit commit (shading-record fetch, area-light quad, hit-record writes) runs after the loop, coalesced
//     over the chunk, in the same kernel.
// Per-ray arithmetic and visit order are those of trace.hip (flx_trace.h: slab, moller_trumbore, near child first,
// leaf triangles in index order), so results are bit-identical to it and to the oracle.
//
// Replaces reference kernels traceExtension (src/wf_extrays.cl:5-36 -> bvh_intersect, src/bvh.cl:234-310) and
// traceShadow (src/wf_shadowrays.cl:6-38 -> bvh_occluded, src/bvh.cl:312-373).
#include "flx_trace.h"

namespace flxd {

#define STREAM_NONE 0xFFFFFFFFu
#ifndef STREAM_MIN_WAVES
#define STREAM_MIN_WAVES 1
#endif

// UNIFIED = false (trace_mode 2): while-while inside the wave -- descend until every live lane stands on a leaf, then
//   intersect all pending leaves; refill between phases.  Lanes still wait for the wave's deepest descent.
// UNIFIED = true (trace_mode 3): one WORK ITEM per lane per trip -- an inner-node test or ONE triangle test -- behind one
//   unified 64-byte fetch (node record or triangle record), so no lane ever waits for another lane's phase and the
//   loads of both kinds are in flight together; the end of a leaf run is a flag in the triangle record (TriRec.c.w).
template <bool ANY_HIT, bool STATS, bool UNIFIED>
__global__ __launch_bounds__(TRACE_BLOCK, STREAM_MIN_WAVES) void k_trace_stream(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux,
                                                                               float4 *hitraw, int refillMin, int innerMin)
{
    __shared__ uint32_t s_stack[LDS_LEVELS * TRACE_BLOCK];
    const int QID = ANY_HIT ? FLX_Q_SHADOW : FLX_Q_EXTENSION;
    const uint32_t qlen = ANY_HIT ? qs.counters[QID] : ext_len(qs);
    const uint32_t *queue = qs.q[QID];
    uint32_t chunk = (qlen + gridDim.x - 1u) / gridDim.x;
    if (chunk < 64u) chunk = 64u;
    const uint32_t base = blockIdx.x * chunk;
    if (base >= qlen) return;
    const uint32_t end = min(qlen, base + chunk);
    uint32_t next = base;                                   // wave-uniform cursor into the chunk

    Stack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.spill = aux.spill + (blockIdx.x * TRACE_BLOCK + threadIdx.x);
    unsigned long long *wstats = STATS ? aux.stats + (ANY_HIT ? 12 : 8) : nullptr;
#define FLX_WAVE_TICK(k) do { if (STATS) { const uint64_t m_ = __ballot(true); \
        if (lane_id() == (uint32_t)__ffsll((long long)m_) - 1u) atomicAdd(&wstats[k], 1ull); } } while (0)

    // per-lane ray
    uint32_t slot = STREAM_NONE;                            // closest hit: queue index of the lane's ray; any hit: its gid
    uint32_t cur = FLX_RAY_DONE;
    int sp = 0;
    f3 orig = mk3(0.0f), dir = mk3(0.0f), dinv = mk3(0.0f);
    float tbest = 0.0f, ubest = 0.0f, vbest = 0.0f;
    int tribest = -1;
    bool occluded = false;
    uint32_t nInner = 0, nTri = 0, nRays = 0;

    if constexpr (UNIFIED) {
        for (;;) {
            FLX_WAVE_TICK(0);
            const uint64_t im = __ballot(cur == FLX_RAY_DONE);
            const uint32_t nIdle = (uint32_t)__popcll(im);
            if ((int)nIdle >= refillMin) {                      // wave-uniform, rare
                if (cur == FLX_RAY_DONE && slot != STREAM_NONE) {
                    if (ANY_HIT) st.blocked[slot] = occluded ? 1u : 0u;
                    else hitraw[slot] = make_float4(tbest, ubest, vbest, __int_as_float(tribest));
                    slot = STREAM_NONE;
                }
                if (next < end) {
                    const uint32_t my = next + mbcnt(im);
                    if (cur == FLX_RAY_DONE && my < end) {
                        const uint32_t gid = queue[my];
                        float4 o4, d4;
                        if (!ANY_HIT) { o4 = rd4(st.at(S_ORIG, gid)); d4 = rd4(st.at(S_DIR, gid)); }
                        else { o4 = rd4(st.at(S_SHO, gid)); d4 = rd4(st.at(S_SHD, gid)); }
                        orig = ld3(o4); dir = ld3(d4);
                        dinv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                        tbest = ANY_HIT ? o4.w : FLX_FLT_MAX;
                        ubest = 0.0f; vbest = 0.0f; tribest = -1; occluded = false;
                        sp = 0; cur = sc.rootRef;
                        slot = ANY_HIT ? gid : my;
                        if (STATS) nRays++;
                        if (ANY_HIT && p.useAreaLight) {         // the light quad itself blocks first (src/wf_shadowrays.cl:32-33)
                            float tl = tbest;
                            if (light_quad(p.areaLight, orig, dir, &tl)) { occluded = true; cur = FLX_RAY_DONE; }
                        }
                    }
                    next = min(end, next + nIdle);
                } else if (nIdle == 64u) break;
            }
            if (cur != FLX_RAY_DONE) {
                const bool inner = !(cur & FLX_LEAF_BIT);
                const char *rec = inner ? reinterpret_cast<const char *>(sc.bnodes) + (size_t)cur * sizeof(BNode)
                                        : reinterpret_cast<const char *>(sc.trirecs) + (size_t)(cur & ~FLX_LEAF_BIT) * sizeof(TriRec);
                const float4 *np = reinterpret_cast<const float4 *>(rec);
                const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];     // a triangle record is 48 B: n3 over-reads (buffer padded)
                bool popNext = false;
                if (inner) {
                    FLX_WAVE_TICK(1);
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
                    } else popNext = true;
                } else {
                    FLX_WAVE_TICK(3);
                    if (STATS) nTri++;
                    float t, u, v;
                    if (moller_trumbore(orig, dir, ld3(n0), ld3(n1), ld3(n2), &t, &u, &v) && t > 0.0f && t < tbest) {
                        if (ANY_HIT) occluded = true;
                        else { tbest = t; ubest = u; vbest = v; tribest = __float_as_int(n0.w); }
                    }
                    if (ANY_HIT && occluded) { cur = FLX_RAY_DONE; }
                    else if (__float_as_uint(n2.w) != 0u) popNext = true;       // last triangle of the leaf run
                    else cur = cur + 1u;
                }
                if (popNext) {
                    cur = (sp == 0) ? FLX_RAY_DONE : stk.pop(sp - 1);
                    sp = (sp == 0) ? 0 : sp - 1;
                }
            }
        }
    } else {
        for (;;) {
            FLX_WAVE_TICK(0);
            // ---- retire finished rays
            const bool idle = cur == FLX_RAY_DONE;
            if (idle && slot != STREAM_NONE) {
                if (ANY_HIT) st.blocked[slot] = occluded ? 1u : 0u;
                else hitraw[slot] = make_float4(tbest, ubest, vbest, __int_as_float(tribest));
                slot = STREAM_NONE;
            }
            // ---- refill idle lanes from the chunk (wave-uniform decision)
            const uint64_t im = __ballot(idle);
            const uint32_t nIdle = (uint32_t)__popcll(im);
            if (next < end && (int)nIdle >= refillMin) {
                const uint32_t my = next + mbcnt(im);
                if (idle && my < end) {
                    const uint32_t gid = queue[my];
                    float4 o4, d4;
                    if (!ANY_HIT) { o4 = rd4(st.at(S_ORIG, gid)); d4 = rd4(st.at(S_DIR, gid)); }
                    else { o4 = rd4(st.at(S_SHO, gid)); d4 = rd4(st.at(S_SHD, gid)); }
                    orig = ld3(o4); dir = ld3(d4);
                    dinv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                    tbest = ANY_HIT ? o4.w : FLX_FLT_MAX;
                    ubest = 0.0f; vbest = 0.0f; tribest = -1; occluded = false;
                    sp = 0; cur = sc.rootRef;
                    slot = ANY_HIT ? gid : my;
                    if (STATS) nRays++;
                    if (ANY_HIT && p.useAreaLight) {             // the light quad itself blocks first (src/wf_shadowrays.cl:32-33)
                        float tl = tbest;
                        if (light_quad(p.areaLight, orig, dir, &tl)) { occluded = true; cur = FLX_RAY_DONE; }
                    }
                }
                next = min(end, next + nIdle);
            } else if (nIdle == 64u) {
                break;                                          // every lane idle and the chunk is exhausted
            }
            // ---- descend (and pop).  The wave leaves this loop when no lane descends any more, or when fewer than
            // `innerMin` lanes still do while others wait on leaves: the stragglers keep their node and rejoin later, so
            // neither the descent nor the leaf phase runs with a nearly empty wave.
            for (;;) {
                const bool in = !(cur & FLX_LEAF_BIT);
                const uint64_t mi = __ballot(in);
                if (mi == 0ull) break;
                if ((int)__popcll(mi) < innerMin && __ballot(cur != FLX_RAY_DONE && !in) != 0ull) break;
                if (in) {
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
            }
            // ---- intersect the pending leaves together
            if ((cur & FLX_LEAF_BIT) && cur != FLX_RAY_DONE) {
                const float4 *tp = reinterpret_cast<const float4 *>(sc.trirecs + (cur & ~FLX_LEAF_BIT));
                float4 a = tp[0], b = tp[1], c = tp[2];
                const int count = __float_as_int(b.w);
                FLX_WAVE_TICK(2);
                for (int k = 0;;) {
                    FLX_WAVE_TICK(3);
                    if (STATS) nTri++;
                    float t, u, v;
                    if (moller_trumbore(orig, dir, ld3(a), ld3(b), ld3(c), &t, &u, &v) && t > 0.0f && t < tbest) {
                        if (ANY_HIT) { occluded = true; break; }
                        tbest = t; ubest = u; vbest = v; tribest = __float_as_int(a.w);
                    }
                    if (++k >= count) break;
                    tp += 3;
                    a = tp[0]; b = tp[1]; c = tp[2];
                }
                if ((ANY_HIT && occluded) || sp == 0) cur = FLX_RAY_DONE;
                else cur = stk.pop(--sp);
            }
        }
    }
#undef FLX_WAVE_TICK

    uint32_t nHit = 0;
    if (!ANY_HIT) {
        // ---- commit: shading attributes of the winning triangle (src/bvh.cl:271-279), implicit area-light hit
        // (src/wf_extrays.cl:28-29, src/intersect.cl:124-155); hitraw was written by other lanes of this same wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t i = base + threadIdx.x; i < end; i += TRACE_BLOCK) {
            const uint32_t gid = queue[i];
            const float4 raw = hitraw[i];
            const float4 o4 = rd4(st.at(S_ORIG, gid));
            const float4 d4 = rd4(st.at(S_DIR, gid));
            const f3 ro = ld3(o4), rdir = ld3(d4);
            float t = raw.x;
            const float u = raw.y, v = raw.z;
            int tri = __float_as_int(raw.w);
            f3 P = mk3(0.0f), N = mk3(0.0f);
            float tu = 0.0f, tv = 0.0f;
            int matId = -1;
            uint32_t flags = 0;
            if (tri >= 0) {
                const float4 *sp4 = reinterpret_cast<const float4 *>(sc.shade + tri);
                const float4 a = sp4[0], b = sp4[1], c = sp4[2], d = sp4[3];
                P = ro + t * rdir;
                N = normalize(bary(u, v, ld3(a), ld3(b), ld3(c)));
                const f3 uv = bary(u, v, mk3(a.w, b.w, 0.0f), mk3(c.w, d.x, 0.0f), mk3(d.y, d.z, 0.0f));
                tu = uv.x; tv = uv.y;
                matId = __float_as_int(d.w);
                if (STATS) nHit++;
            }
            if (p.sampleImpl && p.useAreaLight) {
                if (light_quad(p.areaLight, ro, rdir, &t)) {
                    if (STATS && tri >= 0) nHit--;
                    flags = 1u; P = ro + t * rdir; N = V(p.areaLight.N); tri = 0; matId = 0;
                }
            }
            wr4(st.at(S_DIR, gid), mk4u(rdir, __float_as_uint(d4.w) + 1u));                 // pathLen += 1
            wr4(st.at(S_HITP, gid), mk4(P, t));
            const uint32_t keep = __float_as_uint(reinterpret_cast<const float *>(st.at(S_HITN, gid))[3]) & 2u;   // backfaceHit belongs to `logic`
            wr4(st.at(S_HITN, gid), mk4u(N, flags | keep));
            wr4(st.at(S_HITUV, gid), make_float4(tu, tv, __int_as_float(tri), __int_as_float(matId)));
        }
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

template <bool ANY_HIT, bool UNIFIED>
static void launch_s(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill,
                     unsigned long long *stats, float4 *hitraw, int refillMin, int innerMin, uint32_t gridWaves)
{
    uint32_t blocks = gridWaves;
    const uint32_t need = (st.numTasks + TRACE_BLOCK - 1) / TRACE_BLOCK;
    if (blocks > need) blocks = need;
    if (blocks == 0) blocks = 1;
    TraceAux aux{spill, blocks * TRACE_BLOCK, stats};
    if (stats) hipLaunchKernelGGL((k_trace_stream<ANY_HIT, true, UNIFIED>), dim3(blocks), dim3(TRACE_BLOCK), 0, s, st, qs, sc, p, aux, hitraw, refillMin, innerMin);
    else hipLaunchKernelGGL((k_trace_stream<ANY_HIT, false, UNIFIED>), dim3(blocks), dim3(TRACE_BLOCK), 0, s, st, qs, sc, p, aux, hitraw, refillMin, innerMin);
}

void launch_extend_stream(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill,
                          unsigned long long *stats, float4 *hitraw, int refillMin, int innerMin, uint32_t gridWaves, bool unified)
{
    if (unified) launch_s<false, true>(s, st, qs, sc, p, spill, stats, hitraw, refillMin, innerMin, gridWaves);
    else launch_s<false, false>(s, st, qs, sc, p, spill, stats, hitraw, refillMin, innerMin, gridWaves);
}

void launch_shadow_stream(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill,
                          unsigned long long *stats, int refillMin, int innerMin, uint32_t gridWaves, bool unified)
{
    if (unified) launch_s<true, true>(s, st, qs, sc, p, spill, stats, nullptr, refillMin, innerMin, gridWaves);
    else launch_s<true, false>(s, st, qs, sc, p, spill, stats, nullptr, refillMin, innerMin, gridWaves);
}

} // namespace flxd
