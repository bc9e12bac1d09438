// trace4r.hip -- closest-hit and any-hit traversal of the 4-wide tree with PERSISTENT WAVES AND LANE REFILL (round 3).
//
// Why.  k_extend4 / k_shadow4 (trace4.hip) are VALU-issue-bound: 5.3e8 VALU wave-instructions per 4 M-ray launch at the ~4 SIMD-cycles
// every plain VALU instruction costs on gfx950 (scripts/ubench/valu_rate.hip) IS the kernel time (0.85 ms), and only 26 % of the lanes of
// those instructions do work -- a thread-per-ray wave lives as long as its longest ray (~25 node visits against a mean of 10.5) and,
// inside it, the descent loop runs as long as its longest descent.  Here a wave keeps its lanes busy: the grid is the number of waves
// the machine holds at once, wave w works through the 64-ray blocks w, w + G, w + 2G, ... of the queue (no atomics: a hot counter
// sustains only ~88 atomics/us), and whenever at least `refillMin` lanes have finished their ray, those lanes write their result and
// take the next rays of the wave's block.
//
// Round 2 tried this (scripts/experiments/trace4p.hip.txt) and lost: 100 VGPRs -> 4 waves per SIMD, and the full commit of
// traceExtension (shading-record gather, normalisation, light quad: ~150 instructions + 4 loads) ran at every refill for a handful of
// lanes.  Differences here: (1) the closest-hit kernel does NOT commit -- a finished lane stores its raw result {u, v, triangle, t} in
// the path's HITUV record (one 16-byte store) and k_commit4, a streaming kernel over the same queue, turns it into the hit record
// afterwards (it runs beside the shadow traversal, which is VALU-bound while this one is memory-bound); (2) nothing of the commit is
// live in the loop, so the kernel keeps the register budget of k_extend4.
//
// Per-ray arithmetic is exactly that of k_extend4 / k_shadow4 (same wide_node_visit / wide_leaf_visit, the ray's own stack column), so
// the results are bit-identical to those kernels whatever lane or wave a ray lands on (tests/test_gpu_wide.py).
// Replaces reference kernels traceExtension (src/wf_extrays.cl:5-36) and traceShadow (src/wf_shadowrays.cl:6-38).
#include "flx_trace4.h"

namespace flxd {

#ifndef WIDE_R_MIN_WAVES
#define WIDE_R_MIN_WAVES 1
#endif
#define R_NONE 0xFFFFFFFFu

__device__ __noinline__ bool light_quad_call(const flx_arealight &L, f3 orig, f3 dir, float *t) { return light_quad(L, orig, dir, t); }

template <bool ANY_HIT, bool LIGHT, int ANY_ORDER, bool INLINE_COMMIT = false>
__global__ __launch_bounds__(WIDE_BLOCK, WIDE_R_MIN_WAVES) void k_trace4r(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux, int refillMin, int waitMax)
{
    __shared__ uint32_t s_stack[WIDE_LDS_LEVELS * WIDE_BLOCK];
    const int QID = ANY_HIT ? FLX_Q_SHADOW : FLX_Q_EXTENSION;
    const uint32_t qlen = ANY_HIT ? qs.counters[QID] : ext_len(qs);
    const uint32_t *queue = qs.q[QID];
    const uint32_t nblk = (qlen + 63u) >> 6;
    uint32_t blk = blockIdx.x, pos = 0;              // wave-uniform cursor: current block, rays of it already handed out
    if (blk >= nblk) return;

    WStack stk;
    stk.lds = s_stack + threadIdx.x;
    stk.stride = aux.totalThreads;
    stk.spill = aux.spill + (blockIdx.x * WIDE_BLOCK + threadIdx.x);
    stk.base = 0;
    const float4 *wn = reinterpret_cast<const float4 *>(sc.wnodes);

    WRay r; r.setup(mk3(0.0f), mk3(1.0f), sc.wideClamp);
    uint32_t gid = R_NONE, cur = FLX_RAY_DONE;
    int sp = 0;
    float tbest = 0.0f, ubest = 0.0f, vbest = 0.0f;
    int tribest = -1;
    bool occluded = false;
    uint32_t nI = 0, nT = 0;                         // (visit counters of the shared helpers; unused without STATS)

    for (;;) {
        const bool idle = cur == FLX_RAY_DONE;
        const uint64_t idleMask = __ballot(idle);
        const uint32_t nIdle = (uint32_t)__popcll(idleMask);
        if ((int)nIdle >= refillMin || nIdle == 64u) {                   // wave-uniform
            if (idle && gid != R_NONE) {
                if (ANY_HIT) {
                    // the light quad blocks too (src/wf_shadowrays.cl:32-33 tests it first; the result is the OR either way).  Tested here,
                    // for the rays the tree did not block, so that nothing of it is live while the tree is walked.
                    if (LIGHT && !occluded) { float tl = tbest; occluded = light_quad_call(p.areaLight, r.orig, r.dir, &tl); }
                    st.blocked[gid] = occluded ? 1u : 0u;
                }
                else if (INLINE_COMMIT) {
                    // the full commit of traceExtension, here and now, for the lanes that finished (pathLen is re-read: nothing of the
                    // commit stays live while the tree is walked)
                    uint32_t flags; int matId;
                    const float plen = reinterpret_cast<const float *>(st.at(S_DIR, gid))[3];
                    commit_hit<LIGHT>(st, sc, p, gid, r.orig, r.dir, plen, tbest, ubest, vbest, tribest, flags, matId);
                }
                else wr4(st.at(S_HITUV, gid), make_float4(ubest, vbest, __int_as_float(tribest), tbest));      // raw result -> k_commit4
                gid = R_NONE;
            }
            if (blk < nblk) {
                const uint32_t blkLen = min(64u, qlen - (blk << 6));
                const uint32_t avail = blkLen - pos;
                const uint32_t rank = mbcnt(idleMask);
                if (idle && rank < avail) {
                    gid = queue[(blk << 6) + pos + rank];
                    const float4 o4 = rd4(st.at(ANY_HIT ? S_SHO : S_ORIG, gid));
                    const float4 d4 = rd4(st.at(ANY_HIT ? S_SHD : S_DIR, gid));
                    r.setup(ld3(o4), ld3(d4), sc.wideClamp);
                    tbest = ANY_HIT ? o4.w : FLX_FLT_MAX;                // shadow: shadowRayLen
                    ubest = 0.0f; vbest = 0.0f; tribest = -1; occluded = false;
                    sp = 0; stk.base = 0; cur = sc.wrootRef;
                }
                pos += min(nIdle, avail);
                if (pos >= blkLen) { blk += gridDim.x; pos = 0; }
            } else if (nIdle == 64u) break;
        }
        // descent round: lanes on an inner node visit it; the round ends when none is left -- or when `waitMax` lanes stand on a leaf (they
        // should not wait for the longest descent of the wave; waitMax 64 = the plain while-while loop), or when enough lanes have finished
        // for a refill that can still be served.  Leaves are consumed by the leaf phase and idle lanes by the refill, so every early
        // exit makes progress.
        for (;;) {
            const bool inner = !(cur & FLX_WIDE_LEAF_BIT);
            const uint64_t mI = __ballot(inner);
            if (mI == 0ull) break;
            const int nDone = (int)__popcll(__ballot(cur == FLX_RAY_DONE));
            if (64 - (int)__popcll(mI) - nDone >= waitMax) break;
            if (blk < nblk && nDone >= refillMin && waitMax < 64) break;
            if (inner) wide_node_visit<ANY_HIT, ANY_ORDER>(wn, stk, r, tbest, sp, cur);
        }
        if (cur != FLX_RAY_DONE && (cur & FLX_WIDE_LEAF_BIT)) {
            if (wide_leaf_visit<ANY_HIT, false>(sc.wleaf, r, cur, tbest, ubest, vbest, tribest, nT, nullptr)) { occluded = true; cur = FLX_RAY_DONE; }
            else cur = stk.pop(sp);
        }
    }
    (void)nI;
}

// The commit of traceExtension for every ray of the extension queue, from the raw result k_trace4r left in HITUV (flx_trace.h: commit_hit).
__global__ __launch_bounds__(256) void k_commit4(State st, Queues qs, Scene sc, flx_render_params p)
{
    const uint32_t qlen = ext_len(qs);
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < qlen; idx += gridDim.x * blockDim.x) {
        const uint32_t gid = qs.q[FLX_Q_EXTENSION][idx];
        const float4 raw = rd4t(st.at(S_HITUV, gid));
        const float4 o4 = rd4(st.at(S_ORIG, gid));
        const float4 d4 = rd4(st.at(S_DIR, gid));
        uint32_t flags; int matId;
        commit_hit(st, sc, p, gid, ld3(o4), ld3(d4), d4.w, raw.w, raw.x, raw.y, __float_as_int(raw.z), flags, matId);
    }
}

// grid = resident waves of the device for this kernel (occupancy x CUs), capped by the number of 64-ray blocks
template <class K>
static uint32_t persistent_grid(K kernel, int &cached, uint32_t numCUs, uint32_t numTasks)
{
    if (cached == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(kernel), WIDE_BLOCK, 0) != hipSuccess || n <= 0) n = 16;
        cached = n;
    }
    const uint32_t g = numCUs * (uint32_t)cached;
    const uint32_t blocks = (numTasks + 63u) / 64u;
    return g < blocks ? g : blocks;
}

void launch_extend4r(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill, uint32_t numCUs, int refill)
{
    const int refillMin = refill & 0xFF, waitMax = ((refill >> 8) & 0xFF) ? ((refill >> 8) & 0xFF) : 64;          // option value: refillMin | waitMax << 8
    static int occ[2] = {0, 0};
    TraceAux aux{spill, ((st.numTasks + 255u) / 256u) * 256u, nullptr};
    // The commit: inline at the refill where it is cheap in registers -- without the implicit area-light quad (two more triangle tests with
    // their corner set-up) it fits the 72 VGPRs of 7 waves per SIMD; with it the kernel needs 102 -> 4 waves, so scenes with an area light
    // take the separate pass.  Bit 16 of the option forces the separate pass (A/B).
    const bool lightQuad = p.sampleImpl && p.useAreaLight;
    if (!lightQuad && !(refill & 0x10000)) {
        const uint32_t grid = persistent_grid(k_trace4r<false, false, 0, true>, occ[1], numCUs, st.numTasks);
        hipLaunchKernelGGL((k_trace4r<false, false, 0, true>), dim3(grid), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux, refillMin, waitMax);
        return;
    }
    const uint32_t grid = persistent_grid(k_trace4r<false, false, 0>, occ[0], numCUs, st.numTasks);
    hipLaunchKernelGGL((k_trace4r<false, false, 0>), dim3(grid), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux, refillMin, waitMax);
    hipLaunchKernelGGL(k_commit4, dim3(numCUs * 8), dim3(256), 0, s, st, qs, sc, p);
}

void launch_shadow4r(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill, uint32_t numCUs, int refill)
{
    const int refillMin = refill & 0xFF, waitMax = ((refill >> 8) & 0xFF) ? ((refill >> 8) & 0xFF) : 64;          // option value: refillMin | waitMax << 8
    static int occ[3] = {0, 0, 0};
    TraceAux aux{spill, ((st.numTasks + 255u) / 256u) * 256u, nullptr};
    const bool farFirst = p.useEnvMap && !p.useAreaLight;            // visit order of the any-hit traversal (trace4.hip: launch_shadow4)
    if (p.useAreaLight) {
        const uint32_t grid = persistent_grid(k_trace4r<true, true, 0>, occ[0], numCUs, st.numTasks);
        hipLaunchKernelGGL((k_trace4r<true, true, 0>), dim3(grid), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux, refillMin, waitMax);
    } else if (farFirst) {
        const uint32_t grid = persistent_grid(k_trace4r<true, false, 1>, occ[1], numCUs, st.numTasks);
        hipLaunchKernelGGL((k_trace4r<true, false, 1>), dim3(grid), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux, refillMin, waitMax);
    } else {
        const uint32_t grid = persistent_grid(k_trace4r<true, false, 0>, occ[2], numCUs, st.numTasks);
        hipLaunchKernelGGL((k_trace4r<true, false, 0>), dim3(grid), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux, refillMin, waitMax);
    }
}

} // namespace flxd
