// trace4r.hip -- closest-hit and any-hit traversal of the 4-wide tree with PERSISTENT WAVES AND LANE REFILL (round 3).
//
// Why.  k_extend4 / k_shadow4 (trace4.hip) are VALU-issue-bound: 5.0e8 VALU wave-instructions per 4 M-ray launch at the ~4 SIMD-cycles most
// VALU instructions cost on gfx950 (scripts/ubench/valu_pairs*.hip: everything but mov / mul / add / logic ops issues on ONE of the SIMD's two
// pipes) IS the kernel time (0.85 ms), and only 26 % of the lanes of those instructions do work: a thread-per-ray wave lives as long as its
// longest ray (~25 node visits against a mean of 10.5) and, inside it, every descent round lasts as long as its longest descent.  Here
//   * the grid is the number of waves the machine holds at once; a wave takes 64-ray blocks of the queue on demand (one atomic per block on
//     one of eight per-XCD cursors: a single hot counter sustains only ~88 atomics/us) and hands the next rays of its block to lanes that
//     have finished as soon as `refillMin` of them are idle;
//   * a descent round ends when `waitMax` lanes stand on a leaf instead of when the last lane does (the lanes still descending simply go on
//     in the next round).
// Measured (kitchen, 4 M rays, refillMin 16, waitMax 32): VALU instructions 5.0e8 -> 2.8e8, lanes per instruction 16.7 -> 29.7, kernel time
// 0.844 -> 0.604 ms.  Refill alone (waitMax 64): 4.1e8 instructions, 0.754 ms.
//
// The commit.  Round 2 tried persistent waves with the full commit of traceExtension at the refill and lost (100 VGPRs, and the commit's
// instruction stream and memory round trips ran for a handful of lanes at a time); round 3 measured three more placements (archived in
// scripts/experiments/trace4r_commit_variants.hip.txt): by the finishing lanes at the refill without the light quad (72 VGPRs, but
// 0.60 -> 0.85 ms), parked in LDS and committed 64 at a time in completion order (0.90 ms) or in queue order (1.02 ms: too many partial
// flushes), and as a separate streaming pass over the queue (0.60 + 0.21 ms).  Every one costs more than it saves, because the commit is pure
// memory work (200 B per ray, three dependent round trips) that the thread-per-ray kernel hides under its VALU-bound waves and a
// latency-bound persistent kernel cannot.  So the kernel does not commit at all: a finished lane stores the RAW result
// {u, v, triangle, t} in the path's HITUV record (one 16-byte store) and the commit happens where the record is consumed anyway -- inside the
// next fused logic pass (logic.hip: k_logic<FUSE, RAW>: it has the ray in registers and needs the shading record next) -- or, if anything else
// looks first, in k_materialise (flx_trace.h: RAW HIT RECORDS; api.hip: settle).
//
// Per-ray arithmetic is exactly that of k_extend4 / k_shadow4 (same wide_node_visit / wide_leaf_visit, the ray's own stack column), so the
// results are bit-identical to those kernels whatever lane or wave a ray lands on (tests/test_gpu_wide.py).
// Replaces reference kernels traceExtension (src/wf_extrays.cl:5-36) and traceShadow (src/wf_shadowrays.cl:6-38).
#include "flx_trace4.h"
#include <cstdlib>

namespace flxd {

#ifndef WIDE_R_MIN_WAVES
#define WIDE_R_MIN_WAVES 1
#endif
#define R_NONE 0xFFFFFFFFu

#ifdef FLX_LAB_RSTATS
// lab build only (scripts/exp_lane_use.py): wave-level accounting of where the closest-hit kernel's instructions go -- stats[18] descent rounds,
// [19] lanes visiting a node in them, [20] leaf phases, [21] lanes on a leaf in them, [22] triangle-loop iterations, [23] lanes testing a triangle
unsigned long long *g_lab_rstats = nullptr;
#define RSTAT(k, v) do { if (ANY_HIT == false && rstats) { acc[k] += (unsigned long long)(v); } } while (0)
#else
#define RSTAT(k, v) do { } while (0)
#endif

template <bool ANY_HIT, int ANY_ORDER>
__global__ __launch_bounds__(WIDE_BLOCK, WIDE_R_MIN_WAVES) void k_trace4r(State st, Queues qs, Scene sc, flx_render_params p, TraceAux aux, int refillMin, int waitMax, uint32_t *cursor)
{
    __shared__ uint32_t s_stack[WIDE_LDS_LEVELS * WIDE_BLOCK];
#ifdef FLX_LAB_RSTATS
    unsigned long long *rstats = aux.stats; unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
#endif
    const int QID = ANY_HIT ? FLX_Q_SHADOW : FLX_Q_EXTENSION;
    const uint32_t qlen = ANY_HIT ? qs.counters[QID] : ext_len(qs);
    const uint32_t *queue = qs.q[QID];
    const uint32_t nblk = (qlen + 63u) >> 6;
    // Which 64-ray block next.  Rays differ in cost by an order of magnitude (sky vs. foliage), so a static share per wave (blocks w, w + G, ...)
    // leaves wave slots idle towards the end of the launch (courtyard: 1.60 -> 1.50 ms with this; kitchen and conference unchanged).  Blocks
    // are handed out on demand: list x = blocks x, x + 8, x + 16, ... belongs to XCD x (block b runs on XCD b % 8: rays that are neighbours in the queue
    // stay on one L2), a wave takes the next block of its XCD's list with ONE atomic per 64 rays (8 hot words at ~14 atomics/us each; a
    // single word saturates at ~88/us, and so do several words of one cache line: the cursors sit 256 B apart -- side by side behind the queue
    // counters they made the kernel 2.7 x slower) and moves on to the next XCD's list when its own is exhausted.  The ticket for the block
    // after the current one is taken when the current one is started, so the atomic's round trip is never waited for.
    uint32_t xcdList = blockIdx.x & 7u, listsTried = 0;
    auto issue_fetch = [&]() -> uint32_t {            // lane 0 takes a ticket of the current list; nobody waits for it here
        uint32_t k = 0;
        if (threadIdx.x == 0) k = atomicAdd(&cursor[xcdList * FLX_CURSOR_STRIDE], 1u);
        return k;
    };
    auto resolve = [&](uint32_t ticket) -> uint32_t { // wave-uniform: the block a ticket stands for, or the first block of another list, or none
        for (;;) {
            const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket) * 8u + xcdList;
            if (b < nblk) return b;
            if (++listsTried >= 8u) return 0xFFFFFFFFu;
            xcdList = (xcdList + 1u) & 7u;             // this list is exhausted: on to the next XCD's
            ticket = issue_fetch();
        }
    };
    uint32_t blk = resolve(issue_fetch()), pos = 0;  // wave-uniform: current block, rays of it already handed out
    if (blk >= nblk) return;
    uint32_t nextTicket = issue_fetch();             // (in flight while the current block is traced)
    // the path ids of the current block, one per lane, loaded when the block is taken: a refill then gets its ids from registers (one
    // cross-lane read) and waits for ONE memory round trip -- the rays -- instead of two
    uint32_t gidBlk = queue[min((blk << 6) + threadIdx.x, qlen - 1u)];

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
    uint32_t nT = 0;                                 // (visit counter of the shared leaf helper; unused without STATS)

    for (;;) {
        const bool idle = cur == FLX_RAY_DONE;
        const uint64_t idleMask = __ballot(idle);
        const uint32_t nIdle = (uint32_t)__popcll(idleMask);
        if ((int)nIdle >= refillMin || nIdle == 64u) {                   // wave-uniform
            if (idle && gid != R_NONE) {
                if (ANY_HIT) st.blocked[gid] = occluded ? 1u : 0u;
                else wr4(st.at(S_HITUV, gid), make_float4(ubest, vbest, __uint_as_float(FLX_RAW | (uint32_t)(tribest + 1)), tbest));      // flx_trace.h: RAW HIT RECORDS
                gid = R_NONE;
            }
            if (blk < nblk) {
                const uint32_t blkLen = min(64u, qlen - (blk << 6));
                const uint32_t avail = blkLen - pos;
                const uint32_t rank = mbcnt(idleMask);
                const uint32_t cand = (uint32_t)__shfl((int)gidBlk, (int)((pos + rank) & 63u), 64);       // (every lane takes part: ds_bpermute reads active lanes only)
                if (idle && rank < avail) {
                    gid = cand;
                    const float4 o4 = rd4(st.at(ANY_HIT ? S_SHO : S_ORIG, gid));
                    const float4 d4 = rd4(st.at(ANY_HIT ? S_SHD : S_DIR, gid));
                    r.setup(ld3(o4), ld3(d4), sc.wideClamp);
                    tbest = ANY_HIT ? o4.w : FLX_FLT_MAX;                // shadow: shadowRayLen
                    ubest = 0.0f; vbest = 0.0f; tribest = -1; occluded = false;
                    sp = 0; stk.base = 0; cur = sc.wrootRef;
                }
                pos += min(nIdle, avail);
                if (pos >= blkLen) {
                    blk = resolve(nextTicket); pos = 0;
                    if (blk < nblk) { nextTicket = issue_fetch(); gidBlk = queue[min((blk << 6) + threadIdx.x, qlen - 1u)]; }
                }
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
            RSTAT(0, 1); RSTAT(1, __popcll(mI));
            if (inner) wide_node_visit<ANY_HIT, ANY_ORDER>(wn, stk, r, tbest, sp, cur);
        }
#ifdef FLX_LAB_RSTATS
        { const uint64_t mL = __ballot(cur != FLX_RAY_DONE && (cur & FLX_WIDE_LEAF_BIT)); if (mL) { RSTAT(2, 1); RSTAT(3, __popcll(mL)); } }
#endif
        if (cur != FLX_RAY_DONE && (cur & FLX_WIDE_LEAF_BIT)) {
#ifdef FLX_LAB_RSTATS
            {   // the leaf visit of flx_trace4.h with the triangle loop counted (closest hit only)
                const float4 *lp = sc.wleaf + (cur & FLX_WIDE_OFF_MASK);
                const float4 b0 = lp[0], b1 = lp[1];
                const float bmin[3] = {b0.x, b0.y, b0.z}, bmax[3] = {b1.x, b1.y, b1.z};
                float tnear;
                bool hitAny = false;
                if (slab(bmin, bmax, r.orig, r.dinv, tbest, &tnear)) {
                    const int count = __float_as_int(b0.w);
                    const float4 *tp = lp + 2;
                    for (int k = 0; k < count; k++, tp += 3) {
                        if (ANY_HIT == false && rstats) { const uint64_t m_ = __ballot(true); if (lane_id() == (uint32_t)__ffsll((long long)m_) - 1u) { atomicAdd(&rstats[22], 1ull); atomicAdd(&rstats[23], (unsigned long long)__popcll(m_)); } }
                        float t, u, v;
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        if (moller_trumbore(r.orig, r.dir, ld3(a), ld3(b), ld3(c), &t, &u, &v) && t > 0.0f && t < tbest) {
                            if (ANY_HIT) { hitAny = true; break; }
                            tbest = t; ubest = u; vbest = v; tribest = __float_as_int(a.w);
                        }
                    }
                }
                if (hitAny) { occluded = true; cur = FLX_RAY_DONE; } else cur = stk.pop(sp);
            }
#else
            if (wide_leaf_visit<ANY_HIT, false>(sc.wleaf, r, cur, tbest, ubest, vbest, tribest, nT, nullptr)) { occluded = true; cur = FLX_RAY_DONE; }
            else cur = stk.pop(sp);
#endif
        }
    }
#ifdef FLX_LAB_RSTATS
    if (ANY_HIT == false && rstats && threadIdx.x == 0) for (int k = 0; k < 4; k++) atomicAdd(&rstats[18 + k], acc[k]);
#endif
}

// The commit of traceExtension for every path whose hit record is still raw (flx_trace.h: RAW HIT RECORDS): what k_logic<FUSE, RAW> does in
// registers, done in memory for whoever wants to look first.  Runs over all paths (a raw record says so itself).
__global__ __launch_bounds__(256) void k_materialise(State st, Scene sc, flx_render_params p)
{
    for (uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x; gid < st.numTasks; gid += gridDim.x * blockDim.x) {
        const float4 raw = rd4t(st.at(S_HITUV, gid));
        if (!hit_is_raw(__float_as_uint(raw.z))) continue;
        const float4 o4 = rd4(st.at(S_ORIG, gid));
        const float4 d4 = rd4(st.at(S_DIR, gid));
        const f3 orig = ld3(o4), dir = ld3(d4);
        const HitVals h = hit_values_raw(sc, p, orig, dir, raw);
        wr4(st.at(S_DIR, gid), mk4u(dir, __float_as_uint(d4.w) + 1u));            // pathLen += 1
        wr4(st.at(S_HITP, gid), mk4(h.P, h.t));
        const uint32_t keep = hit_keep_flags(d4.w, __float_as_uint(reinterpret_cast<const float *>(st.at(S_HITN, gid))[3]));
        wr4(st.at(S_HITN, gid), mk4u(h.N, h.flags | keep));
        wr4(st.at(S_HITUV, gid), make_float4(h.tu, h.tv, __int_as_float(h.tri), __int_as_float(h.matId)));
    }
}

// The area-light quad of the ANY-HIT query, applied after the traversal for scenes that have one (two more triangle tests with their corner
// set-up: inlined in k_trace4r they cost 30 VGPRs = three waves per SIMD; as a streaming pass ~40 B per ray).  The quad blocks like any
// triangle (src/wf_shadowrays.cl:32-33); the reference tests it first, the result is the OR either way.  (The closest-hit query's implicit
// light hit, src/wf_extrays.cl:28-29, is part of the commit: hit_values.)
__global__ __launch_bounds__(256) void k_lightfix4(State st, Queues qs, flx_render_params p)
{
    const uint32_t qlen = qs.counters[FLX_Q_SHADOW];
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < qlen; idx += gridDim.x * blockDim.x) {
        const uint32_t gid = qs.q[FLX_Q_SHADOW][idx];
        if (st.blocked[gid]) continue;
        const float4 o4 = rd4(st.at(S_SHO, gid)), d4 = rd4(st.at(S_SHD, gid));
        float tl = o4.w;                                                 // shadowRayLen
        if (light_quad(p.areaLight, ld3(o4), ld3(d4), &tl)) st.blocked[gid] = 1u;
    }
}

// grid = resident waves of the device for this kernel (occupancy x CUs), capped by the number of 64-ray blocks
template <class K>
static uint32_t persistent_grid(K kernel, int &cached, uint32_t numCUs, uint32_t numTasks, const char *capEnv = nullptr)
{
    if (cached == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(kernel), WIDE_BLOCK, 0) != hipSuccess || n <= 0) n = 16;
        cached = n;
    }
    int perCU = cached;
#ifdef FLX_LAB      // lab build only (-DFLX_LAB): A/B hooks that leave wave slots to a concurrent kernel (profiles/r03_slot_split_ab.txt)
    { static const char *e = getenv("FLX_PERSISTENT_WAVES_PER_CU"); if (e && atoi(e) > 0 && atoi(e) < perCU) perCU = atoi(e); }
    if (capEnv) { const char *e = getenv(capEnv); if (e && atoi(e) > 0 && atoi(e) < perCU) perCU = atoi(e); }                          // (per kernel family)
#else
    (void)capEnv;
#endif
    uint32_t g = numCUs * (uint32_t)perCU;
    const uint32_t blocks = (numTasks + 63u) / 64u;
    // A wave should get at least ~3 blocks of 64 rays: with fewer the launch is all ramp-up and tail (the refill has nothing to refill from).  Matters only for small
    // wavefronts -- the reference's own wfBufferSize of 2^20 paths is 16 384 blocks for 7 168 wave slots: capped at 14 ... 20 waves per CU the kitchen step gains 3.5 %
    // there (3750-3765 -> 3884-3896 Mrays/s), at 4 M paths and beyond the cap is not reached (profiles/r06_small_wavefront_grid.txt; fewer than 20 per CU lose at 4 M).
    if (blocks / 3u < g) g = blocks / 3u > numCUs ? blocks / 3u : numCUs;
    return g < blocks ? g : blocks;
}

// refill = refillMin | waitMax << 8.  Leaves RAW hit records behind (the caller remembers: api.hip, flx_ctx::rawHits).
void launch_extend4r(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill, uint32_t numCUs, int refill, uint32_t *cursor)
{
    const int refillMin = (refill & 0xFF) ? (refill & 0xFF) : 1, waitMax = ((refill >> 8) & 0xFF) ? ((refill >> 8) & 0xFF) : 64;      // (refillMin 0 would spin: api.hip, refill_value_ok)
    static int occ = 0;
    TraceAux aux{spill, ((st.numTasks + 255u) / 256u) * 256u, nullptr};
#ifdef FLX_LAB_RSTATS
    aux.stats = g_lab_rstats;
#endif
    const uint32_t grid = persistent_grid(k_trace4r<false, 0>, occ, numCUs, st.numTasks, "FLX_PERSISTENT_WAVES_EXT");
    hipLaunchKernelGGL((k_trace4r<false, 0>), dim3(grid), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux, refillMin, waitMax, cursor);
}

void launch_materialise(hipStream_t s, const State &st, const Scene &sc, const flx_render_params &p, uint32_t numCUs)
{
    hipLaunchKernelGGL(k_materialise, dim3(numCUs * 8), dim3(256), 0, s, st, sc, p);
}

void launch_shadow4r(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const flx_render_params &p, uint32_t *spill, uint32_t numCUs, int refill, uint32_t *cursor)
{
    const int refillMin = (refill & 0xFF) ? (refill & 0xFF) : 1, waitMax = ((refill >> 8) & 0xFF) ? ((refill >> 8) & 0xFF) : 64;
    static int occ[2] = {0, 0};
    TraceAux aux{spill, ((st.numTasks + 255u) / 256u) * 256u, nullptr};
    // visit order of the any-hit traversal (trace4.hip: launch_shadow4): far -> near when every shadow ray runs toward the environment light
    if (p.useEnvMap && !p.useAreaLight) {
        const uint32_t grid = persistent_grid(k_trace4r<true, 1>, occ[1], numCUs, st.numTasks, "FLX_PERSISTENT_WAVES_SHADOW");
        hipLaunchKernelGGL((k_trace4r<true, 1>), dim3(grid), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux, refillMin, waitMax, cursor);
    } else {
        const uint32_t grid = persistent_grid(k_trace4r<true, 0>, occ[0], numCUs, st.numTasks, "FLX_PERSISTENT_WAVES_SHADOW");
        hipLaunchKernelGGL((k_trace4r<true, 0>), dim3(grid), dim3(WIDE_BLOCK), 0, s, st, qs, sc, p, aux, refillMin, waitMax, cursor);
    }
    if (p.useAreaLight) hipLaunchKernelGGL(k_lightfix4, dim3(numCUs * 8), dim3(256), 0, s, st, qs, p);
}

} // namespace flxd
