// logic.hip -- the per-path state machine of the wavefront loop + deterministic queue building.
//
// Replaces reference kernel `logic` (src/wf_logic.cl:14-314) and its queue-append helpers
// (src/wf_logic.cl:322-519, src/utils.cl:328-358).  Per path, in the reference's order: russian
// roulette, zero-throughput termination, implicit environment / area-light hit with MIS, consume the
// previous vertex' NEE sample if its shadow ray was unblocked, splat + regenerate on termination,
// otherwise tangent-space normal, backface flip, next-event estimation (env-map alias sampling or
// area-light sampling) and material-queue selection.
//
// Queue building is NOT a per-thread global atomic (reference, non-NVIDIA) and not an inline-PTX
// warp aggregate (reference, NVIDIA): the kernel only records, per path, one byte of queue
// membership and, per 256-path block, seven counts (wave64 ballot + popcount).  A one-block scan
// turns the counts into offsets and a scatter kernel writes the queues with ballot/prefix ranks,
// i.e. a STABLE compaction: queue order = ascending path id = the order sequential execution of
// the reference produces.  Only the raygen queue's order is observable (pixel
// assignment), and it makes device runs bit-reproducible.
//
// FUSED = logic + the material kernels in one pass (flx_wf_logic ... flx_wf_materials with nothing but genRays between
// them, see api.hip): a continuing path goes straight on to its material step (flx_bsdf.h: material_step, the code the
// per-queue kernels run) with T, seed, hit record, normal and light direction still in registers -- the 96 B per path the
// material kernels re-read and the 32 B both kernels write twice never travel.  Waves are no longer BSDF-uniform (the
// reference's reason for the per-material queues), but the pass is bound by streaming the path state, not by the BSDF
// arithmetic.  Queue contents, counters and the extension-queue order are the ones the separate kernels produce: the
// scatter kernel appends the material lists to the extension queue at the slots the material kernels would compute.
#include "flx_bsdf.h"
#include "flx_trace.h"          // hit_values_raw: the commit of traceExtension for RAW hit records

namespace flxd {

#ifndef LOGIC_BLOCK
#define LOGIC_BLOCK 256
#endif
// FUSE = the BSDF types the fused pass evaluates inline (0: none, the plain logic kernel); paths of the other types take the usual
// route through their material queue and the `k_material_rest` kernel (material.hip).  Inlining costs registers for every type compiled
// in (logic alone 68 VGPRs; + diffuse 94; + glossy 104; all six 106 = 4 waves/SIMD, 87 with the re-reads below), so the host picks the set per scene from the
// BSDF types its triangles use (api.hip) -- the reference specialises its kernels per scene the same way (-DBXDF_USE_*, src/clcontext.cpp).
__host__ __device__ constexpr bool fuse_inlines_list(int fuse, uint32_t ml)     // ml = material_list(): 1 diffuse .. 5 delta
{
    return ml == 1u ? (fuse & USE_DIFFUSE) != 0 : ml == 2u ? (fuse & USE_GLOSSY) != 0 : ml == 3u ? (fuse & USE_GGX_REFL) != 0
         : ml == 4u ? (fuse & USE_GGX_REFR) != 0 : ml == 5u ? (fuse & USE_DELTA) != 0 : false;
}
// membership byte: bit0 raygen, bit1 shadow, bits 2..4 material queue (0 none, 1 diffuse, 2 glossy,
// 3 ggx reflection, 4 ggx refraction, 5 delta)
#define NUM_LISTS 7   // raygen, shadow, 5 material

// LOGIC_FULL_STORES (round 5): in the fused RAW pass every lane stores every record of the path state, with ONE store instruction per record.
// The pass is the one memory-bound kernel of the loop, and what bounded it was not bandwidth, latency chains, VALU issue or the store acknowledgements
// (profiles/r05_logic_variants_ab.txt, r05_logic_probes_ab.txt: early loads by LDS-DMA, stores behind the last load, +20 % dummy VALU, the implicit
// environment sample removed -- none of them moves its time) but PARTIAL WRITES: the terminating lanes (a fifth of them, at random positions) wrote 3
// of the 12 records, so nine 16-byte-per-lane store streams had holes -- partial 128-byte lines and partial 32-byte sectors, which the memory side
// has to merge with what is there.  scripts/ubench/logic_stream.hip reproduces it with no arithmetic at all: the pass's access pattern streams in
// 0.46-0.55 ms with full-wave stores and in 0.62-0.81 ms when 21 % of the lanes skip nine of the twelve (fewer bytes, +35-45 % time); two divergent
// non-temporal partial stores that together cover the lines are slower still (0.75), only write-back stores get merged in L2 (0.49).
// So the holes are filled.  The RAW pass runs only when genRays and the material kernels follow in the same chain with nothing looking in between
// (api.hip: runLogic), which makes most of a non-owning lane's records DEAD -- a terminating path is regenerated (hit record, normal, lastEmission,
// lastBsdf, origin, direction: rewritten or covered by the regeneration flags, flx_device.h), a path of a non-inlined BSDF type gets lastBsdf, lastT,
// origin from its material kernel -- and those take whatever the lane has in registers; the few members that must survive (shadow origin / direction and
// lastT of a terminating path: genRays does not reset them, src/wf_raygen.cl:77-96) are loaded and written back (+ ~10 bytes per path of reads).
#ifndef LOGIC_FULL_STORES
#define LOGIC_FULL_STORES 1
#endif
// LOGIC_REGROUP (round 5): the all-types pass runs the material step BSDF-UNIFORM.  Inlining every BSDF type means a wave runs every type's code one after
// the other with the lanes of the other types idle -- the pass is instruction-bound at 19 of 64 lanes on the conference scene (three types of similar
// weight) -- which is the very thing the reference's per-material queues exist to avoid.  Here the 256 paths of a block hand the material step's inputs
// (20 words: hit point, normal, uv, material, ray, light direction, throughput, seed) through LDS, sorted by BSDF type with ballot / popcount ranks, each
// lane runs the step of the item in ITS slot -- a wave then holds one type, two at a boundary -- and the 16 result words travel back the same way.  Same
// arithmetic per path, whichever lane does it: bit-identical.  Needs every thread of the block inside the pass (block barriers): launch_logic takes it only
// when the path count is a multiple of the block size and `first` is off.  It is a template parameter, not a run-time flag, and compiled for 5 blocks per CU
// (LOGIC_REGROUP_MIN_BLOCKS): round 5's run-time-flag version carried the worker item's 20 inputs on top of the owner's state in EVERY all-types pass -- 119 VGPRs
// against 92, 4 instead of 5 waves per SIMD -- and won only where one BSDF type dominates (courtyard +3.8 %, egyptcat +3.4 %, conference -0.5 %).  As its own
// instance under the 5-wave bound the allocator fits it into 96 VGPRs with no scratch (34 SGPRs spilled to lanes), 20.7 KB of LDS per block, and the picture turns
// round: conference 5392 -> 5678 / 5376 -> 5654 Mrays/s (+5.2 %: three types of similar weight, the case the reference's per-material queues exist for), courtyard
// and egyptcat within +-0.5 % (profiles/r06_regroup_ab.txt; the 100-VGPR build without the bound, 4 waves: -2 ... -4 % everywhere).  Shipped on for every
// all-types pass (api.hip: flx_upload_scene, option "regroup").
#ifndef LOGIC_REGROUP
#define LOGIC_REGROUP 1
#endif
#ifndef LOGIC_LAZY_HITN
#define LOGIC_LAZY_HITN 1
#endif
#ifndef LOGIC_REGROUP_MIN_BLOCKS      // blocks per CU the regrouped pass is compiled for (x 4 waves per block / 4 SIMDs = waves per SIMD)
#define LOGIC_REGROUP_MIN_BLOCKS 5
#endif

struct LogicAux {
    uint8_t *member;          // numTasks
    uint32_t *blockCounts;    // NUM_LISTS x numBlocks
    uint32_t *blockOffsets;   // NUM_LISTS x numBlocks
    uint32_t numBlocks;
    uint32_t stride;          // elements between two lists in blockCounts / blockOffsets: numBlocks rounded up for the scan's uint4 accesses
    // in-kernel regeneration (k_logic<FUSE, RAW>, LOGIC_REGEN): one status word per wave for the decoupled look-back over the terminating paths
    unsigned long long *lookback;
    uint32_t epoch;           // launch number: a status word counts only if it carries this launch's epoch (no reset between launches)
    uint32_t regen;           // 1: terminating lanes are regenerated here (the genRays of this chain is not launched) | 2: PREPARED for the genRays that follows (below) | 0: off
    uint32_t appendExt;       // regen: append the regenerated paths to the extension queue at extBase + rank (genRays' appendExt)
    uint32_t *error;          // set when a look-back gives up (never observed; a hang would take the box down, a flag fails the test)
};

// ---- decoupled look-back over waves (Merrill & Garland 2016), for ONE running count: how many paths with a smaller id terminate in this pass = the
// index genRays' queue would give the path = its pixel (src/wf_raygen.cl:25).  Status word: bits 0..25 value | 26..27 flag (1 the wave's own count,
// 2 the inclusive prefix up to and including the wave) | 28..63 epoch.  Waves are dispatched in id order, a wave publishes its count before it
// looks back and never waits for a later wave: no deadlock; the spin is bounded all the same.
#define LB_VALUE(x) ((uint32_t)((x) & 0x3FFFFFFull))
#define LB_FLAG(x) ((uint32_t)(((x) >> 26) & 3ull))
#define LB_EPOCH(x) ((uint32_t)((x) >> 28))
#define LB_PACK(e, f, v) (((unsigned long long)(e) << 28) | ((unsigned long long)(f) << 26) | (unsigned long long)(v))
// sum of a value <= 127 over the ACTIVE lanes where `take` holds (ballots only see active lanes: right for the partial last wave too)
__device__ __forceinline__ uint32_t wave_sum7(uint32_t v, bool take)
{
    uint32_t s = 0u;
    #pragma unroll
    for (int b = 0; b < 7; b++) s += (uint32_t)__popcll(__ballot(take && ((v >> b) & 1u))) << b;
    return s;
}
// the calling lanes are lanes 0 .. win - 1 of the wave (a full wave, or the head of the last one); returns the exclusive prefix of this wave's count
__device__ __forceinline__ uint32_t lookback_exclusive(const LogicAux &aux, uint32_t wave, uint32_t count)
{
    const uint32_t lane = lane_id();
    const uint32_t win = (uint32_t)__popcll(__ballot(true));
    unsigned long long *st = aux.lookback;
    if (lane == 0u) __hip_atomic_store(st + wave, LB_PACK(aux.epoch, wave == 0u ? 2u : 1u, count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0u) return 0u;
    uint32_t sum = 0u;
    long long j = (long long)wave - 1;                   // nearest predecessor not yet accounted for
    for (uint32_t spins = 0;;) {
        const long long idx = j - (long long)lane;
        unsigned long long v = LB_PACK(aux.epoch, 2u, 0u);                     // below wave 0: prefix 0
        if (idx >= 0) v = __hip_atomic_load(st + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = LB_EPOCH(v) == aux.epoch && LB_FLAG(v) != 0u;
        const uint64_t b2 = __ballot(ok && LB_FLAG(v) == 2u), bad = __ballot(!ok);
        const uint32_t first2 = b2 ? (uint32_t)__ffsll((long long)b2) - 1u : win;            // nearest predecessor that already has its prefix
        const uint64_t need = first2 >= 64u ? ~0ull : ((1ull << first2) | ((1ull << first2) - 1ull));
        if (bad & need) {                                                       // someone nearer has not published yet
            if (++spins > (1u << 22)) { if (lane == 0u) *aux.error = 1u; break; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        sum += wave_sum7(LB_VALUE(v), lane < first2);                           // the own counts (<= 64 each) of the waves in between ...
        if (b2) { sum += (uint32_t)__shfl((int)LB_VALUE(v), (int)first2, 64); break; }   // ... on top of the nearest published prefix
        j -= (long long)win;
    }
    if (lane == 0u) __hip_atomic_store(st + wave, LB_PACK(aux.epoch, 2u, sum + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return sum;
}

__device__ __forceinline__ uint32_t material_list(int type, uint32_t separate)
{
    if (!separate) return 1u;                       // WF_SINGLE_MAT_QUEUE: everything in the diffuse queue
    switch (type) {
    case FLX_BXDF_DIFFUSE: return 1u;
    case FLX_BXDF_GLOSSY: return 2u;
    case FLX_BXDF_GGX_ROUGH_REFLECTION: return 3u;
    case FLX_BXDF_GGX_ROUGH_DIELECTRIC: return 4u;
    case FLX_BXDF_IDEAL_REFLECTION:
    case FLX_BXDF_IDEAL_DIELECTRIC: return 5u;
    default: return 0u;                             // reference prints an error and drops the path
    }
}

// RAW: paths may carry RAW hit records (flx_trace.h), left by the persistent-wave extension kernel (trace4r.hip).  The pass then does the
// commit of traceExtension itself, in registers -- shading-record gather, normal / uv interpolation, pathLen + 1 -- right where it would
// have loaded the committed record, and stores what the committed state holds (hit point and uv record of the continuing paths; the normal
// record is written here anyway).  Only used when genRays follows in the same fused chain: a path that terminates is regenerated there, and
// a regenerated path's hit record is dead (flx_device.h: REGENERATED PATHS), so nothing of it has to be stored but the record's raw marker
// cleared.
#ifndef LOGIC_MIN_BLOCKS        // blocks per CU the register allocator must leave room for (x LOGIC_BLOCK / 256 waves per SIMD); 0: whatever it needs
#define LOGIC_MIN_BLOCKS 0
#endif
template <int FUSE, bool RAW = false, bool REGROUP = false>
__global__ __launch_bounds__(LOGIC_BLOCK, (REGROUP ? LOGIC_REGROUP_MIN_BLOCKS : LOGIC_MIN_BLOCKS) > 0 ? (REGROUP ? LOGIC_REGROUP_MIN_BLOCKS : LOGIC_MIN_BLOCKS) : 1) void k_logic(State st, Scene sc, Frame fr, flx_render_params p, LogicAux aux, Queues qs, uint32_t firstIteration)
{
    const uint32_t gid = blockIdx.x * LOGIC_BLOCK + threadIdx.x;
    uint32_t maxId = st.numTasks;
    if (firstIteration) { uint32_t npix = p.width * p.height; maxId = npix < maxId ? npix : maxId; }
    uint32_t member = 0u;
    __shared__ uint32_t s_item[REGROUP ? 20 : 1][LOGIC_BLOCK];                        // [word][slot]: the material step's inputs, then its results
    __shared__ uint32_t s_tc[5][LOGIC_BLOCK / 64];                                    // per wave: lanes with work of BSDF class 0..4

    if (gid < maxId) {
        const float4 thr = rd4(st.at(S_THR, gid));
        const float4 d4 = rd4(st.at(S_DIR, gid));
        const float4 o4 = rd4(st.at(S_ORIG, gid));
        const float4 huv = rd4(st.at(S_HITUV, gid));
        // The old normal record is needed only by a path WITHOUT a RAW hit record (committed by k_materialise, or never traced since its regeneration): a RAW path's
        // normal and flags come from the commit below, and the one bit it would keep from the old record (backfaceHit, hit_keep_flags) is rewritten by every store of
        // this pass.  In the RAW pass almost every path is raw: 16 B per path less to read (LOGIC_LAZY_HITN; round 4 measured it inside the noise of a pass that was
        // still paying for partial writes -- profiles/r04_lazy_hitn_ab.txt; round 6: profiles/r06_lazy_hitn_ab.txt).
        float4 hn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!(RAW && LOGIC_LAZY_HITN != 0 && hit_is_raw(__float_as_uint(huv.z)))) hn = rd4(st.at(S_HITN, gid));
        const float4 ei4 = rd4(st.at(S_EI, gid));
        uint32_t seed = __float_as_uint(thr.w);
        const uint32_t pixIdx = __float_as_uint(ei4.w) & ~FLX_FRESH;
        float eiw = ei4.w;                                                  // pixel index + "no NEE sample since regeneration"
        const f3 rayOrig = ld3(o4), rayDir = ld3(d4);
        const float lastPdfW = o4.w;
        f3 T = ld3(thr);
        f3 Ei = ld3(ei4);
        f3 hitN = ld3(hn);
        uint32_t hflags = __float_as_uint(hn.w);
        int hitI = __float_as_int(huv.z), hitMat = __float_as_int(huv.w);
        f2 hitUV = mk2(huv.x, huv.y);
        uint32_t lenBits = __float_as_uint(d4.w);
        bool isRaw = false; f3 rawP = mk3(0.0f); float rawT = 0.0f;
        if (RAW && hit_is_raw(__float_as_uint(huv.z))) {                    // the commit of traceExtension (flx_trace.h: RAW HIT RECORDS)
            isRaw = true;
            const HitVals h = hit_values_raw(sc, p, rayOrig, rayDir, huv);
            hflags = h.flags | hit_keep_flags(d4.w, hflags);
            hitN = h.N; hitI = h.tri; hitMat = h.matId; hitUV = mk2(h.tu, h.tv); rawP = h.P; rawT = h.t;
            lenBits += 1u;                                                  // pathLen += 1
        }
        const uint32_t len = lenBits & ~FLX_FRESH;                         // flag bits of a regenerated path: flx_device.h
        // A regenerated path carries genRays' reset values in the members its flags cover (flx_device.h: REGENERATED PATHS) -- values the slot
        // does not hold, because k_raygen skips those stores.  The reference's loop never runs `logic` on a path before an extension kernel has
        // traced it, but a host may (logic -> genRays -> logic): such a path must see EMPTY_HIT (src/wf_raygen.cl:95-96) and terminate as a
        // miss like the reference's, not shade the slot's previous hit.  Found by tests/test_gpu_fuzz.py (round 4).
        const bool fresh = (lenBits & FLX_FRESH) != 0u;                    // no material kernel since regeneration: lastSpecular is genRays' 1
        if (fresh && len == 0u && !isRaw) { hitI = -1; hitMat = -1; hitN = mk3(0.0f); hflags = 0u; hitUV = mk2(0.0f, 0.0f); }
        bool Tdirty = false;

        // russian roulette (src/wf_logic.cl:60-69)
        bool terminate = (len >= p.maxBounces + 1u);
        if (terminate && p.useRoulette) {
            float contProb = clampf(luminance(T), 0.01f, 0.5f);
            terminate = (rand01(&seed) > contProb);
            T = T / contProb;
            Tdirty = true;
        }
        if (is_zero(T) || lastPdfW == 0.0f) terminate = true;        // :72-73

        if (hitI < 0 && !terminate) {                                 // implicit environment sample, :84-107
            float weight = 1.0f;
            const bool lastSpecular = fresh || __float_as_uint(rd4(st.at(S_LT, gid)).w) != 0u;
            f3 bg = mk3(0.0f);
            if (p.useEnvMap && (len == 1u || p.sampleImpl)) bg = eval_env_dir(sc, rayDir) * p.envMapStrength;
            if (p.sampleImpl && p.sampleExpl && p.useEnvMap && len > 1u && !lastSpecular) {
                float lightPickProb = st.pickProb[gid];
                float directPdfW = env_map_pdf(sc, rayDir);
                weight = (lastPdfW * lightPickProb) / (lastPdfW * lightPickProb + directPdfW);
            }
            Ei = Ei + weight * T * bg;
            terminate = true;
        } else if (p.useAreaLight && (hflags & 1u) && !terminate) {   // implicit area-light sample, :111-131
            float misWeight = 1.0f;
            const bool lastSpecular = fresh || __float_as_uint(rd4(st.at(S_LT, gid)).w) != 0u;
            const f3 hitP = (RAW && isRaw) ? rawP : ld3(rd4(st.at(S_HITP, gid)));
            if (p.sampleExpl && len > 1u && !lastSpecular) {
                float directPdfA = 1.0f / (4.0f * p.areaLight.size.x * p.areaLight.size.y);
                float directPdfW = pdf_a_to_w(directPdfA, length(hitP - rayOrig), dot(normalize(-rayDir), hitN));
                float lightPickProb = st.pickProb[gid];
                misWeight = lastPdfW / (lastPdfW + directPdfW * lightPickProb);
            }
            Ei = Ei + T * misWeight * V(p.areaLight.E);
            terminate = true;
        }

        // consume the light sample generated at the previous vertex (:135-156)
        // (fetching these four records with the first batch of loads, unconditionally, instead of behind shadowRayBlocked: 110 VGPRs with the
        //  commit's shading record in flight, no gain -- profiles/r03_logic_nee_early_ab.txt)
        const uint32_t blockedOld = st.blocked[gid];
        if (blockedOld == 0u) {
            float4 le = rd4(st.at(S_LEMIT, gid));
            float4 lb = rd4(st.at(S_LBSDF, gid));
            const float4 lt = rd4(st.at(S_LT, gid));
            float directPdfW = rd4(st.at(S_SHD, gid)).w;
            // members a regenerated path's flags cover hold genRays' reset values (flx_device.h), not what the slot stores: lastBsdf /
            // lastPdfImplicit until a material kernel has run (logic -> genRays -> extension -> logic -> shadow -> logic without the material
            // kernels consumes the light sample with lastBsdf = 0: found by tests/test_gpu_fuzz.py), lastEmission / lastCosTh / lastPdfDirect
            // until logic has sampled a light
            if (fresh) lb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if ((__float_as_uint(ei4.w) & FLX_FRESH) != 0u) { le = make_float4(0.0f, 0.0f, 0.0f, 0.0f); directPdfW = 0.0f; }
            const float lightPickProb = st.pickProb[gid];
            const float cosTh = le.w, bsdfPdfW = lb.w;
            float weight = 1.0f;
            if (p.sampleImpl) weight = (directPdfW * lightPickProb) / (directPdfW * lightPickProb + bsdfPdfW);
            f3 contrib = ld3(lb) * ld3(lt) * ld3(le) * weight * cosTh / (lightPickProb * directPdfW);
            Ei = Ei + contrib;
        }

        // ---- FULL: every record of the path state is stored by EVERY lane, in one store instruction per record (LOGIC_FULL_STORES, see the macro).
        // Lane classes at the store phases:  T terminating (regenerated by the genRays that follows) | C1 continuing, BSDF inlined here |
        // C2 continuing, served by the material kernel that follows | C0 continuing, no queue (unknown BSDF type: the reference drops the path).
        constexpr bool FULL = LOGIC_FULL_STORES != 0 && RAW && FUSE != 0;
        // ---- REGEN (aux.regen): a terminating lane is regenerated HERE instead of by the genRays kernel of the chain -- its new origin, direction,
        // radiance + pixel and throughput + seed records ride on the full-line stores below, where genRays wrote four isolated 16-byte records per
        // path (partial sectors again: 0.15-0.19 ms per iteration for 1.8 M paths).  The one thing genRays has that a lane here has not is its index
        // in the raygen queue (= pixel, src/wf_raygen.cl:25): the number of terminating paths with a smaller id -- a decoupled look-back over the waves.
        uint32_t regenRank = 0u, regenLocal = 0u;
        // ---- PREPARED REGENERATION (aux.regen == 2; round 6): what a regenerated path needs is all a function of its seed -- jitter, thin-lens origin, the
        // reset values -- EXCEPT the pixel (its index in the raygen queue), i.e. except the direction and the pixel word.  k_raygen is bound by its isolated
        // 16-byte stores into scattered paths (seven per path: four records + three scalars), while this pass stores every record of every lane
        // anyway: the terminating lane computes jitter + lens origin here, its origin / throughput + new seed / shadowRayBlocked / lastLightPickProb go out with
        // the full-line stores below, the jitter and the seed after it travel in the (dead) direction record, and the k_raygen of the chain only reads that
        // record and stores the direction and the pixel (misc.hip: k_raygen, prepared): two scattered stores per path instead of seven.  No look-back needed.
        const bool regen = FULL && aux.regen == 1u;
        const bool prep = FULL && aux.regen == 2u;
        if (regen) {
            const uint64_t tb = __ballot(terminate);
            regenRank = lookback_exclusive(aux, gid >> 6, (uint32_t)__popcll(tb)) + mbcnt(tb);
            regenLocal = (*fr.currPixelIdx + regenRank) % fr.localPixels;
        }
        flx_material mat;
        bool backface = false, haveL = false, neeBlocked = false, inlined = false;
        f3 hitP = rawP, orig = mk3(0.0f), Lnee = mk3(0.0f), neeLi = mk3(0.0f);
        float hitT = rawT, neeLen = 0.0f, neePdf = 0.0f, neeCos = 0.0f, neePick = 0.0f;
        uint32_t ml = 0u;
        auto splat = [&]() {                                          // splat + regenerate, :163-177
#ifndef FLX_LAB_LOGIC_NOSPLAT      // lab build only (RESULTS INVALID: no image): the pass without its 4 float atomics per terminating path
            if (len > 0u) {
                float *px = fr.pixels + (size_t)pixIdx * 4;
                unsafeAtomicAdd(px + 0, Ei.x); unsafeAtomicAdd(px + 1, Ei.y);
                unsafeAtomicAdd(px + 2, Ei.z); unsafeAtomicAdd(px + 3, 1.0f);
            }
#endif
        };
        if (terminate) {
            if (!FULL) {
                splat();
                wr4(st.at(S_EI, gid), mk4(Ei, ei4.w));
                wr4(st.at(S_THR, gid), mk4u(T, seed));
                if (RAW && isRaw) wr4(st.at(S_HITUV, gid), make_float4(hitUV.x, hitUV.y, __int_as_float(hitI), __int_as_float(hitMat)));      // (clears the raw marker)
            }
            member = 1u;
        } else {
            mat = sc.materials[hitMat];                               // :180-184
            hitN = tangent_space_normal(sc, hitN, hitUV, hitI, mat.map_N);
            backface = dot(hitN, rayDir) > 0.0f;
            if (backface) hitN = hitN * -1.0f;
            if (!(RAW && isRaw)) { const float4 hp = rd4(st.at(S_HITP, gid)); hitP = ld3(hp); hitT = hp.w; }
            if (!FULL && RAW && isRaw) {                              // the committed hit record: point and uv here, the normal below
                if (FUSE == USE_ALL) {                                // (re-read at the material step below: keep the lines)
                    wr4t(st.at(S_HITP, gid), mk4(rawP, rawT));
                    wr4t(st.at(S_HITUV, gid), make_float4(hitUV.x, hitUV.y, __int_as_float(hitI), __int_as_float(hitMat)));
                } else {
                    wr4(st.at(S_HITP, gid), mk4(rawP, rawT));
                    wr4(st.at(S_HITUV, gid), make_float4(hitUV.x, hitUV.y, __int_as_float(hitI), __int_as_float(hitMat)));
                }
            }
            orig = hitP - 1e-3f * rayDir;
            if (fr.aovNormal) {                                       // denoiser features, :186-209
                if (len == 1u) {
                    const f3 n = camera_space_normal(p, hitN);
                    float *px = fr.aovNormal + (size_t)pixIdx * 4;
                    unsafeAtomicAdd(px + 0, n.x); unsafeAtomicAdd(px + 1, n.y); unsafeAtomicAdd(px + 2, n.z); unsafeAtomicAdd(px + 3, 1.0f);
                }
                if (!FLX_BXDF_IS_SINGULAR(mat.type) && !st.firstDiffuse[gid]) {
                    st.firstDiffuse[gid] = 1u;
                    const f3 albedo = mat_float3(sc, V(mat.Kd), hitUV, mat.map_Kd);              // not gamma-corrected
                    float *px = fr.aovAlbedo + (size_t)pixIdx * 4;
                    unsafeAtomicAdd(px + 0, albedo.x); unsafeAtomicAdd(px + 1, albedo.y); unsafeAtomicAdd(px + 2, albedo.z); unsafeAtomicAdd(px + 3, 1.0f);
                }
            }
            if (!FULL) wr4(st.at(S_HITN, gid), mk4u(hitN, (hflags & 1u) | (backface ? 2u : 0u)));   // :212-213

            if (p.sampleExpl && !FLX_BXDF_IS_SINGULAR(mat.type)) {    // next event estimation, :217-302
                uint32_t den = p.useEnvMap + p.useAreaLight; if (den < 1u) den = 1u;
                const float envMapProb = (float)p.useEnvMap / (float)den;
                const bool useEnvMap = rand01(&seed) < envMapProb;
                const bool useAreaLight = !useEnvMap && p.useAreaLight;
                if (useEnvMap && p.useEnvMap) {
                    f3 L; float directPdfW = 0.0f;
#ifdef FLX_LAB_LOGIC_NONEE          // lab build only (RESULTS INVALID: radiance and the shadow rays' directions; the paths' flow is untouched): the pass without
                                    // the environment map's table / texel gathers of the light sample (3 + 4 scattered lines per continuing lane)
                    { const float r = rand01(&seed); L = mk3(r - 0.5f, 0.7f, 0.3f - r); directPdfW = 0.1f + r; }
                    const float lenL = 2.0f * p.worldRadius;
                    L = normalize(L);
                    const float cosTh = fmaxf_(0.0f, dot(L, hitN));
                    const f3 envMapLi = mk3(directPdfW, cosTh, 0.5f) * p.envMapStrength;
#else
                    // (sample_env_alias, normalize, eval_env_dir: precomputed per texel, flx_device.h neeRec -- the same values bit for bit)
                    const int uvInd = sample_env_index(sc, rand01(&seed));
                    const float4 n0 = sc.neeRec[2 * (size_t)uvInd], n1 = sc.neeRec[2 * (size_t)uvInd + 1];
                    L = ld3(n0); directPdfW = n0.w;
                    const float lenL = 2.0f * p.worldRadius;
                    const float cosTh = fmaxf_(0.0f, dot(L, hitN));
                    const f3 envMapLi = ld3(n1) * p.envMapStrength;
#endif
                    neeLen = lenL; neePdf = directPdfW; neeLi = envMapLi; neeCos = cosTh; neePick = envMapProb;
                    haveL = true; Lnee = L;
                }
                if (useAreaLight) {
                    const float lightPickProb = 1.0f - envMapProb;
                    const float directPdfA = 1.0f / (4.0f * p.areaLight.size.x * p.areaLight.size.y);
                    f3 posL = V(p.areaLight.pos);
                    const float r1 = 2.0f * rand01(&seed) - 1.0f;
                    const float r2 = 2.0f * rand01(&seed) - 1.0f;
                    posL = posL + r1 * p.areaLight.size.x * V(p.areaLight.right);
                    posL = posL + r2 * p.areaLight.size.y * V(p.areaLight.up);
                    f3 L = posL - orig;
                    const float lenL = length(L) * 0.995f;
                    L = normalize(L);
                    const float cosLight = fmaxf_(dot(V(p.areaLight.N), -L), 0.0f);
                    if (cosLight > 0.0f) {
                        const float directPdfW = pdf_a_to_w(directPdfA, lenL, cosLight);
                        const float cosTh = fmaxf_(0.0f, dot(L, hitN));
                        neeLen = lenL; neePdf = directPdfW; neeLi = V(p.areaLight.E); neeCos = cosTh; neePick = lightPickProb;
                        haveL = true; Lnee = L;
                    } else {
                        neeBlocked = true;
                    }
                }
                if (haveL) {                                          // the light sample (:262-268 / :289-295)
                    eiw = __uint_as_float(pixIdx);
                    member |= 2u;
                    if (!FULL) {
                        wr4(st.at(S_SHO, gid), mk4(orig, neeLen));
                        wr4(st.at(S_SHD, gid), mk4(Lnee, neePdf));
                        wr4(st.at(S_LEMIT, gid), mk4(neeLi, neeCos));
                        st.pickProb[gid] = neePick;
                    }
                }
                if (!FULL && neeBlocked) st.blocked[gid] = 1u;        // (a few lanes with an area light behind the surface; FULL: with store phase B)
            }
            if (!FULL) wr4(st.at(S_EI, gid), mk4(Ei, eiw));
            (void)Tdirty;
            ml = material_list(mat.type, p.wfSeparateQueues);
            member |= ml << 2;
            inlined = FUSE != 0 && fuse_inlines_list(FUSE, ml);
        }

        f3 Lold = mk3(0.0f);                                          // the stored light direction of a lane without a new one
        if (FULL) {
            // ---- store phase A (all lanes): hit record, normal, light sample, radiance.  A terminating lane's hit record, normal and lastEmission
            // are dead (the genRays of this chain regenerates it: flx_device.h REGENERATED PATHS) and take whatever the registers hold; its shadow
            // origin / direction are NOT reset by genRays (src/wf_raygen.cl:77-96) and are written back as loaded -- and so are the records of a
            // continuing lane that sampled no light.
            float4 sho = mk4(orig, neeLen), shd = mk4(Lnee, neePdf), le = mk4(neeLi, neeCos);
            float pick = neePick;
#ifdef FLX_LAB_LOGIC_NOOLD          // lab build only (RESULTS INVALID: exported shadow records of regenerated paths): the pass without the write-back loads
            if (!haveL && !terminate) {
#else
            if (!haveL) {
#endif
                sho = rd4(st.at(S_SHO, gid)); shd = rd4(st.at(S_SHD, gid));
                if (!terminate) { le = rd4(st.at(S_LEMIT, gid)); pick = st.pickProb[gid]; } else pick = 1.0f;      // (T: genRays' lastLightPickProb)
                Lold = ld3(shd);
            }
            const float4 hp4 = mk4(hitP, hitT), huv4 = make_float4(hitUV.x, hitUV.y, __int_as_float(hitI), __int_as_float(hitMat));
            if (FUSE == USE_ALL) { wr4t(st.at(S_HITP, gid), hp4); wr4t(st.at(S_HITUV, gid), huv4); }      // (re-read at the material step below: keep the lines)
            else { wr4(st.at(S_HITP, gid), hp4); wr4(st.at(S_HITUV, gid), huv4); }
            wr4(st.at(S_HITN, gid), mk4u(hitN, (hflags & 1u) | (backface ? 2u : 0u)));
            wr4(st.at(S_SHO, gid), sho);
            wr4(st.at(S_SHD, gid), shd);
            wr4(st.at(S_LEMIT, gid), le);
            st.pickProb[gid] = pick;
            // (REGEN, T: genRays' Ei = 0 and the new pixel with the "no NEE sample since regeneration" flag; the splat below still has the old Ei)
            wr4(st.at(S_EI, gid), (regen && terminate) ? mk4u(mk3(0.0f), FLX_FRESH | regenLocal) : mk4(Ei, eiw));
        }

        MatStep o;
        o.bsdfNEE = mk3(0.0f); o.bsdfPdfW = 0.0f; o.singular = 0u; o.newT = T; o.orig = mk3(0.0f); o.pdfW = 0.0f; o.newDir = mk3(0.0f);
        bool regrouped = false;
        if constexpr (REGROUP) {                                      // (compile-time: k_logic<USE_ALL, true, true>; every thread of the block is here: launch_logic)
            regrouped = true;
            const bool work = !terminate && inlined;
            const uint32_t wv = threadIdx.x >> 6;
            uint32_t cls = 4u;                                        // BSDF class 0 diffuse .. 4 delta / anything else
            if (work) { const uint32_t c_ = material_list(mat.type, 1u); cls = c_ ? c_ - 1u : 4u; }
            uint64_t bt[5];
            #pragma unroll
            for (int t = 0; t < 5; t++) bt[t] = __ballot(work && cls == (uint32_t)t);
            if ((threadIdx.x & 63u) == 0u) { for (int t = 0; t < 5; t++) s_tc[t][wv] = (uint32_t)__popcll(bt[t]); }
            __syncthreads();
            uint32_t slot = 0u, total = 0u;
            #pragma unroll
            for (int t = 0; t < 5; t++) {
                uint32_t before = 0u, all = 0u;
                for (uint32_t w = 0; w < LOGIC_BLOCK / 64; w++) { const uint32_t n_ = s_tc[t][w]; all += n_; if (w < wv) before += n_; }
                if (cls == (uint32_t)t) slot = total + before + mbcnt(bt[t]);
                total += all;
            }
            if (work) {
                const f3 L = haveL ? Lnee : Lold;
                s_item[0][slot] = __float_as_uint(hitP.x); s_item[1][slot] = __float_as_uint(hitP.y); s_item[2][slot] = __float_as_uint(hitP.z);
                s_item[3][slot] = __float_as_uint(hitN.x); s_item[4][slot] = __float_as_uint(hitN.y); s_item[5][slot] = __float_as_uint(hitN.z);
                s_item[6][slot] = __float_as_uint(hitUV.x); s_item[7][slot] = __float_as_uint(hitUV.y);
                s_item[8][slot] = (uint32_t)hitMat; s_item[9][slot] = backface ? 1u : 0u;
                s_item[10][slot] = __float_as_uint(rayDir.x); s_item[11][slot] = __float_as_uint(rayDir.y); s_item[12][slot] = __float_as_uint(rayDir.z);
                s_item[13][slot] = __float_as_uint(L.x); s_item[14][slot] = __float_as_uint(L.y); s_item[15][slot] = __float_as_uint(L.z);
                s_item[16][slot] = __float_as_uint(T.x); s_item[17][slot] = __float_as_uint(T.y); s_item[18][slot] = __float_as_uint(T.z);
                s_item[19][slot] = seed;
            }
            __syncthreads();
            const uint32_t k = threadIdx.x;
            if (k < total) {                                          // the item in MY slot: someone's path of (mostly) my wave's BSDF type
                SurfHit h;
                h.P = mk3(__uint_as_float(s_item[0][k]), __uint_as_float(s_item[1][k]), __uint_as_float(s_item[2][k]));
                h.N = mk3(__uint_as_float(s_item[3][k]), __uint_as_float(s_item[4][k]), __uint_as_float(s_item[5][k]));
                h.uv = mk2(__uint_as_float(s_item[6][k]), __uint_as_float(s_item[7][k]));
                const int mid = (int)s_item[8][k]; const bool bf = s_item[9][k] != 0u;
                const f3 dIn = mk3(__uint_as_float(s_item[10][k]), __uint_as_float(s_item[11][k]), __uint_as_float(s_item[12][k]));
                const f3 Lk = mk3(__uint_as_float(s_item[13][k]), __uint_as_float(s_item[14][k]), __uint_as_float(s_item[15][k]));
                const f3 Tk = mk3(__uint_as_float(s_item[16][k]), __uint_as_float(s_item[17][k]), __uint_as_float(s_item[18][k]));
                uint32_t sk = s_item[19][k];
                const MatStep r_ = material_step<FUSE>(sc, h, sc.materials[mid], bf, dIn, Lk, Tk, &sk);
                s_item[0][k] = __float_as_uint(r_.bsdfNEE.x); s_item[1][k] = __float_as_uint(r_.bsdfNEE.y); s_item[2][k] = __float_as_uint(r_.bsdfNEE.z);
                s_item[3][k] = __float_as_uint(r_.bsdfPdfW); s_item[4][k] = r_.singular;
                s_item[5][k] = __float_as_uint(r_.newT.x); s_item[6][k] = __float_as_uint(r_.newT.y); s_item[7][k] = __float_as_uint(r_.newT.z);
                s_item[8][k] = __float_as_uint(r_.orig.x); s_item[9][k] = __float_as_uint(r_.orig.y); s_item[10][k] = __float_as_uint(r_.orig.z);
                s_item[11][k] = __float_as_uint(r_.pdfW);
                s_item[12][k] = __float_as_uint(r_.newDir.x); s_item[13][k] = __float_as_uint(r_.newDir.y); s_item[14][k] = __float_as_uint(r_.newDir.z);
                s_item[15][k] = sk;
            }
            __syncthreads();
            if (work) {
                o.bsdfNEE = mk3(__uint_as_float(s_item[0][slot]), __uint_as_float(s_item[1][slot]), __uint_as_float(s_item[2][slot]));
                o.bsdfPdfW = __uint_as_float(s_item[3][slot]); o.singular = s_item[4][slot];
                o.newT = mk3(__uint_as_float(s_item[5][slot]), __uint_as_float(s_item[6][slot]), __uint_as_float(s_item[7][slot]));
                o.orig = mk3(__uint_as_float(s_item[8][slot]), __uint_as_float(s_item[9][slot]), __uint_as_float(s_item[10][slot]));
                o.pdfW = __uint_as_float(s_item[11][slot]);
                o.newDir = mk3(__uint_as_float(s_item[12][slot]), __uint_as_float(s_item[13][slot]), __uint_as_float(s_item[14][slot]));
                seed = s_item[15][slot];
            }
        }
        if (!terminate && !regrouped) {
            if (inlined) {
                // the material kernel's step for this path (material.hip: material_body), fed from registers.  The "stored light
                // direction" is this iteration's sample when NEE stored one, else whatever the record holds (the material kernels
                // evaluate toward it regardless, src/wf_mat_diffuse.cl:34-37; logic consumes the result only behind an unblocked ray)
                const f3 L = haveL ? Lnee : (FULL ? Lold : ld3(rd4(st.at(S_SHD, gid))));
                if (FUSE == USE_ALL) {          // (a RAW path stored its hit point and uv record above: same thread, same addresses, program order)
                    // the all-types kernel is register-bound (106 VGPRs, 4 waves/SIMD): re-reading the hit point, uv, material and ray
                    // direction here (L1/L2-hot, this thread loaded them above) instead of keeping them live across the NEE code brings
                    // it to 87 VGPRs and 5 waves (-3 % kernel time); the diffuse-only kernel stays at 5 waves either way
                    const float4 hp2 = rd4t(st.at(S_HITP, gid)), huv2 = rd4t(st.at(S_HITUV, gid)), d42 = rd4t(st.at(S_DIR, gid));
                    SurfHit h; h.P = ld3(hp2); h.N = hitN; h.uv = mk2(huv2.x, huv2.y);
                    o = material_step<FUSE>(sc, h, sc.materials[__float_as_int(huv2.w)], backface, ld3(d42), L, T, &seed);
                } else {
                    SurfHit h; h.P = hitP; h.N = hitN; h.uv = hitUV;
                    o = material_step<FUSE>(sc, h, mat, backface, rayDir, L, T, &seed);
                }
                if (!FULL) {
                    wr4(st.at(S_LBSDF, gid), mk4(o.bsdfNEE, o.bsdfPdfW));
                    wr4(st.at(S_LT, gid), mk4u(T, o.singular));
                    wr4(st.at(S_THR, gid), mk4u(o.newT, seed));
                    wr4(st.at(S_ORIG, gid), mk4(o.orig, o.pdfW));
                    wr4(st.at(S_DIR, gid), mk4u(o.newDir, len));           // pathLen without the FLX_FRESH flag, as the material kernels leave it
                }
            } else if (!FULL) {
                wr4(st.at(S_THR, gid), mk4u(T, seed));
                if (RAW && isRaw) wr4(st.at(S_DIR, gid), mk4u(rayDir, lenBits));      // pathLen + 1 for the material kernel that serves this path
            }
        }
        if (FULL) {
            // ---- store phase B (all lanes): what the material step leaves.  C1: its results.  T: throughput + seed (genRays reads the seed), lastT as
            // loaded (genRays does not reset it); origin, direction and lastBsdf are dead (genRays writes the first two, its flag covers the third).
            // C2: throughput + seed, the old direction with pathLen + 1 (the material kernel reads both); it overwrites lastBsdf, lastT, origin.
            // C0 (no queue: nobody touches the path again): everything as loaded.
            const bool c1 = !terminate && inlined, c0 = !terminate && !inlined && ml == 0u;
            float4 lbs = mk4(o.bsdfNEE, o.bsdfPdfW), lt4 = mk4u(T, o.singular), or4 = mk4(o.orig, o.pdfW), dr4 = mk4u(o.newDir, len);
            if (!c1) {
#ifdef FLX_LAB_LOGIC_NOOLD
                if (c0) lt4 = rd4(st.at(S_LT, gid));
#else
                if (terminate || c0) lt4 = rd4(st.at(S_LT, gid));
#endif
                if (!terminate) { dr4 = rd4t(st.at(S_DIR, gid)); dr4.w = __uint_as_float(lenBits); }     // (this thread loaded the line at the top)
                if (c0) { lbs = rd4(st.at(S_LBSDF, gid)); or4 = rd4t(st.at(S_ORIG, gid)); }
            }
            float4 thr4 = mk4u(c1 ? o.newT : T, seed);
            if (regen && terminate) {                                 // genRays for this lane (misc.hip: k_raygen), index in the raygen queue = regenRank
                uint32_t sd = seed; f3 ro, rd;
                camera_ray(fr, p, regenLocal, &sd, &ro, &rd);
                or4 = mk4(ro, 1.0f);                                  // lastPdfW = 1
                dr4 = mk4u(rd, FLX_FRESH | 0u);                       // pathLen = 0 + "no material kernel since regeneration"
                thr4 = mk4u(mk3(1.0f), sd);
                st.firstDiffuse[gid] = 0u;
                if (aux.appendExt) qs.q[FLX_Q_EXTENSION][ext_len(qs) + regenRank] = gid;
            }
            if (prep && terminate) {                                  // the seed-only half of genRays (flx_shading.h: camera_ray in parts)
                uint32_t sd = seed;
                const float jx = rand01(&sd), jy = rand01(&sd);
                const uint32_t sd2 = sd;
                const f3 ro = camera_lens_origin(p, &sd);
                or4 = mk4(ro, 1.0f);                                  // lastPdfW = 1
                dr4 = make_float4(jx, jy, __uint_as_float(sd2), __uint_as_float(FLX_FRESH | 0u));      // for k_raygen: jitter + the seed behind it (it redoes the lens draws)
                thr4 = mk4u(mk3(1.0f), sd);                           // T = 1, the seed genRays leaves
            }
            // shadowRayBlocked: every lane, one store (was a partial one of the few lanes with the area light behind the surface): 1 for those, genRays' 1 for a
            // lane regenerated or prepared here (lastLightPickProb = 1 went out with store phase A), else what the lane loaded
            st.blocked[gid] = (neeBlocked || ((regen || prep) && terminate)) ? 1u : blockedOld;
            wr4(st.at(S_LBSDF, gid), lbs);
            wr4(st.at(S_LT, gid), lt4);
            wr4(st.at(S_THR, gid), thr4);
            wr4(st.at(S_ORIG, gid), or4);
            wr4(st.at(S_DIR, gid), dr4);
            if (terminate) splat();
        }
    }

    if (gid < st.numTasks) aux.member[gid] = (uint8_t)member;

    // per-block counts of the 7 lists: ballot + popcount per wave, 4 waves per block
    __shared__ uint32_t s_cnt[NUM_LISTS][LOGIC_BLOCK / 64];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t ml = member >> 2;
    uint64_t b0 = __ballot(member & 1u), b1 = __ballot(member & 2u);
    uint64_t m1 = __ballot(ml == 1u), m2 = __ballot(ml == 2u), m3 = __ballot(ml == 3u), m4 = __ballot(ml == 4u), m5 = __ballot(ml == 5u);
    if ((threadIdx.x & 63u) == 0u) {
        s_cnt[0][wave] = (uint32_t)__popcll(b0); s_cnt[1][wave] = (uint32_t)__popcll(b1);
        s_cnt[2][wave] = (uint32_t)__popcll(m1); s_cnt[3][wave] = (uint32_t)__popcll(m2);
        s_cnt[4][wave] = (uint32_t)__popcll(m3); s_cnt[5][wave] = (uint32_t)__popcll(m4); s_cnt[6][wave] = (uint32_t)__popcll(m5);
    }
    __syncthreads();
    if (threadIdx.x < NUM_LISTS) {
        uint32_t s = 0;
        for (int w = 0; w < LOGIC_BLOCK / 64; w++) s += s_cnt[threadIdx.x][w];
        aux.blockCounts[threadIdx.x * aux.stride + blockIdx.x] = s;
    }
}

// one block per list: exclusive scan of the list's per-block counts; the total is added to its queue counter.
// Every thread owns a contiguous run of counts (read as uint4s), the 16 waves scan their threads' sums with shuffles, wave 0 scans the
// 16 wave totals: two barriers in all (the first version's Hillis-Steele over 1024 partial sums took 20 barriers and 21 us, during
// which nothing else runs -- every later kernel of the iteration waits for the queues).
__global__ __launch_bounds__(1024) void k_queue_scan(LogicAux aux, uint32_t *counters)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t listToCounter[NUM_LISTS] = {FLX_Q_RAYGEN, FLX_Q_SHADOW, FLX_Q_DIFFUSE, FLX_Q_GLOSSY, FLX_Q_GGX_REFL, FLX_Q_GGX_REFR, FLX_Q_DELTA};
    const uint32_t nb = aux.numBlocks;
    const uint32_t per = ((nb + 1023u) / 1024u + 3u) & ~3u;          // counts per thread, a multiple of 4 (the arrays are padded to it)
    const int l = blockIdx.x;
    const uint32_t *cnt = aux.blockCounts + (size_t)l * aux.stride;
    uint32_t *off = aux.blockOffsets + (size_t)l * aux.stride;
    const uint32_t base = counters[listToCounter[l]];
    const uint32_t lo = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t i = 0; i < per; i += 4) {
        if (lo + i < nb) {                                            // (the pad beyond nb is zero-filled once at allocation and never written)
            const uint4 v = *reinterpret_cast<const uint4 *>(cnt + lo + i);
            s += v.x + v.y + v.z + v.w;
        }
    }
    // inclusive scan of s over the wave
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = s;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += v; }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    if (wave == 0u) {
        uint32_t w = lane < 16u ? s_wave[lane] : 0u, wi = w;
        for (int d = 1; d < 16; d <<= 1) { const uint32_t v = __shfl_up(wi, d, 64); if (lane >= (uint32_t)d) wi += v; }
        if (lane < 16u) s_wave[lane] = wi - w;                        // exclusive prefix of the wave totals
        if (lane == 15u) counters[listToCounter[l]] = base + wi;
    }
    __syncthreads();
    uint32_t run = base + s_wave[wave] + inc - s;
    for (uint32_t i = 0; i < per; i += 4) {
        if (lo + i < nb) {
            const uint4 v = *reinterpret_cast<const uint4 *>(cnt + lo + i);
            uint4 o;
            o.x = run; run += v.x; o.y = run; run += v.y; o.z = run; run += v.z; o.w = run; run += v.w;
            *reinterpret_cast<uint4 *>(off + lo + i) = o;
        }
    }
}

// stable scatter: rank within block by wave ballots, block base from the scan.
// fuse != 0: the scatter also writes the extension-queue entries of EVERY continuing path -- inlined BSDF type or not; the material
// kernel that serves the other types afterwards does not append -- as one list in path-id order behind (or in front of) the regenerated
// paths, where the separate material kernels append one segment per material queue.  The extension queue is a set (the reference fills
// it with atomic_inc in whatever order the work-items arrive, src/utils.cl:328-358); its order only decides which lane traces which ray,
// and neighbouring path ids are neighbouring pixels' paths: on the same 4 M rays k_extend4 takes 0.730 ms in path-id order against
// 0.800 ms in per-material segments on the conference scene (three BSDF types of similar weight), 0.912 against 0.933 ms on the kitchen,
// 1.30 / 1.10 ms shuffled (scripts/exp_octant.py).  Needs material queues that were empty before this `logic` (the host checks).
__global__ __launch_bounds__(LOGIC_BLOCK) void k_queue_scatter(Queues qs, LogicAux aux, uint32_t numTasks, int fuse, uint32_t raygenFirst, uint32_t byPathId)
{
    const uint32_t gid = blockIdx.x * LOGIC_BLOCK + threadIdx.x;
    const uint32_t member = gid < numTasks ? aux.member[gid] : 0u;
    const uint32_t ml = member >> 2;
    const uint32_t wave = threadIdx.x >> 6;
    __shared__ uint32_t s_cnt[NUM_LISTS][LOGIC_BLOCK / 64];
    __shared__ uint32_t s_off[NUM_LISTS];
    if (threadIdx.x < NUM_LISTS) s_off[threadIdx.x] = aux.blockOffsets[(size_t)threadIdx.x * aux.stride + blockIdx.x];   // in flight beside the member bytes
    uint64_t bal[NUM_LISTS];
    bal[0] = __ballot(member & 1u); bal[1] = __ballot(member & 2u);
    bal[2] = __ballot(ml == 1u); bal[3] = __ballot(ml == 2u); bal[4] = __ballot(ml == 3u); bal[5] = __ballot(ml == 4u); bal[6] = __ballot(ml == 5u);
    if ((threadIdx.x & 63u) == 0u)
        for (int l = 0; l < NUM_LISTS; l++) s_cnt[l][wave] = (uint32_t)__popcll(bal[l]);
    __syncthreads();
    uint32_t *const outq[NUM_LISTS] = {qs.q[FLX_Q_RAYGEN], qs.q[FLX_Q_SHADOW], qs.q[FLX_Q_DIFFUSE], qs.q[FLX_Q_GLOSSY], qs.q[FLX_Q_GGX_REFL], qs.q[FLX_Q_GGX_REFR], qs.q[FLX_Q_DELTA]};
    #pragma unroll
    for (int l = 0; l < NUM_LISTS; l++) {
        bool in = l == 0 ? (member & 1u) : l == 1 ? ((member & 2u) != 0u) : (ml == (uint32_t)(l - 1));
        if (in) {
            uint32_t r = s_off[l];
            for (uint32_t w = 0; w < wave; w++) r += s_cnt[l][w];
            r += mbcnt(bal[l]);
            outq[l][r] = gid;
            if (fuse != 0 && byPathId == 0u && l >= 2 && fuse_inlines_list(fuse, (uint32_t)(l - 1))) {
                // option ext_order 0: the separate kernels' order -- one segment per material queue; the types that are not inlined
                // are appended by their material kernel
                uint32_t base = ext_len(qs) + (raygenFirst ? qs.counters[FLX_Q_RAYGEN] : 0u);
                for (int q = FLX_Q_DIFFUSE; q < FLX_Q_DIFFUSE + (l - 2); q++) base += qs.counters[q];
                qs.q[FLX_Q_EXTENSION][base + r] = gid;
            }
        }
    }
    if (fuse != 0 && byPathId == 1u && ml != 0u) {
        // the continuing paths of ALL material lists, in path-id order (see the comment above the kernel)
        uint32_t r = s_off[2] + s_off[3] + s_off[4] + s_off[5] + s_off[6];
        for (uint32_t w = 0; w < wave; w++) r += s_cnt[2][w] + s_cnt[3][w] + s_cnt[4][w] + s_cnt[5][w] + s_cnt[6][w];
        r += mbcnt(bal[2] | bal[3] | bal[4] | bal[5] | bal[6]);
        qs.q[FLX_Q_EXTENSION][ext_len(qs) + (raygenFirst ? qs.counters[FLX_Q_RAYGEN] : 0u) + r] = gid;
    }
    if (fuse != 0 && byPathId == 2u && ((member & 1u) != 0u || ml != 0u)) {
        // ext_order 2: EVERY path that will be traced -- the regenerated ones (genRays follows in this chain and does not append: k_raygen,
        // appendExt 0) merged with the continuing ones -- as ONE list in path-id order.  In the steady state every path traces an extension
        // ray, so the queue is the identity permutation: the traversal kernel's path-state accesses are fully coalesced (round 3 measured
        // -2 % / -2 % / -6 % of the closest-hit kernel for that order on host-reordered queues, scripts/exp_ray_order.py).  The same SET as the
        // separate kernels' [regenerated | material segments] (the reference's order is whatever its atomic_inc produces).
        uint32_t r = s_off[0] + s_off[2] + s_off[3] + s_off[4] + s_off[5] + s_off[6];
        for (uint32_t w = 0; w < wave; w++) r += s_cnt[0][w] + s_cnt[2][w] + s_cnt[3][w] + s_cnt[4][w] + s_cnt[5][w] + s_cnt[6][w];
        r += mbcnt(bal[0] | bal[2] | bal[3] | bal[4] | bal[5] | bal[6]);
        qs.q[FLX_Q_EXTENSION][ext_len(qs) + r] = gid;
    }
}

// elements per list in the block-count / block-offset arrays (api.hip allocates NUM_LISTS x this, zero-filled)
uint32_t logic_aux_stride(uint32_t numTasks)
{
    const uint32_t blocks = (numTasks + LOGIC_BLOCK - 1) / LOGIC_BLOCK;
    const uint32_t per = ((blocks + 1023u) / 1024u + 3u) & ~3u;
    return per * 1024u;
}

// queues (bit q) whose paths the fused pass takes through their material step
uint32_t fused_queue_mask(int fuse)
{
    uint32_t m = 0;
    for (uint32_t ml = 1; ml <= 5; ml++) if (fuse_inlines_list(fuse, ml)) m |= 1u << (FLX_Q_DIFFUSE + ml - 1);
    return m;
}

// fuse: 0 = the plain logic kernel | USE_DIFFUSE | USE_ALL  (diffuse + glossy was measured too: never the best of the three)
void launch_logic(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, const Frame &fr, const flx_render_params &p,
                  uint8_t *member, uint32_t *blockCounts, uint32_t *blockOffsets, int firstIteration, int fuse, int raygenFirst, int extByPathId, int raw,
                  unsigned long long *lookback, uint32_t epoch, int regen, int regenAppendExt, uint32_t *error, int regroup)
{
    // the reference launches ceil32(NUM_TASKS) work-items (src/clcontext.cpp:792); here ceil256
    uint32_t blocks = (st.numTasks + LOGIC_BLOCK - 1) / LOGIC_BLOCK;
    LogicAux aux{member, blockCounts, blockOffsets, blocks, logic_aux_stride(st.numTasks), lookback, epoch, (uint32_t)(raw ? regen : 0), (uint32_t)regenAppendExt, error};
    // the BSDF-uniform material step needs every thread of every block inside the pass (block barriers): whole blocks of paths, no `first` cut-off
    const bool rg = LOGIC_REGROUP && raw && regroup && fuse == USE_ALL && !firstIteration && st.numTasks % LOGIC_BLOCK == 0u;
    const dim3 g(blocks), b(LOGIC_BLOCK);
    switch (fuse) {
    case USE_DIFFUSE:
        if (raw) hipLaunchKernelGGL((k_logic<USE_DIFFUSE, true>), g, b, 0, s, st, sc, fr, p, aux, qs, (uint32_t)firstIteration);
        else hipLaunchKernelGGL((k_logic<USE_DIFFUSE, false>), g, b, 0, s, st, sc, fr, p, aux, qs, (uint32_t)firstIteration);
        break;
    case USE_ALL:
        if (rg) hipLaunchKernelGGL((k_logic<USE_ALL, true, true>), g, b, 0, s, st, sc, fr, p, aux, qs, (uint32_t)firstIteration);
        else if (raw) hipLaunchKernelGGL((k_logic<USE_ALL, true>), g, b, 0, s, st, sc, fr, p, aux, qs, (uint32_t)firstIteration);
        else hipLaunchKernelGGL((k_logic<USE_ALL, false>), g, b, 0, s, st, sc, fr, p, aux, qs, (uint32_t)firstIteration);
        break;
    default: fuse = 0; hipLaunchKernelGGL(k_logic<0>, g, b, 0, s, st, sc, fr, p, aux, qs, (uint32_t)firstIteration); break;
    }
    hipLaunchKernelGGL(k_queue_scan, dim3(NUM_LISTS), dim3(1024), 0, s, aux, qs.counters);
    hipLaunchKernelGGL(k_queue_scatter, g, b, 0, s, qs, aux, st.numTasks, fuse, (uint32_t)raygenFirst, (uint32_t)extByPathId);
}
// does this build regenerate terminating paths inside the fused RAW pass (LOGIC_FULL_STORES)?  (api.hip asks before it skips the genRays launch)
int logic_can_regenerate() { return LOGIC_FULL_STORES != 0 ? 1 : 0; }
uint32_t logic_lookback_words(uint32_t numTasks) { return (numTasks + 63u) / 64u; }

} // namespace flxd
