// api.hip -- the C ABI of libfluctus_hip.so (include/fluctus_hip.h): context, uploads with the
// CDNA4 re-layout of the BVH, asynchronous kernel sequencing on one HIP stream, measurement hooks.
#include "flx_device.h"
#include "flx_wide.h"
#include "flx_trace.h"
#include "flx_trace4.h"
#include "../../include/fluctus_hip.h"
#include <string>
#include <vector>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <utility>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl.so.1 is bound with dlopen at the first group call

namespace flxd {
void launch_extend(hipStream_t, const State &, const Queues &, const Scene &, const flx_render_params &, uint32_t *, unsigned long long *, int);
void launch_shadow(hipStream_t, const State &, const Queues &, const Scene &, const flx_render_params &, uint32_t *, unsigned long long *, int);
void launch_extend4(hipStream_t, const State &, const Queues &, const Scene &, const flx_render_params &, uint32_t *, unsigned long long *);
void launch_shadow4(hipStream_t, const State &, const Queues &, const Scene &, const flx_render_params &, uint32_t *, unsigned long long *);
void launch_shadow4_split(hipStream_t, const State &, const Queues &, const Scene &, const flx_render_params &, uint32_t *, uint32_t *, uint32_t, uint4 *, uint32_t, uint4 *, uint32_t, int, int, uint32_t);
uint32_t shadow_split_lists(); uint32_t shadow_split_count_words();
void launch_extend4r(hipStream_t, const State &, const Queues &, const Scene &, const flx_render_params &, uint32_t *, uint32_t, int, uint32_t *);
void launch_shadow4r(hipStream_t, const State &, const Queues &, const Scene &, const flx_render_params &, uint32_t *, uint32_t, int, uint32_t *);
void launch_logic(hipStream_t, const State &, const Queues &, const Scene &, const Frame &, const flx_render_params &, uint8_t *, uint32_t *, uint32_t *, int, int, int, int, int,
                  unsigned long long *, uint32_t, int, int, uint32_t *, int);
void launch_env_nee_table(hipStream_t, const Scene &, float4 *, uint32_t);
int logic_can_regenerate();
uint32_t logic_lookback_words(uint32_t numTasks);
void launch_materialise(hipStream_t, const State &, const Scene &, const flx_render_params &, uint32_t);
void launch_materials(hipStream_t, const State &, const Queues &, const Scene &, uint32_t);
void launch_materials_after_fused(hipStream_t, const State &, const Queues &, const Scene &, uint32_t, int);
uint32_t fused_queue_mask(int);
uint32_t logic_aux_stride(uint32_t);
void launch_reset(hipStream_t, const State &, const Queues &, const Frame &, const flx_render_params &);
void launch_raygen(hipStream_t, const State &, const Queues &, const Frame &, const flx_render_params &, int, int);
void launch_postprocess(hipStream_t, const Frame &, const flx_render_params &);
void launch_state_export(hipStream_t, const State &, float *, float);
void launch_state_import(hipStream_t, const State &, const float *);
void launch_math_probe(hipStream_t, int, const float *, const float *, uint32_t, uint32_t *);
void launch_mk_reset(hipStream_t, const State &, const Frame &, const flx_render_params &);
void launch_mk_raygen(hipStream_t, const State &, const flx_render_params &);
void launch_mk_next_vertex(hipStream_t, const State &, const Scene &, const Frame &, const flx_render_params &, uint32_t *, uint32_t *);
void launch_mk_sample_bsdf(hipStream_t, const State &, const Scene &, const Frame &, const flx_render_params &, uint32_t *, uint32_t *);
void launch_mk_splat(hipStream_t, const State &, const Frame &, const flx_render_params &, uint32_t *, int);
void launch_end_iteration(hipStream_t, uint32_t *, unsigned long long *, uint32_t *, uint32_t, uint32_t, uint32_t *);
void launch_bump_extension(hipStream_t, uint32_t *, uint32_t);
void launch_deinterleave(hipStream_t, const float *, float *, uint32_t, uint32_t, uint32_t);
}

using namespace flxd;
#ifdef FLX_LAB_RSTATS
namespace flxd { extern unsigned long long *g_lab_rstats; }
#endif

static thread_local std::string g_create_error;

struct PendingCounters { void *user; int slot; };
struct PendingEvent { int kernel; hipEvent_t a, b; };

struct flx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;              // the shadow kernel runs here, concurrently with the extension kernel
    hipEvent_t evPreExt = nullptr, evShadow = nullptr, evPostLogic = nullptr;
#ifdef FLX_LAB_NOJOIN
    // lab build only (-DFLX_LAB_NOJOIN; RESULTS INVALID, timing only): the ceiling of taking the any-hit kernel's tail off the step's critical
    // path -- the main stream does not join the shadow stream after flx_wf_shadow; the next-but-one `logic` waits for it instead (the throttle a
    // deferred NEE-consume kernel would impose).  The kernel reads a snapshot of the queue counters (the live ones are cleared under it).
    hipEvent_t evLab[2] = {nullptr, nullptr}; uint32_t labIter = 0; uint32_t *labCounters = nullptr;
#endif
    int phase = 0;                              // the call-sequence state machine (enum Phase below): ONE explicit state instead of deferral / chain booleans
    int overlap = 2;                            // 0 serial | 1 shadow || extension | 2 shadow starts right after logic (the EFFECTIVE schedule)
    int overlapOpt = -1;                        // option "overlap": -1 = the default (pickSchedule), else as set
    uint32_t *spill2 = nullptr;
    // logic + material kernels as one pass (logic.hip: k_logic<FUSED>).  flx_wf_logic is DEFERRED while `fuse` is on: it is
    // launched by the next call -- fused with the material kernels when that call is flx_wf_materials (a flx_wf_raygen between
    // the two is deferred along and launched right after), as the plain kernel when it is anything else.  Every entry point
    // takes one step of the state machine first (enter()), so no call ever observes a state the separate kernels would not have produced.
    int fuse = 1;
    int extOrder = 0;                           // fused pass: extension queue lists the continuing paths 1 by path id | 2 merged with the regenerated ones by path id | 0 one segment per material queue; chosen at flx_upload_scene
    int fuseSet = 1;                            // BSDF types the fused pass inlines (logic.hip): 1 diffuse | 31 all six; chosen at flx_upload_scene
    int regroup = 0, regroupAuto = 0, regroupOpt = -1;           // all-types fused pass with its material step sorted by BSDF type inside each block (logic.hip: LOGIC_REGROUP): the EFFECTIVE choice (flx_upload_scene) | option "regroup": -1 = that choice, 0 / 1 as set
    int pendFirst = 0;                          // the deferred flx_wf_logic's `first` (phases PH_DEFER_*)
    bool matQueuesEmpty = false;                // the five material counters are known to be zero (cleared, nothing appended since)
    bool raygenQueueEmpty = false;              // ... and the raygen counter (ext_order 2 ranks the regenerated paths from zero: extOrderFor)
    uint32_t numTasks = 0;
    std::string err;
    State st {};
    Queues qs {};
    Scene sc {};
    Frame fr {};
    flx_render_params params {};
    bool haveParams = false;
    uint32_t hostPixelIdx = 0;
    // logic aux
    uint8_t *member = nullptr; uint32_t *blockCounts = nullptr, *blockOffsets = nullptr;
    // in-kernel regeneration of the fused RAW pass (logic.hip: REGEN): look-back status words (one per wave, epoch-stamped: never reset), launch counter,
    // device error flag (a look-back that gave up), option "regen" (1: on where the pass allows it), and whether the LAST fused pass regenerated its
    // terminating paths itself -- then the genRays of the chain is not launched (flx_wf_materials)
    unsigned long long *lookback = nullptr; uint32_t logicEpoch = 0; uint32_t *logicError = nullptr; int regenOpt = 0; bool regenDone = false; bool regenUsed = false; int prepOpt = 1; bool prepDone = false;      // (off by default: profiles/r05_regen_ab.txt -- the look-back costs more than genRays)
    // trace aux
    uint32_t *spill = nullptr;
    unsigned long long *stats = nullptr;   // device, 16 counters
    unsigned long long *totals = nullptr;  // device, 8 running queue-length totals
    uint32_t *mkStats = nullptr;           // device RenderStats of the microkernel integrator (4 x u32)
    uint32_t *pinnedMk = nullptr; std::vector<std::pair<void *, int>> pendingMk; int nextMkSlot = 0;
    bool statsOn = false;
    int xcdRemap = 0;           // 1: each XCD gets a contiguous eighth of the queue (measured slower: round-robin keeps all XCDs on the same part of the tree)
    // which tree each traversal kernel walks: 2 = the reference's binary tree in the reference's visit order (bit-exact closest hit),
    // 4 = the 4-wide quantised tree over the same leaves (flx_wide.h): any-hit bit-exact by construction, closest hit exact up to
    // visit-order ties
    int shadowTree = 4, extendTree = 4;
    // persistent waves with lane refill for the 4-wide kernels (trace4r.hip): 0 = thread-per-ray kernels, n > 0 = refill when n lanes are idle
    // (closest hit: on by default -- refillMin 16, waitMax 32: kitchen 0.82 -> 0.61 ms per 4 M rays; any hit: off by default, pickSchedule)
    int refillExt = 16 | (32 << 8), refillShadow = 0;
    int refillShadowOpt = -1;                   // option "refill_shadow": -1 = the default (off), else as set
    // tail splitting of the thread-per-ray any-hit kernel (trace4.hip: k_shadow4s): budget of the pass over the queue | budget of a second pass << 8
    // (0 = the second pass finishes every ray); 0 = off (k_shadow4).  Continuation records: 64 B each, in sub-lists of splitCapA / splitCapB slots (numTasks / 2 and / 8 in all).
    int shadowSplit = 0;
    uint32_t splitParity = 0;                   // counter set of the next split launch (trace4.hip: launch_shadow4_split)
    uint32_t splitLimit = 0;                    // test hook (option shadow_split_limit): use only this many slots per sub-list (0 = all), to reach the full-list path
    uint32_t *splitCounts = nullptr; uint4 *splitRecA = nullptr, *splitRecB = nullptr; uint32_t splitCapA = 0, splitCapB = 0;
    // The persistent-wave extension kernel leaves RAW hit records (flx_trace.h): true from flx_wf_extend until they are committed -- by the
    // fused logic pass of the next iteration (the steady state: nothing else touches hit records between the extension kernel and logic),
    // or by k_materialise as soon as an entry point that could observe a hit record runs (transition(): commitRaw).
    bool rawHits = false;
    bool cursorDirty[2] = {false, false};       // block cursors of the persistent kernels (closest hit, any hit) used since they were last zeroed

    uint32_t wideInfo[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // flx_scene_info
    bool wideOK = false;        // the uploaded scene has a wide tree whose exactness conditions hold (nested boxes)
    uint32_t spillLevels = 0;   // levels per lane in each spill buffer (sized at upload from the tree's depth)
    int eagerBump = 0;          // A/B: bump the extension counter right after raygen / materials (option eager_bump)
    int denoiser = 0;           // USE_OPTIX_DENOISER of the reference: accumulate the denoiser feature buffers
    std::vector<void *> aovAllocs;
    int nodeLayout = 1;         // 1 = sibling-pair record numbering (see flx_upload_scene), 0 = DFS
    int numCUs = 256;
    // multi-GPU group (flx_group_*): RCCL communicator of this rank, root-side staging
    ncclComm_t comm = nullptr;
    bool commShared = false;                    // same-device local group: no communicator, device copies instead
    float *gatherStage = nullptr, *gatherFull = nullptr; size_t gatherStageFloats = 0, gatherFullFloats = 0;
    std::vector<void *> gatherAllocs;
    // owned device allocations
    std::vector<void *> sceneAllocs, envAllocs, frameAllocs, fixedAllocs, spillAllocs;
    // async counter read-back
    flx_queue_counters *pinned = nullptr; int pinnedSlots = 64, nextSlot = 0;
    uint32_t *pinnedIdx = nullptr; int nextIdxSlot = 0;
    std::vector<PendingCounters> pending;
    // profiling
    int profile = 0;              // 0 off | 1 time every kernel | 2 the traversal kernels + span | 3 the extension kernel only | 4 extension + logic + shadow
    hipEvent_t spanStart = nullptr;             // pending FLX_K_TRACE_SPAN start (recorded in flx_wf_extend)
    std::vector<PendingEvent> events;
    std::vector<hipEvent_t> eventPool;
    double kMs[FLX_K_COUNT] = {0}; uint64_t kLaunches[FLX_K_COUNT] = {0};
};

#define HIPCHK(c, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return 1; } } while (0)
#define NEED(c, cond, msg) do { if (!(cond)) { (c)->err = msg; return 1; } } while (0)

template <class T> static int dalloc(flx_ctx *c, std::vector<void *> &own, T **p, size_t count)
{
    void *d = nullptr;
    HIPCHK(c, hipMalloc(&d, (count ? count : 1) * sizeof(T)));
    own.push_back(d);
    *p = (T *)d;
    return 0;
}
static void freeAll(std::vector<void *> &v) { for (void *p : v) (void)hipFree(p); v.clear(); }

static hipEvent_t getEvent(flx_ctx *c)
{
    if (!c->eventPool.empty()) { hipEvent_t e = c->eventPool.back(); c->eventPool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct ScopedTimer {
    flx_ctx *c; int k; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    bool on;
    // profile level 1 = every kernel, 2 = the two traversal kernels + their span, 3 = the extension kernel only, 4 = the three kernels the bench line
    // prices against a roof: extension, (fused) logic, shadow  (each event pair costs a few us of stream time)
    ScopedTimer(flx_ctx *c_, int k_, hipStream_t s_ = nullptr) : c(c_), k(k_), s(s_ ? s_ : c_->stream)
    {
        on = c->profile == 1 || (c->profile == 2 && (k == FLX_K_EXTEND || k == FLX_K_SHADOW)) || (c->profile == 3 && k == FLX_K_EXTEND) ||
             (c->profile == 4 && (k == FLX_K_EXTEND || k == FLX_K_SHADOW || k == FLX_K_LOGIC || k == FLX_K_LOGIC_FUSED));
        if (on) { a = getEvent(c); b = getEvent(c); (void)hipEventRecord(a, s); }
    }
    ~ScopedTimer() { if (on) { (void)hipEventRecord(b, s); c->events.push_back({k, a, b}); } }
};

static uint32_t localPixels(const flx_ctx *c)
{
    uint32_t npix = c->params.width * c->params.height;
    if (npix <= c->fr.rank) return 1;
    return (npix - c->fr.rank + c->fr.nranks - 1) / c->fr.nranks;
}

// denoiser feature buffers (4 x float4 per local pixel) exist only while the option is on
static int allocAov(flx_ctx *c)
{
    freeAll(c->aovAllocs);
    c->fr.aovAlbedo = c->fr.aovNormal = c->fr.aovAlbedoOut = c->fr.aovNormalOut = nullptr;
    if (!c->denoiser || !c->fr.localPixels) return 0;
    const size_t n = (size_t)c->fr.localPixels * 4;
    float *buf[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; i++) {
        HIPCHK(c, dalloc(c, c->aovAllocs, &buf[i], n) ? hipErrorOutOfMemory : hipSuccess);
        HIPCHK(c, hipMemsetAsync(buf[i], 0, n * 4, c->stream));
    }
    c->fr.aovAlbedo = buf[0]; c->fr.aovNormal = buf[1]; c->fr.aovAlbedoOut = buf[2]; c->fr.aovNormalOut = buf[3];
    return 0;
}

static int allocFrame(flx_ctx *c)
{
    uint32_t lp = localPixels(c);
    if (lp == c->fr.localPixels && c->fr.pixels) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    freeAll(c->frameAllocs);
    HIPCHK(c, dalloc(c, c->frameAllocs, &c->fr.pixels, (size_t)lp * 4) ? hipErrorOutOfMemory : hipSuccess);
    HIPCHK(c, dalloc(c, c->frameAllocs, &c->fr.preview, (size_t)lp * 4) ? hipErrorOutOfMemory : hipSuccess);
    HIPCHK(c, hipMemsetAsync(c->fr.pixels, 0, (size_t)lp * 16, c->stream));
    HIPCHK(c, hipMemsetAsync(c->fr.preview, 0, (size_t)lp * 16, c->stream));
    c->fr.localPixels = lp;
    return allocAov(c);
}

extern "C" {

// ---- The call-sequence state machine.  The library defers and fuses behind the reference's entry points; what may be deferred, fused,
// left raw or started early depends on WHAT WAS CALLED SINCE -- one explicit phase, one transition function, every entry point declares
// its class of call.  (Rounds 2-3 kept this in ten booleans and five macros; tests/test_gpu_fuzz.py covers the (phase, call) pairs.)
//   PH_IDLE                nothing deferred, nothing known about the calls since the last `logic`
//   PH_DEFER_LOGIC         flx_wf_logic was called and is DEFERRED: the next call decides whether it runs fused with the material kernels
//   PH_DEFER_LOGIC_RAYGEN  ... and flx_wf_raygen behind it, deferred along (its queue does not exist yet)
//   PH_CHAIN               `logic` has been launched and only genRays / material kernels were enqueued since: the shadow kernel's inputs are
//                          complete and nothing enqueued since touches them (flx_wf_shadow below)
//   PH_CHAIN_EXT           ... and the extension kernel is the last thing enqueued: flx_wf_shadow may start right behind `logic` (overlap 2)
//   PH_EXT                 the extension kernel is the last thing enqueued, the chain since `logic` is broken: flx_wf_shadow runs beside it (overlap 1, 2)
// Orthogonal DATA flags stay what they are: rawHits (hit records of the last extension launch are RAW, flx_trace.h), matQueuesEmpty,
// qs.extPend (lazy extension counter), cursorDirty.
enum Phase { PH_IDLE = 0, PH_DEFER_LOGIC = 1, PH_DEFER_LOGIC_RAYGEN = 2, PH_CHAIN = 3, PH_CHAIN_EXT = 4, PH_EXT = 5 };
enum Call {
    CALL_LOGIC, CALL_RAYGEN, CALL_MATERIALS, CALL_EXTEND, CALL_SHADOW,
    CALL_QUIET,        // enqueues at most a read-back of counters / nothing: flx_get_counters_async, flx_finish, flx_counter_totals, flx_profile_enable
    CALL_NEUTRAL,      // touches counters, cursor or framebuffer, never a hit record: flx_clear_queues, flx_pixel_index_*, flx_end_iteration_async, flx_read_pixels
    CALL_PEEK,         // may observe hit records or queues, or changes how later kernels run, without enqueueing work of its own: flx_stream, flx_queue_read, trace-stat getters, plain options
    CALL_OBSERVE       // everything else: exports, imports, uploads, parameters, resets, options that re-plan the schedule, the microkernels, the gather
};
struct Step { bool launchDeferred, commitRaw; int next; };
static Step transition(int ph, Call call)
{
    const bool deferred = ph == PH_DEFER_LOGIC || ph == PH_DEFER_LOGIC_RAYGEN;
    const bool chain = deferred || ph == PH_CHAIN || ph == PH_CHAIN_EXT;       // only genRays / materials / extension since `logic`
    switch (call) {
    case CALL_LOGIC:     return {deferred, false, PH_IDLE};                     // (flx_wf_logic then enters PH_DEFER_LOGIC or PH_CHAIN; it hands RAW records to the fused pass or commits them itself)
    case CALL_RAYGEN:    if (ph == PH_DEFER_LOGIC) return {false, false, PH_DEFER_LOGIC_RAYGEN};
                         return {deferred, false, chain ? PH_CHAIN : PH_IDLE};  // genRays reads no hit record (and its paths' records are dead: flx_device.h)
    case CALL_MATERIALS: if (deferred) return {false, false, PH_CHAIN};         // the fused pass runs now (flx_wf_materials)
                         return {false, true, chain ? PH_CHAIN : PH_IDLE};      // the separate material kernels read the hit records
    case CALL_EXTEND:    return {deferred, true, chain ? PH_CHAIN_EXT : PH_EXT};   // (a second extension launch needs the first one's records committed: pathLen)
    case CALL_SHADOW:    return {deferred, false, PH_IDLE};                     // {shadowOrig, shadowDir} -> shadowRayBlocked: no hit record
    case CALL_QUIET:     return {deferred, false, deferred ? PH_CHAIN : ph};
    case CALL_NEUTRAL:   return {deferred, false, PH_IDLE};
    case CALL_PEEK:      return {deferred, true, deferred ? PH_CHAIN : ph};
    case CALL_OBSERVE:   return {deferred, true, PH_IDLE};
    }
    return {deferred, true, PH_IDLE};
}
static int enter(flx_ctx *c, Call call);
#define ENTER(c, call) do { if (enter(c, call)) return 1; } while (0)
// lazy extension counter (flx_device.h): make counters[EXTENSION] in memory current before anything outside the
// raygen / material / extension / end-of-iteration kernels looks at it or overwrites the source counters
static void flushExt(flx_ctx *c) { if (c->qs.extPend) { launch_bump_extension(c->stream, c->qs.counters, c->qs.extPend); c->qs.extPend = 0; } }
const char *flx_last_error(flx_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

// refill = refillMin | waitMax << 8 (trace4r.hip).  refillMin 0 with a waitMax of 1..63 would end every descent round before a node is
// visited (0 finished lanes >= refillMin) while no lane is idle for the refill to serve: the kernel would spin forever.  0 = the
// thread-per-ray kernel; otherwise refillMin 1..64 and waitMax 0 (= 64) .. 64.
static bool refill_value_ok(int v) { return v == 0 || (v > 0 && (v & 0xFF) >= 1 && (v & 0xFF) <= 64 && (v >> 8) <= 64); }

int flx_create(int device, uint32_t num_tasks, flx_ctx **out)
{
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_create_error = "flx_create: no HIP device available (libfluctus_hip.so has no CPU fallback)"; return 1; }
    if (device < 0 || device >= ndev) { g_create_error = "flx_create: bad device index"; return 1; }
    if (num_tasks == 0) { g_create_error = "flx_create: num_tasks must be > 0"; return 1; }
    flx_ctx *c = new flx_ctx();
    c->device = device; c->numTasks = num_tasks;
#ifdef FLX_LAB
    // lab build only (scripts/build_variants.py, -DFLX_LAB): A/B hooks for whole test-suite runs -- the defaults of the refill_extend /
    // refill_shadow options.  The shipped library reads no environment variable here.
    if (const char *e = getenv("FLX_REFILL_EXTEND")) { const int v = atoi(e); if (refill_value_ok(v)) c->refillExt = v; }
    if (const char *e = getenv("FLX_REFILL_SHADOW")) { const int v = atoi(e); if (v < 0 || refill_value_ok(v)) { c->refillShadowOpt = v; c->refillShadow = v < 0 ? 0 : v; } }
#endif
    auto fail = [&](const char *what, hipError_t err) { g_create_error = std::string(what) + ": " + hipGetErrorString(err); flx_destroy(c); return 1; };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    // (stream priorities were tried: a high-priority shadow stream keeps the extension kernel at its undisturbed 0.86 ms and inflates
    //  the material kernel instead, a high-priority main stream changes nothing -- resident waves are not displaced; the step time
    //  stays within 0.7 % in every combination, so both streams have the default priority)
#ifndef FLX_STREAM_PRIO          // 0: both streams at the default priority | 1: the main stream (logic -> genRays -> materials -> closest hit) above the any-hit stream
#define FLX_STREAM_PRIO 0
#endif
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const int pMain = FLX_STREAM_PRIO ? greatest : 0, pSecond = FLX_STREAM_PRIO ? least : 0;
        if ((e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, pMain)) != hipSuccess) return fail("hipStreamCreate", e);
        if ((e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, pSecond)) != hipSuccess) return fail("hipStreamCreate", e);
    }
    if ((e = hipEventCreateWithFlags(&c->evPreExt, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&c->evShadow, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->evPostLogic, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    const size_t N = num_tasks;
    c->st.numTasks = num_tasks;
    // (staggering the twelve arrays inside their allocations -- 272 / 4112 / 65808 elements per array -- changes nothing: the fused pass lands on
    //  one of three levels, 0.457 / 0.479 / 0.507 ms, from one process to the next with or without it, and so does ONE allocation for all twelve; profiles/r03_state_stagger_ab.txt, r03_state_slab_ab.txt)
    for (int r = 0; r < S_NUM_REC; r++) {
        if (dalloc(c, c->fixedAllocs, &c->st.rec[r], N)) return fail("hipMalloc(state)", hipErrorOutOfMemory);
        (void)hipMemsetAsync(c->st.rec[r], 0, N * sizeof(float4), c->stream);
    }
    if (dalloc(c, c->fixedAllocs, &c->st.phase, N) || dalloc(c, c->fixedAllocs, &c->mkStats, 4)) return fail("hipMalloc(state)", hipErrorOutOfMemory);
    (void)hipMemsetAsync(c->st.phase, 0, N * 4, c->stream); (void)hipMemsetAsync(c->mkStats, 0, 16, c->stream);
    if (dalloc(c, c->fixedAllocs, &c->st.blocked, N) || dalloc(c, c->fixedAllocs, &c->st.pickProb, N) || dalloc(c, c->fixedAllocs, &c->st.firstDiffuse, N))
        return fail("hipMalloc(state)", hipErrorOutOfMemory);
    (void)hipMemsetAsync(c->st.blocked, 0, N * 4, c->stream); (void)hipMemsetAsync(c->st.pickProb, 0, N * 4, c->stream); (void)hipMemsetAsync(c->st.firstDiffuse, 0, N * 4, c->stream);
    for (int q = 0; q < FLX_NUM_QUEUES; q++) {
        if (dalloc(c, c->fixedAllocs, &c->qs.q[q], N)) return fail("hipMalloc(queue)", hipErrorOutOfMemory);
        (void)hipMemsetAsync(c->qs.q[q], 0, N * 4, c->stream);
    }
    if (dalloc(c, c->fixedAllocs, &c->qs.counters, 8) || dalloc(c, c->fixedAllocs, &c->qs.cursors, FLX_NUM_BLOCK_CURSORS * FLX_CURSOR_STRIDE)) return fail("hipMalloc(counters)", hipErrorOutOfMemory);
    (void)hipMemsetAsync(c->qs.counters, 0, 32, c->stream);
    (void)hipMemsetAsync(c->qs.cursors, 0, 4 * FLX_NUM_BLOCK_CURSORS * FLX_CURSOR_STRIDE, c->stream);
    const size_t auxStride = logic_aux_stride(num_tasks);          // per list, padded for the scan kernel's uint4 accesses
    if (dalloc(c, c->fixedAllocs, &c->member, N) || dalloc(c, c->fixedAllocs, &c->blockCounts, (size_t)7 * auxStride) || dalloc(c, c->fixedAllocs, &c->blockOffsets, (size_t)7 * auxStride))
        return fail("hipMalloc(logic aux)", hipErrorOutOfMemory);
    // (the look-back words of option "regen" and the continuation records of option "shadow_split" -- both off by default, together ~45 B per path -- are
    //  allocated when the option is first switched on: optionBuffers)
    if (dalloc(c, c->fixedAllocs, &c->logicError, 1)) return fail("hipMalloc(logic error flag)", hipErrorOutOfMemory);
    (void)hipMemsetAsync(c->logicError, 0, 4, c->stream);
    (void)hipMemsetAsync(c->blockCounts, 0, (size_t)7 * auxStride * 4, c->stream);      // the pad behind each list's counts stays zero
    (void)hipMemsetAsync(c->blockOffsets, 0, (size_t)7 * auxStride * 4, c->stream);
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->numCUs = prop.multiProcessorCount; }
    if (dalloc(c, c->fixedAllocs, &c->stats, FLX_NUM_TRACE_STATS)) return fail("hipMalloc(stats)", hipErrorOutOfMemory);
    (void)hipMemsetAsync(c->stats, 0, FLX_NUM_TRACE_STATS * 8, c->stream);
#ifdef FLX_LAB_RSTATS
    flxd::g_lab_rstats = c->stats;
#endif
    // per sub-list: numTasks / 2 (first pass) and / 8 (second pass) records over all lists, whole waves (allocated by optionBuffers)
    c->splitCapA = ((num_tasks / 2 / shadow_split_lists()) + 64u) & ~63u; c->splitCapB = ((num_tasks / 8 / shadow_split_lists()) + 64u) & ~63u;
    if (dalloc(c, c->fixedAllocs, &c->totals, 8)) return fail("hipMalloc(totals)", hipErrorOutOfMemory);
    (void)hipMemsetAsync(c->totals, 0, 64, c->stream);
    if (dalloc(c, c->fixedAllocs, &c->fr.currPixelIdx, 1)) return fail("hipMalloc(cursor)", hipErrorOutOfMemory);
    (void)hipMemsetAsync(c->fr.currPixelIdx, 0, 4, c->stream);
    c->fr.rank = 0; c->fr.nranks = 1; c->fr.localPixels = 0;
    if ((e = hipHostMalloc((void **)&c->pinned, sizeof(flx_queue_counters) * c->pinnedSlots)) != hipSuccess) return fail("hipHostMalloc", e);
    if ((e = hipHostMalloc((void **)&c->pinnedIdx, sizeof(uint32_t) * c->pinnedSlots)) != hipSuccess) return fail("hipHostMalloc", e);
    if ((e = hipHostMalloc((void **)&c->pinnedMk, 16 * c->pinnedSlots)) != hipSuccess) return fail("hipHostMalloc", e);
    // dummy 1x1 black environment map (reference: CLContext::setupScene, src/clcontext.cpp:513-518)
    {
        float4 *rgba; float2 *rec; float *pdf;
        if (dalloc(c, c->envAllocs, &rgba, 1) || dalloc(c, c->envAllocs, &rec, 1) || dalloc(c, c->envAllocs, &pdf, 1))
            return fail("hipMalloc(env)", hipErrorOutOfMemory);
        const float one = 1.0f; const float2 rec1 = make_float2(1.0f, 0.0f /* alias 0 */);
        (void)hipMemsetAsync(rgba, 0, 16, c->stream);
        (void)hipMemcpy(rec, &rec1, 8, hipMemcpyHostToDevice); (void)hipMemcpy(pdf, &one, 4, hipMemcpyHostToDevice);
        c->sc.envRGBA = rgba; c->sc.aliasRec = rec; c->sc.pdfTable = pdf; c->sc.envW = c->sc.envH = 1;
        float4 *nee; if (dalloc(c, c->envAllocs, &nee, 2)) return fail("hipMalloc(env)", hipErrorOutOfMemory);
        launch_env_nee_table(c->stream, c->sc, nee, 1u); c->sc.neeRec = nee;
    }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return fail("hipStreamSynchronize", e);
    *out = c;
    return 0;
}

int flx_destroy(flx_ctx *c)
{
    if (!c) return 0;
    c->phase = PH_IDLE;                                 // deferred kernels of a context that is going away: dropped
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { flx_group_destroy(c); }
    freeAll(c->gatherAllocs);
    freeAll(c->sceneAllocs); freeAll(c->spillAllocs); freeAll(c->envAllocs); freeAll(c->frameAllocs); freeAll(c->aovAllocs); freeAll(c->fixedAllocs);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinnedIdx) (void)hipHostFree(c->pinnedIdx);
    if (c->pinnedMk) (void)hipHostFree(c->pinnedMk);
    for (auto &ev : c->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto e : c->eventPool) (void)hipEventDestroy(e);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->evPreExt) (void)hipEventDestroy(c->evPreExt);
    if (c->evShadow) (void)hipEventDestroy(c->evShadow);
    if (c->evPostLogic) (void)hipEventDestroy(c->evPostLogic);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

uint32_t flx_num_tasks(flx_ctx *c) { return c->numTasks; }
// (an interop caller enqueues its own work behind ours on this stream: a deferred flx_wf_logic / flx_wf_raygen must be in it by then)
void *flx_stream(flx_ctx *c) { (void)enter(c, CALL_PEEK); return (void *)c->stream; }

// How the two traversals share the machine.  Two persistent kernels cannot run side by side (each fills every wave slot), so the second stream
// serves the THREAD-PER-RAY any-hit kernel: started right after `logic` (schedule 2) it runs beside genRays / the material kernels and then
// fills the slots the persistent closest-hit kernel's waves leave as they retire.  Measured with the final round-3 kernels (blocks handed out
// on demand; profiles/r03_wave_slots_sweep.txt, one box, Mrays/s, schedule 1 / schedule 2 / serial with a persistent any-hit kernel):
//   kitchen 4951-5249 / 5208-5489 / 4800-4840     conference 4931-4996 / 5017-5019 / 4590-4720     courtyard 2165 / 2181 / 1998-2076
// (an earlier build with a static share of blocks per wave preferred the serial schedule for the courtyard, whose tree comes from HBM; with
// balanced waves it does not).  Options "overlap" and "refill_shadow" override; -1 = these defaults.
static void pickSchedule(flx_ctx *c)
{
    c->overlap = c->overlapOpt >= 0 ? c->overlapOpt : 2;
    c->refillShadow = c->refillShadowOpt >= 0 ? c->refillShadowOpt : 0;
}

// Device buffers of the two features that are off by default, allocated when the option is first switched on (round 5's advisor: at 16 M paths the
// continuation records alone were 670 MB that the default configuration never touched) and kept until the context goes away.
static int optionBuffers(flx_ctx *c, bool split, bool regen)
{
    HIPCHK(c, hipSetDevice(c->device));
    if (split && !c->splitRecA) {
        uint32_t *cnt = nullptr; uint4 *a = nullptr, *b = nullptr;
        if (dalloc(c, c->fixedAllocs, &cnt, shadow_split_count_words()) || dalloc(c, c->fixedAllocs, &a, (size_t)c->splitCapA * shadow_split_lists() * 4) ||
            dalloc(c, c->fixedAllocs, &b, (size_t)c->splitCapB * shadow_split_lists() * 4)) { c->err = "flx_set_option(shadow_split): out of device memory for the continuation records"; return 1; }
        HIPCHK(c, hipMemsetAsync(cnt, 0, (size_t)shadow_split_count_words() * 4, c->stream));
        c->splitCounts = cnt; c->splitRecA = a; c->splitRecB = b;
    }
    if (regen && !c->lookback) {
        unsigned long long *lb = nullptr;
        if (dalloc(c, c->fixedAllocs, &lb, (size_t)logic_lookback_words(c->numTasks))) { c->err = "flx_set_option(regen): out of device memory for the look-back words"; return 1; }
        HIPCHK(c, hipMemsetAsync(lb, 0, (size_t)logic_lookback_words(c->numTasks) * 8, c->stream));
        c->lookback = lb;
    }
    return 0;
}

// ---- scene upload: reference wire arrays -> traversal layout -------------------------------
int flx_upload_scene(flx_ctx *c, const void *trisv, size_t ntris, const uint32_t *indices, size_t nidx,
                     const void *nodesv, size_t nnodes, const void *materials, size_t nmat,
                     const void *texdesc, size_t ntex, const uint8_t *texdata, size_t texbytes)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, trisv && ntris && indices && nidx && nodesv && nnodes, "flx_upload_scene: empty scene");
    NEED(c, materials && nmat, "flx_upload_scene: at least the default material is required");
    HIPCHK(c, hipSetDevice(c->device));
    const flx_triangle *tris = (const flx_triangle *)trisv;
    const flx_node *nodes = (const flx_node *)nodesv;

    // 0. which BSDF types the fused logic+material pass inlines for this scene (logic.hip).  Inlining a type costs registers whether or not
    // a path of that type shows up, routing a type through its queue costs a second trip over the path state: measured on the three bench
    // scenes, a mostly-diffuse scene (kitchen 96 %, courtyard 65 % of the surface area) wants the diffuse step alone inline (+1..3 % Mrays/s
    // over the separate kernels, inlining everything +-0), a scene whose surfaces are mostly glossy / GGX (conference: 13 % diffuse) wants
    // them all (+11 %).  The reference specialises its kernels per scene too (-DBXDF_USE_*).  Option "fuse_set" overrides.
    {
        const flx_material *mats = (const flx_material *)materials;
        double areaAll = 0.0, areaDiffuse = 0.0;
        for (size_t i = 0; i < ntris; i++) {
            const flx_triangle &t = tris[i];
            const double ax = (double)t.v1.p.x - t.v0.p.x, ay = (double)t.v1.p.y - t.v0.p.y, az = (double)t.v1.p.z - t.v0.p.z;
            const double bx = (double)t.v2.p.x - t.v0.p.x, by = (double)t.v2.p.y - t.v0.p.y, bz = (double)t.v2.p.z - t.v0.p.z;
            const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
            const double a = std::sqrt(cx * cx + cy * cy + cz * cz);
            areaAll += a;
            if (t.matId >= 0 && (size_t)t.matId < nmat && mats[t.matId].type == FLX_BXDF_DIFFUSE) areaDiffuse += a;
        }
        // Round 3 (persistent closest hit, RAW commit in the pass, shadow rays on the second stream; profiles/r03_fuse_set_ab.txt, same box, diffuse ->
        // all): courtyard (65 % diffuse) 2158 -> 2258 and 2221 -> 2274 Mrays/s at 1440p, 2088 -> 2167 and 2189 -> 2200 at 2160p -- a third of
        // its paths took the second trip -- kitchen (96 %) 5297 -> 5152 and 5485 -> 5198.  Hence all types below 3/4 diffuse (round 2: 1/2).
        c->fuseSet = (areaAll > 0.0 && areaDiffuse < 0.75 * areaAll) ? 31 : 1;
        // ... and whether the all-types pass sorts its material step by BSDF type inside each block (logic.hip: LOGIC_REGROUP, k_logic<31, true, true>).  Round 5's
        // build of it needed 119 VGPRs (4 waves per SIMD) and paid only where one type dominates (profiles/r05_regroup_ab.txt); as a template instance of its own,
        // compiled for 5 blocks per CU, it fits 96 VGPRs without scratch, and the same-box A/B at 16 M paths reads (profiles/r06_regroup_ab.txt, off -> on, Mrays/s):
        // conference 5392 -> 5678 and 5376 -> 5654 (+5.2 %), courtyard-1440p 2512 -> 2503 and 2483 -> 2491, egyptcat 6079 -> 6074 and 6009 -> 6042 (both +-0.5 %: the
        // box's spread).  On whenever the all-types pass runs; option "regroup" overrides.
        c->regroupAuto = 1;
        c->regroup = c->regroupOpt >= 0 ? c->regroupOpt : c->regroupAuto;
        // ... and the order in which the fused pass lists the continuing paths in the extension queue (logic.hip: k_queue_scatter): one
        // segment per material queue, as the separate kernels append them, or all of them by path id.  Same-box A/B, Mrays/s segments ->
        // path id: conference 4318 -> 4446 (+3 %: three BSDF types of similar weight, the segments cut the id order into thirds),
        // kitchen 4278 -> 4230, courtyard 1678 -> 1652 (one dominant type: its segment IS the id order, and the small segments of the
        // other types are rays leaving the same few objects).  Option "ext_order" overrides.
        // Round 4, 8 M paths (profiles/r04_ext_order_ab.txt, same box): with the diffuse-only pass the kitchen's closest-hit kernel takes 1.056 ms on
        // the per-queue segments, 1.037 by path id, 1.035 with the regenerated paths merged in (ext_order 2: the queue is the identity permutation in
        // the steady state; step +1 %); conference and courtyard (all-types pass) do not move between 1 and 2.  Hence 2 with the diffuse-only pass.
        c->extOrder = c->fuseSet == 31 ? 1 : 2;
    }

    // 1. leaf triangle records, in index-list order (a leaf is a contiguous run of the list)
    std::vector<TriRec> trirecs(nidx);
    for (size_t s = 0; s < nidx; s++) {
        NEED(c, indices[s] < ntris, "flx_upload_scene: index out of range");
        const flx_triangle &t = tris[indices[s]];
        int idx = (int)indices[s], zero = 0;
        float fi, fz; memcpy(&fi, &idx, 4); memcpy(&fz, &zero, 4);
        trirecs[s].a = make_float4(t.v0.p.x, t.v0.p.y, t.v0.p.z, fi);
        trirecs[s].b = make_float4(t.v1.p.x, t.v1.p.y, t.v1.p.z, fz);
        trirecs[s].c = make_float4(t.v2.p.x, t.v2.p.y, t.v2.p.z, 0.0f);
    }
    // 2. inner-node records: both child boxes + refs; DFS numbering of inner nodes only
    // Record numbering ("sibling pairs"): the vector L1 and the L2 move 128-B lines, a BNode is 64 B.  The two inner children
    // of a node get the two halves of ONE 128-B-aligned line, allocated when their parent is numbered (pre-order, so a
    // root-to-leaf path stays roughly contiguous): descending into the nearer child brings the farther child's record
    // along, and the later pop of that sibling finds its line in L1/L2 instead of missing.  Single inner children are
    // packed two to a line.  Option node_layout 0 (set before the upload) = plain DFS numbering, for A/B.
    std::vector<int32_t> innerId(nnodes, -1);
    uint32_t ninner = 0, nrecords = 0;
    for (size_t i = 0; i < nnodes; i++) if (nodes[i].nPrims == 0) ninner++;
    const int nodeLayout = c->nodeLayout;
    if (nodeLayout == 0 || ninner == 0) {
        for (size_t i = 0; i < nnodes; i++) if (nodes[i].nPrims == 0) innerId[i] = (int32_t)nrecords++;     // reference DFS order
    } else {
        std::vector<uint32_t> todo; todo.reserve(128);
        innerId[0] = 0; nrecords = 2;                                   // the root's line-mate stays empty
        int32_t spare = -1;                                             // free half of a line opened for a single inner child
        todo.push_back(0);
        while (!todo.empty()) {
            const uint32_t i = todo.back(); todo.pop_back();
            const uint32_t l = i + 1, r = nodes[i].iStartOrRight;
            NEED(c, l < nnodes && r < nnodes, "flx_upload_scene: child index out of range");
            const bool li = nodes[l].nPrims == 0, ri = nodes[r].nPrims == 0;
            // an inner child that already has a record is reachable twice: cyclic or shared node array (e.g. a corrupt cache file)
            NEED(c, !(li && innerId[l] >= 0) && !(ri && innerId[r] >= 0) && r > i, "flx_upload_scene: malformed node array (node reachable twice)");
            if (li && ri) { innerId[l] = (int32_t)nrecords; innerId[r] = (int32_t)nrecords + 1; nrecords += 2; }
            else if (li || ri) {
                const uint32_t ch = li ? l : r;
                if (spare >= 0) { innerId[ch] = spare; spare = -1; }
                else { innerId[ch] = (int32_t)nrecords; spare = (int32_t)nrecords + 1; nrecords += 2; }
            }
            if (ri) todo.push_back(r);                                  // left subtree first
            if (li) todo.push_back(l);
        }
        for (size_t i = 0; i < nnodes; i++) NEED(c, nodes[i].nPrims != 0 || innerId[i] >= 0, "flx_upload_scene: inner node unreachable from the root");
    }
    auto childRef = [&](uint32_t ni, bool &ok) -> uint32_t {
        if (ni >= nnodes) { ok = false; return 0; }
        const flx_node &n = nodes[ni];
        if (n.nPrims == 0) return (uint32_t)innerId[ni];
        if ((size_t)n.iStartOrRight + n.nPrims > nidx) { ok = false; return 0; }
        int cnt = n.nPrims; float fc; memcpy(&fc, &cnt, 4);
        trirecs[n.iStartOrRight].b.w = fc;               // leaf count lives in the run's first record
        { uint32_t one = 1u; float fl; memcpy(&fl, &one, 4); trirecs[n.iStartOrRight + n.nPrims - 1].c.w = fl; }   // end-of-run flag (trace_mode 3)
        return FLX_LEAF_BIT | n.iStartOrRight;
    };
    std::vector<BNode> bnodes(ninner ? nrecords : 1);
    memset(bnodes.data(), 0, bnodes.size() * sizeof(BNode));
    bool ok = true;
    if (ninner == 0) {
        // the whole scene is one leaf: synthetic root whose two children are that leaf
        BNode &b = bnodes[0];
        const flx_node &n = nodes[0];
        const float mn[3] = {n.bmin.x, n.bmin.y, n.bmin.z}, mx[3] = {n.bmax.x, n.bmax.y, n.bmax.z};
        for (int k = 0; k < 3; k++) { b.lmin[k] = b.rmin[k] = mn[k]; b.lmax[k] = b.rmax[k] = mx[k]; }
        b.left = b.right = childRef(0, ok); b.pad[0] = b.pad[1] = 0;
    } else {
        for (size_t i = 0; i < nnodes; i++) {
            if (nodes[i].nPrims != 0) continue;
            BNode &b = bnodes[innerId[i]];
            uint32_t l = (uint32_t)i + 1, r = nodes[i].iStartOrRight;
            NEED(c, l < nnodes && r < nnodes, "flx_upload_scene: child index out of range");
            const flx_node &ln = nodes[l], &rn = nodes[r];
            b.lmin[0] = ln.bmin.x; b.lmin[1] = ln.bmin.y; b.lmin[2] = ln.bmin.z; b.lmax[0] = ln.bmax.x; b.lmax[1] = ln.bmax.y; b.lmax[2] = ln.bmax.z;
            b.rmin[0] = rn.bmin.x; b.rmin[1] = rn.bmin.y; b.rmin[2] = rn.bmin.z; b.rmax[0] = rn.bmax.x; b.rmax[1] = rn.bmax.y; b.rmax[2] = rn.bmax.z;
            b.left = childRef(l, ok); b.right = childRef(r, ok); b.pad[0] = b.pad[1] = 0;
        }
    }
    NEED(c, ok, "flx_upload_scene: malformed node array");
    // 3. shading records per ORIGINAL triangle index
    std::vector<ShadeRec> shade(ntris);
    for (size_t i = 0; i < ntris; i++) {
        const flx_triangle &t = tris[i];
        float fm; int m = t.matId; memcpy(&fm, &m, 4);
        NEED(c, m >= 0 && (size_t)m < nmat, "flx_upload_scene: triangle material id out of range");
        shade[i].a = make_float4(t.v0.n.x, t.v0.n.y, t.v0.n.z, t.v0.t.x);
        shade[i].b = make_float4(t.v1.n.x, t.v1.n.y, t.v1.n.z, t.v0.t.y);
        shade[i].c = make_float4(t.v2.n.x, t.v2.n.y, t.v2.n.z, t.v1.t.x);
        shade[i].d = make_float4(t.v1.t.y, t.v2.t.x, t.v2.t.y, fm);
    }
    // 4. the 4-wide quantised tree over the same leaves (flx_wide.h) + the depth of the binary tree (stack-spill sizing)
    // (Round 4 re-optimised the inner topology over the reference's leaves before this collapse -- subtree reinsertion, archived in
    //  scripts/experiments/flx_wide_opt.h: node visits -0.8 % kitchen / -5 % conference / -1.3 % courtyard on the device, both traversal
    //  kernels within 0-3 %, 25 s more upload time on the courtyard; below the bar, not shipped.  profiles/r04_wide_opt_ab.txt)
    flxw::WideTree wide;
    { const char *werr = nullptr; if (!flxw::build_wide(nodes, nnodes, tris, ntris, indices, nidx, wide, &werr)) { c->err = std::string("flx_upload_scene: ") + werr; return 1; } }
    uint32_t binDepth = 1;
    {   // nodes are in DFS order with parent < child (checked above for the right child; the left child is i + 1)
        std::vector<uint16_t> depth(nnodes, 0);
        for (size_t i = 0; i < nnodes; i++) {
            if (nodes[i].nPrims != 0) continue;
            const uint32_t l = (uint32_t)i + 1, r = nodes[i].iStartOrRight;
            NEED(c, r > i && r < nnodes && l < nnodes, "flx_upload_scene: malformed node array");
            const uint16_t dd = (uint16_t)(depth[i] + 1);
            NEED(c, dd < 4096, "flx_upload_scene: tree deeper than 4095 levels");
            depth[l] = dd; depth[r] = dd;
            if (dd > binDepth) binDepth = dd;
        }
    }
    uint32_t spillLevels = 1;
    if (binDepth + 1 > LDS_LEVELS) spillLevels = binDepth + 1 - LDS_LEVELS;
    // the 4-wide kernels page whole groups of 8 levels between their LDS ring and level-indexed spill rows (flx_trace4.h)
    if (wide.maxStack > WIDE_LDS_LEVELS - 4 && wide.maxStack + 8 > spillLevels) spillLevels = wide.maxStack + 8;

    // Allocate and fill the new scene first; the previous one is released (and c->sc switched) only when everything succeeded,
    // so a failed upload leaves the context on its old scene instead of on dangling pointers.
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    std::vector<void *> fresh, freshSpill;
    auto bail = [&]() { freeAll(fresh); freeAll(freshSpill); return 1; };
    BNode *dB; TriRec *dT; ShadeRec *dS; flx_triangle *dTri; flx_material *dM; flx_texdesc *dD; uint8_t *dX; flxw::WNode *dW; float4 *dL;
    if (dalloc(c, fresh, &dB, bnodes.size()) || dalloc(c, fresh, &dT, trirecs.size() + 1) || dalloc(c, fresh, &dS, shade.size()) ||
        dalloc(c, fresh, &dTri, ntris) || dalloc(c, fresh, &dM, nmat) || dalloc(c, fresh, &dD, ntex) || dalloc(c, fresh, &dX, texbytes + 4) ||
        dalloc(c, fresh, &dW, wide.nodes.size()) || dalloc(c, fresh, &dL, wide.leafdata.size() + 4))
        return bail();
    uint32_t *sp1 = c->spill, *sp2 = c->spill2;
    const size_t lanes = ((size_t)c->numTasks + 255) / 256 * 256 + 1024;
    const bool newSpill = spillLevels > c->spillLevels || !c->spill;
    if (newSpill && (dalloc(c, freshSpill, &sp1, lanes * spillLevels) || dalloc(c, freshSpill, &sp2, lanes * spillLevels))) return bail();
#define UPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(); } } while (0)
    UPCHK(hipMemcpy(dB, bnodes.data(), bnodes.size() * sizeof(BNode), hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(dT, trirecs.data(), trirecs.size() * sizeof(TriRec), hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(dS, shade.data(), shade.size() * sizeof(ShadeRec), hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(dTri, tris, ntris * sizeof(flx_triangle), hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(dM, materials, nmat * sizeof(flx_material), hipMemcpyHostToDevice));
    if (ntex) UPCHK(hipMemcpy(dD, texdesc, ntex * sizeof(flx_texdesc), hipMemcpyHostToDevice));
    if (texbytes) UPCHK(hipMemcpy(dX, texdata, texbytes, hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(dW, wide.nodes.data(), wide.nodes.size() * sizeof(flxw::WNode), hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(dL, wide.leafdata.data(), wide.leafdata.size() * sizeof(float4), hipMemcpyHostToDevice));
#undef UPCHK
    freeAll(c->sceneAllocs);
    c->sceneAllocs.swap(fresh);
    if (newSpill) { freeAll(c->spillAllocs); c->spillAllocs.swap(freshSpill); c->spill = sp1; c->spill2 = sp2; c->spillLevels = spillLevels; }
    c->sc.bnodes = dB; c->sc.trirecs = dT; c->sc.shade = dS; c->sc.tris = dTri; c->sc.materials = dM; c->sc.texdesc = dD; c->sc.texdata = dX;
    c->sc.rootRef = 0;
    c->sc.wnodes = dW; c->sc.wleaf = dL; c->sc.wrootRef = wide.rootRef;
    {   // flx_trace4.h, WRay::setup: which clamp of 1 / dir keeps (o - orig) * dinv finite for this scene
        const flx_node &r0 = nodes[0];
        const float ext[6] = {r0.bmin.x, r0.bmin.y, r0.bmin.z, r0.bmax.x, r0.bmax.y, r0.bmax.z};
        float m = 0.0f; for (float v : ext) m = std::fabs(v) > m ? std::fabs(v) : m;
        c->sc.wideClamp = m < 67108864.0f ? FLX_WIDE_DINV_MAX : FLX_WIDE_DINV_FAR;
    }
    // the exactness argument of the wide any-hit traversal needs nested boxes (flx_wide.h); a tree without them (no builder of
    // ours or of the reference produces one) is traversed with the binary kernels
    c->wideOK = wide.nested;
    c->wideInfo[0] = (uint32_t)wide.nodes.size(); c->wideInfo[1] = (uint32_t)(wide.leafdata.size()); c->wideInfo[2] = wide.maxStack; c->wideInfo[3] = wide.nested ? 1u : 0u;
    c->wideInfo[4] = binDepth; c->wideInfo[5] = spillLevels; c->wideInfo[6] = (uint32_t)bnodes.size(); c->wideInfo[7] = wide.maxLeafCount;
    pickSchedule(c);
    return 0;
}

int flx_upload_envmap(flx_ctx *c, const float *rgb, int w, int h, const float *prob, const int *alias, const float *pdf)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, rgb && prob && alias && pdf && w > 0 && h > 0, "flx_upload_envmap: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)w * h;
    std::vector<float4> rgba(n);
    for (size_t i = 0; i < n; i++) rgba[i] = make_float4(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2], 1.0f);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the probability and alias tables of the reference (src/envmap.cpp:31-114) merged into one record per texel (flx_device.h: aliasRec); the pdf table
    // stays as it is (env_map_pdf, and the per-texel NEE table below is built from it).  An alias outside the table (a malformed upload) is clamped like the kernel's own index clamp.
    std::vector<float2> rec(n);
    for (size_t i = 0; i < n; i++) {
        int a = alias[i]; if (a < 0) a = 0; if ((size_t)a >= n) a = (int)n - 1;
        float af; memcpy(&af, &a, 4);
        rec[i] = make_float2(prob[i], af);
    }
    // The new map is allocated and filled first; the previous one is released and c->sc switched only when every allocation, copy and the table
    // kernel have succeeded, so a failed upload leaves the context on its old map instead of on dangling pointers (round 5's advisor).
    std::vector<void *> fresh;
    float4 *dR; float2 *dRec; float *dF; float4 *dNee;
    auto bail = [&]() { freeAll(fresh); return 1; };
    if (dalloc(c, fresh, &dR, n) || dalloc(c, fresh, &dRec, n) || dalloc(c, fresh, &dF, n) || dalloc(c, fresh, &dNee, 2 * n)) return bail();
#define UPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(); } } while (0)
    UPCHK(hipMemcpy(dR, rgba.data(), n * 16, hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(dRec, rec.data(), n * 8, hipMemcpyHostToDevice));
    UPCHK(hipMemcpy(dF, pdf, n * 4, hipMemcpyHostToDevice));
    Scene tmp = c->sc;
    tmp.envRGBA = dR; tmp.aliasRec = dRec; tmp.pdfTable = dF; tmp.envW = w; tmp.envH = h;
    launch_env_nee_table(c->stream, tmp, dNee, (uint32_t)n); UPCHK(hipGetLastError());
    UPCHK(hipStreamSynchronize(c->stream));
#undef UPCHK
    freeAll(c->envAllocs);
    c->envAllocs.swap(fresh);
    c->sc.envRGBA = dR; c->sc.aliasRec = dRec; c->sc.pdfTable = dF; c->sc.envW = w; c->sc.envH = h; c->sc.neeRec = dNee;
    return 0;
}

int flx_set_params(flx_ctx *c, const void *p240)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, p240, "flx_set_params: null");
    HIPCHK(c, hipSetDevice(c->device));
    memcpy(&c->params, p240, sizeof(flx_render_params));   // kernels receive the struct by value at launch = in-order semantics
    NEED(c, c->params.width > 0 && c->params.height > 0, "flx_set_params: zero-sized framebuffer");
    c->haveParams = true;
    return allocFrame(c);
}

int flx_set_partition(flx_ctx *c, uint32_t rank, uint32_t nranks)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, nranks >= 1 && rank < nranks, "flx_set_partition: bad rank");
    c->fr.rank = rank; c->fr.nranks = nranks;
    return c->haveParams ? allocFrame(c) : 0;
}
uint32_t flx_local_pixels(flx_ctx *c) { return c->fr.localPixels; }

#define READY(c, call) do { ENTER(c, call); NEED(c, (c)->haveParams, "set params first (flx_set_params)"); NEED(c, (c)->sc.bnodes, "upload a scene first (flx_upload_scene)"); HIPCHK(c, hipSetDevice((c)->device)); } while (0)
#define LAUNCHED(c) HIPCHK(c, hipGetLastError())

int flx_wf_reset(flx_ctx *c) { READY(c, CALL_OBSERVE); flushExt(c); c->raygenQueueEmpty = false; /* k_reset fills the raygen queue */ { ScopedTimer t(c, FLX_K_RESET); launch_reset(c->stream, c->st, c->qs, c->fr, c->params); } LAUNCHED(c); return 0; }
// Appending a source queue a second time before the pending lengths were folded into the counter would compute slots from
// a base that counts the first append twice (ext_len) while extPend |= bit stays idempotent: flush first, so that every
// call order the reference's atomic append accepts (src/utils.cl:328-358) works here too.
static void flushExtIfPending(flx_ctx *c, uint32_t bits) { if (c->qs.extPend & bits) flushExt(c); }
static int runRaygen(flx_ctx *c, int appendExt = 1, bool alreadyDone = false, bool prepared = false)
{
    flushExtIfPending(c, 1u << FLX_Q_RAYGEN);
    // alreadyDone: the fused RAW pass of this chain regenerated the paths (and appended them) itself: only the bookkeeping of the call is left
    if (!alreadyDone) { ScopedTimer t(c, FLX_K_RAYGEN); launch_raygen(c->stream, c->st, c->qs, c->fr, c->params, appendExt, prepared ? 1 : 0); }
    c->qs.extPend |= 1u << FLX_Q_RAYGEN;
    if (c->eagerBump) flushExt(c);
    LAUNCHED(c);
    return 0;
}
// commit the RAW hit records the persistent-wave extension kernel left (trace4r.hip: k_materialise)
static int materialise(flx_ctx *c)
{
    if (!c->rawHits) return 0;
    c->rawHits = false;
    HIPCHK(c, hipSetDevice(c->device));
    launch_materialise(c->stream, c->st, c->sc, c->params, (uint32_t)c->numCUs);
    LAUNCHED(c);
    return 0;
}
// how this fused pass lists the traced paths in the extension queue: ext_order 2 (regenerated + continuing paths merged by path id) needs
// genRays to follow in the same chain -- the scatter writes the regenerated paths' entries, genRays then must not -- and falls back to 1
// (continuing paths by id, genRays appends its own block whenever it is called) otherwise
// ... and a raygen queue that was EMPTY before this logic pass: the merged list is ranked from the scan offsets, which start at the raygen counter's old
// value, while genRays with appendExt 0 would never fill the slots in front (flx_wf_reset leaves numTasks entries there without a clear: round 4's advisor)
static int extOrderFor(const flx_ctx *c, int fused, int raygenFirst) { return !fused ? 0 : (c->extOrder == 2 && (!raygenFirst || !c->raygenQueueEmpty)) ? 1 : c->extOrder; }
// the BSDF set the fused pass inlines NOW: the scene's choice (flx_upload_scene / option "fuse_set") with separate material queues; with a single material
// queue (WF_SINGLE_MAT_QUEUE: every BSDF type sits in the diffuse list) only a pass that inlines every type can serve it, so it is the all-types pass
// whatever the scene's choice says -- round 5: egyptcat under the reference's benchmark protocol ran the separate logic + k_material<31> + k_materialise
// (1.0 + 1.4 + 0.4 ms per 16 M paths) because its surfaces are mostly diffuse; the all-types pass takes their place: +16 % (profiles/r05_egyptcat_fuse.txt)
static int fuseSetNow(const flx_ctx *c) { return c->params.wfSeparateQueues ? c->fuseSet : 31; }
static int runLogic(flx_ctx *c, int first, int fused, int raygenFirst)
{
    // RAW hit records are committed by the fused pass itself when genRays follows in the same chain (logic.hip: k_logic<FUSE, RAW>); the plain
    // kernel and a chain without genRays get them committed first
    // ... and only when the pass covers EVERY path: with `first` set, logic stops at min(numTasks, pixels) (src/wf_logic.cl:45-48) and the paths
    // beyond would keep their RAW records (found by tests/test_gpu_fuzz.py, round 4)
    const int raw = (c->rawHits && fused != 0 && raygenFirst && !first) ? 1 : 0;
    if (!raw && materialise(c)) return 1;
    c->rawHits = false;
    flushExt(c);                                       // logic's scan overwrites the source-queue counters
#ifdef FLX_LAB_NOJOIN
    if (c->labIter >= 2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->evLab[c->labIter & 1u], 0));      // the shadow kernel BEFORE the last one
#endif
    // REGEN: the RAW pass regenerates its terminating paths itself when their index in the raygen queue is their rank among the terminating paths,
    // i.e. when the raygen queue was empty before this pass (as for ext_order 2); the genRays call of the chain then launches nothing (flx_wf_materials)
    const int order = extOrderFor(c, fused, raygenFirst);
    // (the look-back's status word carries the running prefix in 26 bits, logic.hip LB_VALUE: beyond 2^26 paths the genRays kernel does the job)
    const int regen = (raw && c->regenOpt && c->lookback && logic_can_regenerate() && c->raygenQueueEmpty && c->fr.localPixels > 0 && c->numTasks <= (1u << 26)) ? 1 : 0;
    c->regenDone = regen != 0;
    if (regen) c->regenUsed = true;
    // PREPARED REGENERATION (logic.hip; option "regen_prep", default on): the seed-only half of genRays inside the RAW pass, the pixel-dependent half in the
    // k_raygen that follows.  Same conditions as the in-kernel regeneration (every entry of the raygen queue is a path this pass terminated), minus the look-back.
    const int prep = (raw && !regen && c->prepOpt && c->raygenQueueEmpty && c->fr.localPixels > 0) ? 1 : 0;
    c->prepDone = prep != 0;
    if (++c->logicEpoch == 0u) c->logicEpoch = 1u;     // (epoch 0 = the zero-filled words of a fresh context)
    { ScopedTimer t(c, fused ? FLX_K_LOGIC_FUSED : FLX_K_LOGIC); launch_logic(c->stream, c->st, c->qs, c->sc, c->fr, c->params, c->member, c->blockCounts, c->blockOffsets, first, fused, raygenFirst, order, raw,
                                                                                c->lookback, c->logicEpoch, regen ? 1 : (prep ? 2 : 0), order == 2 ? 0 : 1, c->logicError, c->regroup); }
    LAUNCHED(c);
    c->matQueuesEmpty = false; c->raygenQueueEmpty = false;
    if (c->overlap == 2) HIPCHK(c, hipEventRecord(c->evPostLogic, c->stream));
    return 0;
}
static uint32_t materialBits(const flx_ctx *c)
{
    return c->params.wfSeparateQueues ? ((1u << FLX_Q_DIFFUSE) | (1u << FLX_Q_GLOSSY) | (1u << FLX_Q_GGX_REFL) | (1u << FLX_Q_GGX_REFR) | (1u << FLX_Q_DELTA)) : (1u << FLX_Q_DIFFUSE);
}
// one step of the state machine: launch what flx_wf_logic / flx_wf_raygen deferred as the separate kernels (the caller is not the
// flx_wf_materials that fuses them), commit RAW hit records if the call could observe one, enter the next phase
static int enter(flx_ctx *c, Call call)
{
    const Step st = transition(c->phase, call);
    if (st.launchDeferred) {
        const bool withRaygen = c->phase == PH_DEFER_LOGIC_RAYGEN;
        c->phase = PH_CHAIN;
        HIPCHK(c, hipSetDevice(c->device));
        if (runLogic(c, c->pendFirst, 0, 0)) return 1;     // (commits RAW hit records first)
        if (withRaygen && runRaygen(c)) return 1;
    }
    if (st.commitRaw && materialise(c)) return 1;
    c->phase = st.next;
    return 0;
}
int flx_wf_raygen(flx_ctx *c)
{
    if (c->phase == PH_DEFER_LOGIC) { ENTER(c, CALL_RAYGEN); return 0; }      // deferred behind the deferred flx_wf_logic (its queue does not exist yet)
    READY(c, CALL_RAYGEN);
    return runRaygen(c);
}
int flx_wf_extend(flx_ctx *c)
{
    READY(c, CALL_EXTEND);
    const bool chainIntact = c->phase == PH_CHAIN_EXT;
    // "everything enqueued before the extension kernel": what a concurrent shadow kernel waits for -- unless it may start right after
    // `logic` (overlap 2 with the chain intact: it then waits for evPostLogic instead, and this marker would only put one more barrier
    // packet in front of the extension kernel)
    if (c->overlap && !(c->overlap == 2 && chainIntact)) HIPCHK(c, hipEventRecord(c->evPreExt, c->stream));
    if (c->profile == 1 || c->profile == 2) { if (c->spanStart) c->eventPool.push_back(c->spanStart); c->spanStart = getEvent(c); (void)hipEventRecord(c->spanStart, c->stream); }
    {
        ScopedTimer t(c, FLX_K_EXTEND);
        if (c->extendTree == 4 && c->wideOK && c->refillExt > 0 && !c->statsOn) {
            uint32_t *cur = c->qs.cursors;
            if (c->cursorDirty[0]) HIPCHK(c, hipMemsetAsync(cur, 0, 4 * 8 * FLX_CURSOR_STRIDE, c->stream));      // (no k_end_iteration since the last launch)
            launch_extend4r(c->stream, c->st, c->qs, c->sc, c->params, c->spill, (uint32_t)c->numCUs, c->refillExt, cur);
            c->cursorDirty[0] = true; c->rawHits = true;
        }
        else if (c->extendTree == 4 && c->wideOK) launch_extend4(c->stream, c->st, c->qs, c->sc, c->params, c->spill, c->statsOn ? c->stats : nullptr);
        else launch_extend(c->stream, c->st, c->qs, c->sc, c->params, c->spill, c->statsOn ? c->stats : nullptr, c->xcdRemap);
    }
    LAUNCHED(c);
    return 0;
}
int flx_wf_shadow(flx_ctx *c)
{
    // The shadow kernel touches {shadowOrig, shadowDir, shadow queue} -> shadowRayBlocked, the extension kernel
    // {orig, dir, extension queue} -> hit + pathLen: disjoint.  When it directly follows flx_wf_extend (the reference's
    // order, src/tracer.cpp:253-254) it is launched on a second stream that only waits for the work enqueued BEFORE the
    // extension kernel, so the two traversals share the machine and fill each other's tails; the main stream then waits
    // for it, which keeps the single-in-order-queue semantics for everything that follows.
    // overlap 2: its inputs are complete when `logic` is (NEE lives there, src/wf_logic.cl:217-302).  What the reference
    // enqueues between logic and traceShadow -- genRays and the material kernels -- neither writes what the shadow kernel
    // reads nor reads what it writes: they work on {orig, dir, T, lastBsdf, hit record} of the raygen / material queues,
    // and a path is never in the raygen queue (terminated) and the shadow queue (continuing) of the same iteration, so
    // init_path_state's writes to shadowRayBlocked / shadowRayLen touch other paths.  So if ONLY those calls came since
    // flx_wf_logic, the second stream waits for logic alone and the latency-bound shadow traversal also overlaps the
    // HBM-bound raygen and material kernels.
    const bool overlapped = (c->phase == PH_CHAIN_EXT || c->phase == PH_EXT) && c->overlap != 0;
    const bool early = c->phase == PH_CHAIN_EXT && c->overlap == 2;
    READY(c, CALL_SHADOW);
    hipStream_t s = c->stream;
    if (overlapped) { s = c->stream2; HIPCHK(c, hipStreamWaitEvent(s, early ? c->evPostLogic : c->evPreExt, 0)); }
    hipEvent_t earlyStart = nullptr;
    if ((c->profile == 1 || c->profile == 2) && early) { earlyStart = getEvent(c); (void)hipEventRecord(earlyStart, s); }
    {
        ScopedTimer t(c, FLX_K_SHADOW, s);
        uint32_t *spill = overlapped ? c->spill2 : c->spill;
        if (c->shadowTree == 4 && c->wideOK && c->refillShadow > 0 && !c->statsOn) {
            uint32_t *cur = c->qs.cursors + 8 * FLX_CURSOR_STRIDE;
            if (c->cursorDirty[1]) HIPCHK(c, hipMemsetAsync(cur, 0, 4 * 8 * FLX_CURSOR_STRIDE, s));
            launch_shadow4r(s, c->st, c->qs, c->sc, c->params, spill, (uint32_t)c->numCUs, c->refillShadow, cur);
            c->cursorDirty[1] = true;
        }
#ifdef FLX_LAB_NOJOIN
        else if (c->shadowTree == 4 && c->wideOK && overlapped) {
            if (!c->labCounters) HIPCHK(c, hipMalloc(&c->labCounters, 2 * 32));
            Queues q2 = c->qs; q2.counters = c->labCounters + 8 * (c->labIter & 1u);
            HIPCHK(c, hipMemcpyAsync(q2.counters, c->qs.counters, 32, hipMemcpyDeviceToDevice, s));
            launch_shadow4(s, c->st, q2, c->sc, c->params, spill, nullptr);
        }
#endif
        else if (c->shadowTree == 4 && c->wideOK && c->shadowSplit > 0 && !c->statsOn)
            launch_shadow4_split(s, c->st, c->qs, c->sc, c->params, spill, c->splitCounts, c->splitParity++, c->splitRecA, c->splitCapA, c->splitRecB, c->splitCapB, c->shadowSplit & 0xFF, c->shadowSplit >> 8, c->splitLimit);
        else if (c->shadowTree == 4 && c->wideOK) launch_shadow4(s, c->st, c->qs, c->sc, c->params, spill, c->statsOn ? c->stats : nullptr);
        else launch_shadow(s, c->st, c->qs, c->sc, c->params, spill, c->statsOn ? c->stats : nullptr, c->xcdRemap);
    }
    LAUNCHED(c);
#ifdef FLX_LAB_NOJOIN
    if (overlapped) {
        if (!c->evLab[0]) { HIPCHK(c, hipEventCreateWithFlags(&c->evLab[0], hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->evLab[1], hipEventDisableTiming)); }
        HIPCHK(c, hipEventRecord(c->evLab[c->labIter & 1u], s)); c->labIter++;
        return 0;
    }
#endif
    if (overlapped) { HIPCHK(c, hipEventRecord(c->evShadow, s)); HIPCHK(c, hipStreamWaitEvent(c->stream, c->evShadow, 0)); }
    if ((c->profile == 1 || c->profile == 2) && overlapped && c->spanStart) {
        // span of the two traversals: from the earlier start (the shadow kernel's when it ran ahead) to the join
        hipEvent_t b = getEvent(c); (void)hipEventRecord(b, c->stream);
        if (earlyStart) { c->eventPool.push_back(c->spanStart); c->spanStart = earlyStart; earlyStart = nullptr; }
        c->events.push_back({FLX_K_TRACE_SPAN, c->spanStart, b}); c->spanStart = nullptr;
    }
    if (earlyStart) c->eventPool.push_back(earlyStart);
    return 0;
}
int flx_wf_logic(flx_ctx *c, int first)
{
    READY(c, CALL_LOGIC);                              // (runLogic commits RAW hit records, or hands them to the fused pass)
    // fused with the material kernels if flx_wf_materials follows (see flx_ctx::fuse).  The fused scatter numbers the material
    // queues from zero, so they must be empty now (cleared since the last logic: the reference clears all queues every iteration,
    // src/tracer.cpp:257); otherwise, and with the option off, the kernel runs here and now.
    const uint32_t allMat = (1u << FLX_Q_DIFFUSE) | (1u << FLX_Q_GLOSSY) | (1u << FLX_Q_GGX_REFL) | (1u << FLX_Q_GGX_REFR) | (1u << FLX_Q_DELTA);
    // (with a single material queue every BSDF type sits in the diffuse list: only a pass that inlines them all can serve it)
    const bool fusable = c->params.wfSeparateQueues || fused_queue_mask(fuseSetNow(c)) == allMat;      // (always: see fuseSetNow)
    if (c->fuse && c->matQueuesEmpty && fusable) { c->phase = PH_DEFER_LOGIC; c->pendFirst = first; }
    else { if (runLogic(c, first, 0, 0)) return 1; c->phase = PH_CHAIN; }
    return 0;
}
int flx_wf_materials(flx_ctx *c)
{
    const int before = c->phase;
    if (before == PH_DEFER_LOGIC || before == PH_DEFER_LOGIC_RAYGEN) {
        const bool withRaygen = before == PH_DEFER_LOGIC_RAYGEN;
        // [logic, materials] or [logic, raygen, materials]: one fused pass + scan + scatter, then the deferred genRays.  The
        // extension-queue slots are the ones the separate kernels compute in the caller's order: with genRays first the material
        // lists go behind the raygen queue, and genRays itself must not see them as pending yet.
        ENTER(c, CALL_MATERIALS);
        NEED(c, c->haveParams && c->sc.bnodes, "set params and upload a scene first");
        HIPCHK(c, hipSetDevice(c->device));
        const int fuseNow = fuseSetNow(c);
        const int order = extOrderFor(c, fuseNow, withRaygen);
        if (runLogic(c, c->pendFirst, fuseNow, withRaygen)) return 1;
        if (withRaygen && runRaygen(c, order == 2 ? 0 : 1, c->regenDone, c->prepDone)) return 1;
        c->regenDone = false; c->prepDone = false;
        // BSDF types the fused pass does not inline went to their queues as usual: the material kernel for those
        { ScopedTimer t(c, FLX_K_MATERIALS); launch_materials_after_fused(c->stream, c->st, c->qs, c->sc, fused_queue_mask(fuseNow), order); }
        LAUNCHED(c);
        c->qs.extPend |= materialBits(c);
        if (c->eagerBump) flushExt(c);
        return 0;
    }
    READY(c, CALL_MATERIALS);
    const uint32_t bits = materialBits(c);
    flushExtIfPending(c, bits);
    { ScopedTimer t(c, FLX_K_MATERIALS); launch_materials(c->stream, c->st, c->qs, c->sc, c->params.wfSeparateQueues); }
    c->qs.extPend |= bits;
    if (c->eagerBump) flushExt(c);
    LAUNCHED(c); return 0;
}
int flx_postprocess(flx_ctx *c) { READY(c, CALL_OBSERVE); { ScopedTimer t(c, FLX_K_POSTPROCESS); launch_postprocess(c->stream, c->fr, c->params); } LAUNCHED(c); return 0; }

// ---- microkernel integrator.  One path per pixel, framebuffers indexed by the path id: single-GPU only, the pixel
// partition belongs to the wavefront path (allocFrame sizes the buffers for the rank's LOCAL pixels).
#define MK_READY(c) do { READY(c, CALL_OBSERVE); NEED(c, (c)->fr.nranks == 1, "the microkernel integrator is single-GPU: flx_set_partition(ctx, 0, 1) first"); } while (0)
int flx_mk_reset(flx_ctx *c) { MK_READY(c); launch_mk_reset(c->stream, c->st, c->fr, c->params); LAUNCHED(c); return 0; }
int flx_mk_raygen(flx_ctx *c) { MK_READY(c); launch_mk_raygen(c->stream, c->st, c->params); LAUNCHED(c); return 0; }
int flx_mk_next_vertex(flx_ctx *c) { MK_READY(c); launch_mk_next_vertex(c->stream, c->st, c->sc, c->fr, c->params, c->spill, c->mkStats); LAUNCHED(c); return 0; }
int flx_mk_sample_bsdf(flx_ctx *c) { MK_READY(c); launch_mk_sample_bsdf(c->stream, c->st, c->sc, c->fr, c->params, c->spill, c->mkStats); LAUNCHED(c); return 0; }
int flx_mk_splat(flx_ctx *c) { MK_READY(c); launch_mk_splat(c->stream, c->st, c->fr, c->params, c->mkStats, 0); LAUNCHED(c); return 0; }
int flx_mk_splat_preview(flx_ctx *c) { MK_READY(c); launch_mk_splat(c->stream, c->st, c->fr, c->params, c->mkStats, 1); LAUNCHED(c); return 0; }
int flx_mk_stats_async(flx_ctx *c, void *out16)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, out16, "flx_mk_stats_async: null");
    HIPCHK(c, hipSetDevice(c->device));
    if ((int)c->pendingMk.size() >= c->pinnedSlots) { c->err = "too many outstanding stats reads; call flx_finish"; return 1; }
    int slot = c->nextMkSlot; c->nextMkSlot = (c->nextMkSlot + 1) % c->pinnedSlots;
    HIPCHK(c, hipMemcpyAsync(c->pinnedMk + 4 * slot, c->mkStats, 16, hipMemcpyDeviceToHost, c->stream));
    c->pendingMk.push_back({out16, slot});
    return 0;
}
int flx_mk_stats_reset(flx_ctx *c) { ENTER(c, CALL_OBSERVE); HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipMemsetAsync(c->mkStats, 0, 16, c->stream)); return 0; }

int flx_clear_queues(flx_ctx *c)
{
    ENTER(c, CALL_NEUTRAL);
    c->qs.extPend = 0; c->matQueuesEmpty = true; c->raygenQueueEmpty = true;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->qs.counters, 0, 32, c->stream));
    if (c->cursorDirty[0] || c->cursorDirty[1]) {       // the block cursors of the persistent traversal kernels go with the counters
        HIPCHK(c, hipMemsetAsync(c->qs.cursors, 0, 4 * FLX_NUM_BLOCK_CURSORS * FLX_CURSOR_STRIDE, c->stream));
        c->cursorDirty[0] = c->cursorDirty[1] = false;
    }
    return 0;
}

int flx_get_counters_async(flx_ctx *c, void *out32)
{
    NEED(c, out32, "flx_get_counters_async: null");
    ENTER(c, CALL_QUIET);
    HIPCHK(c, hipSetDevice(c->device));
    flushExt(c);
    if ((int)c->pending.size() >= c->pinnedSlots) { c->err = "too many outstanding counter reads; call flx_finish"; return 1; }
    int slot = c->nextSlot; c->nextSlot = (c->nextSlot + 1) % c->pinnedSlots;
    HIPCHK(c, hipMemcpyAsync(&c->pinned[slot], c->qs.counters, 32, hipMemcpyDeviceToHost, c->stream));
    c->pending.push_back({out32, slot});
    return 0;
}

int flx_finish(flx_ctx *c)
{
    ENTER(c, CALL_QUIET);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->regenUsed) {                                  // a look-back of the fused pass that gave up (logic.hip): fail loudly, never silently wrong pixels
        // (only when a pass with in-kernel regeneration ran since the last check: the default configuration pays no read-back here)
        uint32_t e = 0; HIPCHK(c, hipMemcpy(&e, c->logicError, 4, hipMemcpyDeviceToHost));
        c->regenUsed = false;
        if (e) {                                         // reported ONCE: the flag is cleared, the regenerated paths of that pass are wrong -- the caller resets the renderer
            HIPCHK(c, hipMemset(c->logicError, 0, 4));
            c->err = "k_logic: the in-kernel regeneration's look-back timed out (paths regenerated by that pass are invalid: reset the renderer)"; return 1;
        }
    }
    for (auto &p : c->pending) memcpy(p.user, &c->pinned[p.slot], 32);
    c->pending.clear();
    for (auto &p : c->pendingMk) memcpy(p.first, c->pinnedMk + 4 * p.second, 16);
    c->pendingMk.clear();
    for (auto &ev : c->events) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) { c->kMs[ev.kernel] += ms; c->kLaunches[ev.kernel]++; }
        c->eventPool.push_back(ev.a); c->eventPool.push_back(ev.b);
    }
    c->events.clear();
    return 0;
}

int flx_pixel_index_update(flx_ctx *c, uint32_t npix, uint32_t nnew)
{
    ENTER(c, CALL_NEUTRAL);
    NEED(c, npix > 0, "flx_pixel_index_update: zero pixels");
    HIPCHK(c, hipSetDevice(c->device));
    c->hostPixelIdx = (uint32_t)(((uint64_t)c->hostPixelIdx + nnew) % npix);
    int slot = c->nextIdxSlot; c->nextIdxSlot = (c->nextIdxSlot + 1) % c->pinnedSlots;
    c->pinnedIdx[slot] = c->hostPixelIdx;
    HIPCHK(c, hipMemcpyAsync(c->fr.currPixelIdx, &c->pinnedIdx[slot], 4, hipMemcpyHostToDevice, c->stream));
    return 0;
}
int flx_pixel_index_reset(flx_ctx *c)
{
    ENTER(c, CALL_NEUTRAL);
    HIPCHK(c, hipSetDevice(c->device));
    c->hostPixelIdx = 0;
    HIPCHK(c, hipMemsetAsync(c->fr.currPixelIdx, 0, 4, c->stream));
    return 0;
}

int flx_end_iteration_async(flx_ctx *c)
{
    READY(c, CALL_NEUTRAL);
    launch_end_iteration(c->stream, c->qs.counters, c->totals, c->fr.currPixelIdx, c->fr.localPixels, c->qs.extPend, c->qs.cursors);
    c->cursorDirty[0] = c->cursorDirty[1] = false;      // (k_end_iteration zeroes the block cursors with the counters)
    c->qs.extPend = 0;
    c->matQueuesEmpty = true; c->raygenQueueEmpty = true;      // it clears the queue counters
    LAUNCHED(c);
    return 0;
}
int flx_counter_totals(flx_ctx *c, uint64_t *out8, int reset)
{
    ENTER(c, CALL_QUIET);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out8, c->totals, 64, hipMemcpyDeviceToHost, c->stream));
    if (reset) HIPCHK(c, hipMemsetAsync(c->totals, 0, 64, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int flx_read_pixels(flx_ctx *c, int which, float *out)
{
    ENTER(c, CALL_NEUTRAL);                                       // framebuffers only
    NEED(c, c->fr.pixels && out, "flx_read_pixels: no framebuffer");
    HIPCHK(c, hipSetDevice(c->device));
    NEED(c, which >= 0 && which <= 5, "flx_read_pixels: which must be 0..5");
    const float *src[6] = {c->fr.pixels, c->fr.preview, c->fr.aovAlbedoOut, c->fr.aovNormalOut, c->fr.aovAlbedo, c->fr.aovNormal};
    NEED(c, src[which], "flx_read_pixels: the denoiser feature buffers need flx_set_option(ctx, \"denoiser\", 1)");
    HIPCHK(c, hipMemcpyAsync(out, src[which], (size_t)c->fr.localPixels * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int flx_copy_pixels_to_device(flx_ctx *c, void *dst)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, c->fr.pixels && dst, "flx_copy_pixels_to_device: no framebuffer");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, c->fr.pixels, (size_t)c->fr.localPixels * 16, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}


// ---- multi-GPU group: RCCL gather of the per-rank radiance tiles (SURVEY 8(b) flx_create_group / flx_gather, 8(e)).
// The reference is single-device (one cl::CommandQueue, src/clcontext.cpp:25-29).  Rank r renders global pixels p * R + r
// (flx_set_partition); at read-back every rank sends its compact float4[localPixels] accumulation tile to the root over
// RCCL point-to-point (grouped ncclSend / ncclRecv = a gather; xGMI links into the root work in parallel), the root
// de-interleaves into the full image.  Nothing is exchanged per iteration.
namespace {
struct Rccl {
    void *dl = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    std::string err;
};
Rccl g_rccl;
// TEST HOOK: FLX_RCCL_LIB is honoured only together with FLX_ALLOW_RCCL_OVERRIDE=1, so that a stray variable in a production
// environment cannot make the library dlopen an arbitrary path or change which gather path a local group takes.
const char *rccl_override()
{
    const char *over = getenv("FLX_RCCL_LIB"), *allow = getenv("FLX_ALLOW_RCCL_OVERRIDE");
    return (over && *over && allow && strcmp(allow, "1") == 0) ? over : nullptr;
}
bool rccl_load()
{
    if (g_rccl.dl) return true;
    // FLX_RCCL_LIB=<path> + FLX_ALLOW_RCCL_OVERRIDE=1: bind another library with the same entry points (tests/fake_rccl.cpp moves the tiles
    // between host threads on ONE device, so that the N > 1 send / receive code below runs on a 1-GPU box; never set in production)
    void *dl = nullptr;
    const char *over = rccl_override();
    if (over) {
        dl = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!dl) { g_rccl.err = std::string("FLX_RCCL_LIB=") + over + " not loadable: " + dlerror(); return false; }
    }
    if (!dl) dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!dl) dl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!dl) { g_rccl.err = std::string("librccl.so.1 not loadable: ") + dlerror(); return false; }
#define RSYM(field, name) g_rccl.field = (decltype(g_rccl.field))dlsym(dl, #name); if (!g_rccl.field) { g_rccl.err = "librccl: missing symbol " #name; dlclose(dl); return false; }
    RSYM(GetUniqueId, ncclGetUniqueId) RSYM(CommInitRank, ncclCommInitRank) RSYM(CommInitAll, ncclCommInitAll) RSYM(CommDestroy, ncclCommDestroy)
    RSYM(Send, ncclSend) RSYM(Recv, ncclRecv) RSYM(GroupStart, ncclGroupStart) RSYM(GroupEnd, ncclGroupEnd) RSYM(GetErrorString, ncclGetErrorString)
    RSYM(CommCount, ncclCommCount) RSYM(CommUserRank, ncclCommUserRank) RSYM(CommAbort, ncclCommAbort)
#undef RSYM
    g_rccl.dl = dl;
    return true;
}
}
#define NCCLCHK(c, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { (c)->err = std::string(#expr) + ": " + g_rccl.GetErrorString(r_); return 1; } } while (0)

static uint32_t tilePixels(uint32_t npix, uint32_t rank, uint32_t nranks) { return npix <= rank ? 0u : (npix - rank + nranks - 1) / nranks; }

int flx_group_unique_id(void *out128)
{
    if (!out128 || !rccl_load()) { g_create_error = out128 ? g_rccl.err : "flx_group_unique_id: null"; return 1; }
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return 1; }
    static_assert(sizeof(ncclUniqueId) == FLX_GROUP_ID_BYTES, "ncclUniqueId size");
    memcpy(out128, &id, sizeof(id));
    return 0;
}

int flx_group_destroy(flx_ctx *c)
{
    if (c->comm && g_rccl.dl) { (void)hipSetDevice(c->device); (void)g_rccl.CommDestroy(c->comm); }
    c->comm = nullptr; c->commShared = false;
    return 0;
}

static int gatherBuffers(flx_ctx *root, uint32_t nranks);

// Calls between ncclGroupStart and ncclGroupEnd: remember the first failure and keep going, so that the group is ALWAYS closed
// (returning with it open would leave every later collective of the process inside a dangling group).
struct NcclGroup {
    ncclResult_t first = ncclSuccess; const char *what = nullptr;
    void operator()(ncclResult_t r, const char *w) { if (r != ncclSuccess && first == ncclSuccess) { first = r; what = w; } }
    int fail(flx_ctx *c) const { if (first == ncclSuccess) return 0; c->err = std::string(what) + ": " + g_rccl.GetErrorString(first); return 1; }
};
#define NCCLTRY(g, expr) (g)((expr), #expr)

// a rank that cannot take part in a collective its peers have already entered tears the communicator down, so that they fail
// instead of waiting for it forever
static void abortGroup(flx_ctx *c) { if (c->comm && g_rccl.dl) { (void)hipSetDevice(c->device); (void)g_rccl.CommAbort(c->comm); } c->comm = nullptr; }

int flx_group_init(flx_ctx *c, uint32_t rank, uint32_t nranks, const void *id128)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, id128 && nranks >= 1 && rank < nranks, "flx_group_init: bad arguments");
    NEED(c, rccl_load(), g_rccl.err);
    HIPCHK(c, hipSetDevice(c->device));
    flx_group_destroy(c);
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    NCCLCHK(c, g_rccl.CommInitRank(&c->comm, (int)nranks, id, (int)rank));
    if (flx_set_partition(c, rank, nranks)) return 1;
    // Any rank may be asked to be the root of flx_gather: its staging buffers are allocated HERE, where every rank allocates the same
    // amount and an out-of-memory condition is an error of this call on every rank alike -- not inside the collective, where a root
    // that fails before posting its receives would leave the peers blocked in ncclSend.  (A later flx_set_params with a larger frame
    // re-allocates in flx_gather; if THAT fails the root aborts the communicator.)
    if (c->haveParams && gatherBuffers(c, nranks)) return 1;
    return 0;
}

int flx_group_info(flx_ctx *c, uint32_t *out2)
{
    NEED(c, out2, "flx_group_info: null");
    out2[0] = out2[1] = 0;
    if (c->commShared) { out2[0] = c->fr.nranks; out2[1] = c->fr.rank; return 0; }
    NEED(c, c->comm, "flx_group_info: no group");
    int n = 0, r = 0;
    NCCLCHK(c, g_rccl.CommCount(c->comm, &n));
    NCCLCHK(c, g_rccl.CommUserRank(c->comm, &r));
    out2[0] = (uint32_t)n; out2[1] = (uint32_t)r;
    return 0;
}

int flx_group_init_local(flx_ctx **ctxs, uint32_t n)
{
    if (!ctxs || !n || !ctxs[0]) { g_create_error = "flx_group_init_local: bad arguments"; return 1; }
    flx_ctx *c0 = ctxs[0];
    bool distinct = true;
    for (uint32_t i = 0; i < n; i++) { NEED(c0, ctxs[i], "flx_group_init_local: null context"); for (uint32_t j = 0; j < i; j++) if (ctxs[i]->device == ctxs[j]->device) distinct = false; }
    for (uint32_t i = 0; i < n; i++) { ENTER(ctxs[i], CALL_OBSERVE); flx_group_destroy(ctxs[i]); }
    // (with a stand-in transport bound through FLX_RCCL_LIB the communicator path is taken whatever the devices are: tests)
    if (rccl_override()) distinct = true;
    if (distinct) {
        NEED(c0, rccl_load(), g_rccl.err);
        std::vector<ncclComm_t> comms(n); std::vector<int> devs(n);
        for (uint32_t i = 0; i < n; i++) devs[i] = ctxs[i]->device;
        NCCLCHK(c0, g_rccl.CommInitAll(comms.data(), (int)n, devs.data()));
        for (uint32_t i = 0; i < n; i++) ctxs[i]->comm = comms[i];
    } else {
        // several contexts on one device (a 1-GPU box standing in for N ranks: tests): RCCL refuses duplicate devices, the tiles
        // travel with device-to-device copies instead; partition, staging and de-interleave are the same code
        for (uint32_t i = 0; i < n; i++) ctxs[i]->commShared = true;
    }
    for (uint32_t i = 0; i < n; i++) if (flx_set_partition(ctxs[i], i, n)) { if (ctxs[i] != c0) c0->err = ctxs[i]->err; return 1; }
    return 0;
}

static int gatherBuffers(flx_ctx *root, uint32_t nranks)
{
    const uint32_t npix = root->params.width * root->params.height;
    const size_t maxlp = tilePixels(npix, 0, nranks);
    const size_t needStage = (size_t)nranks * maxlp * 4, needFull = (size_t)npix * 4;
    if (needStage > root->gatherStageFloats || needFull > root->gatherFullFloats) {
        HIPCHK(root, hipStreamSynchronize(root->stream));
        freeAll(root->gatherAllocs);
        root->gatherStageFloats = root->gatherFullFloats = 0;
        if (dalloc(root, root->gatherAllocs, &root->gatherStage, needStage) || dalloc(root, root->gatherAllocs, &root->gatherFull, needFull)) return 1;
        root->gatherStageFloats = needStage; root->gatherFullFloats = needFull;
    }
    return 0;
}

static int gatherFinish(flx_ctx *root, uint32_t nranks, float *out_host)
{
    const uint32_t npix = root->params.width * root->params.height;
    launch_deinterleave(root->stream, root->gatherStage, root->gatherFull, npix, nranks, tilePixels(npix, 0, nranks));
    LAUNCHED(root);
    HIPCHK(root, hipMemcpyAsync(out_host, root->gatherFull, (size_t)npix * 16, hipMemcpyDeviceToHost, root->stream));
    HIPCHK(root, hipStreamSynchronize(root->stream));
    return 0;
}

// multi-process: every rank of the communicator calls this; out_host (width*height float4) is written on `root` only.
// Error paths: argument errors that every rank sees alike (no group, no frame, root out of range) return before anything is posted.
// Past that point the peers are, or soon will be, blocked in their ncclSend, so the root either posts every matching receive
// (also when its own output pointer is null: the tiles are received and the error reported afterwards) or aborts the communicator.
int flx_gather(flx_ctx *c, uint32_t root, float *out_host)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, c->comm, "flx_gather: no group (flx_group_init first)");
    NEED(c, c->fr.pixels && c->haveParams, "flx_gather: no framebuffer");
    const uint32_t R = c->fr.nranks, me = c->fr.rank, npix = c->params.width * c->params.height;
    NEED(c, root < R, "flx_gather: bad root");
    HIPCHK(c, hipSetDevice(c->device));
    if (me != root) {
        NcclGroup g;
        NCCLTRY(g, g_rccl.GroupStart());
        NCCLTRY(g, g_rccl.Send(c->fr.pixels, (size_t)tilePixels(npix, me, R) * 4, ncclFloat32, (int)root, c->comm, c->stream));
        NCCLTRY(g, g_rccl.GroupEnd());
        if (g.fail(c)) return 1;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    if (gatherBuffers(c, R)) { const std::string why = c->err; abortGroup(c); c->err = "flx_gather: root cannot allocate its staging buffers (" + why + "); communicator aborted"; return 1; }
    const size_t maxlp = tilePixels(npix, 0, R);
    NcclGroup g;
    NCCLTRY(g, g_rccl.GroupStart());
    for (uint32_t r = 0; r < R; r++) {
        if (r == me) continue;
        NCCLTRY(g, g_rccl.Recv(c->gatherStage + (size_t)r * maxlp * 4, (size_t)tilePixels(npix, r, R) * 4, ncclFloat32, (int)r, c->comm, c->stream));
    }
    NCCLTRY(g, g_rccl.GroupEnd());
    if (g.fail(c)) return 1;
    HIPCHK(c, hipMemcpyAsync(c->gatherStage + (size_t)me * maxlp * 4, c->fr.pixels, (size_t)c->fr.localPixels * 16, hipMemcpyDeviceToDevice, c->stream));
    if (!out_host) { HIPCHK(c, hipStreamSynchronize(c->stream)); c->err = "flx_gather: null output on the root (the tiles were received and dropped)"; return 1; }
    return gatherFinish(c, R, out_host);
}

// single process: the n contexts of flx_group_init_local, driven by one host thread
int flx_gather_local(flx_ctx **ctxs, uint32_t n, uint32_t root, float *out_host)
{
    if (!ctxs || !n || root >= n || !ctxs[root]) { g_create_error = "flx_gather_local: bad arguments"; return 1; }
    flx_ctx *rc = ctxs[root];
    NEED(rc, out_host, "flx_gather_local: null output");
    for (uint32_t i = 0; i < n; i++) {
        ENTER(ctxs[i], CALL_OBSERVE);
        NEED(rc, ctxs[i]->fr.nranks == n && ctxs[i]->fr.rank == i && ctxs[i]->fr.pixels && ctxs[i]->haveParams, "flx_gather_local: contexts are not the group of flx_group_init_local");
        NEED(rc, (ctxs[i]->comm != nullptr) != ctxs[i]->commShared, "flx_gather_local: no group (flx_group_init_local first)");
    }
    HIPCHK(rc, hipSetDevice(rc->device));
    if (gatherBuffers(rc, n)) return 1;                                 // nothing posted yet: a plain error
    const uint32_t npix = rc->params.width * rc->params.height;
    const size_t maxlp = tilePixels(npix, 0, n);
    if (rc->comm) {
        NcclGroup g; hipError_t he = hipSuccess;
        NCCLTRY(g, g_rccl.GroupStart());
        for (uint32_t r = 0; r < n && he == hipSuccess; r++) {
            if (r == root) continue;
            if ((he = hipSetDevice(ctxs[r]->device)) != hipSuccess) break;
            NCCLTRY(g, g_rccl.Send(ctxs[r]->fr.pixels, (size_t)ctxs[r]->fr.localPixels * 4, ncclFloat32, (int)root, ctxs[r]->comm, ctxs[r]->stream));
            if ((he = hipSetDevice(rc->device)) != hipSuccess) break;
            NCCLTRY(g, g_rccl.Recv(rc->gatherStage + (size_t)r * maxlp * 4, (size_t)ctxs[r]->fr.localPixels * 4, ncclFloat32, (int)r, rc->comm, rc->stream));
        }
        NCCLTRY(g, g_rccl.GroupEnd());                                  // always closed, whatever happened above
        (void)hipSetDevice(rc->device);
        if (he != hipSuccess) { rc->err = std::string("flx_gather_local: hipSetDevice: ") + hipGetErrorString(he); return 1; }
        if (g.fail(rc)) return 1;
        for (uint32_t r = 0; r < n; r++) if (r != root) { HIPCHK(rc, hipSetDevice(ctxs[r]->device)); HIPCHK(rc, hipStreamSynchronize(ctxs[r]->stream)); }
        HIPCHK(rc, hipSetDevice(rc->device));
    } else {
        for (uint32_t r = 0; r < n; r++) {
            if (r == root) continue;
            HIPCHK(rc, hipStreamSynchronize(ctxs[r]->stream));           // the tile is complete
            HIPCHK(rc, hipMemcpyAsync(rc->gatherStage + (size_t)r * maxlp * 4, ctxs[r]->fr.pixels, (size_t)ctxs[r]->fr.localPixels * 16, hipMemcpyDeviceToDevice, rc->stream));
        }
    }
    HIPCHK(rc, hipMemcpyAsync(rc->gatherStage + (size_t)root * maxlp * 4, rc->fr.pixels, (size_t)rc->fr.localPixels * 16, hipMemcpyDeviceToDevice, rc->stream));
    return gatherFinish(rc, n, out_host);
}

// ---- measurement
int flx_profile_enable(flx_ctx *c, int on) { ENTER(c, CALL_QUIET); c->profile = on < 0 ? 0 : on > 4 ? 1 : on; return 0; }
int flx_profile_get(flx_ctx *c, int k, double *ms, uint64_t *n) { NEED(c, k >= 0 && k < FLX_K_COUNT, "bad kernel id"); *ms = c->kMs[k]; *n = c->kLaunches[k]; return 0; }
int flx_profile_reset(flx_ctx *c) { for (int k = 0; k < FLX_K_COUNT; k++) { c->kMs[k] = 0; c->kLaunches[k] = 0; } return 0; }
int flx_trace_stats_enable(flx_ctx *c, int on) { ENTER(c, CALL_PEEK); c->statsOn = on != 0; return 0; }
int flx_trace_stats_get(flx_ctx *c, uint64_t *out7)
{
    ENTER(c, CALL_PEEK);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out7, c->stats, 56, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int flx_trace_stats_get_ex(flx_ctx *c, uint64_t *out16)
{
    ENTER(c, CALL_PEEK);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out16, c->stats, 128, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int flx_trace_stats_get_all(flx_ctx *c, uint64_t *out24)
{
    ENTER(c, CALL_PEEK);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out24, c->stats, FLX_NUM_TRACE_STATS * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int flx_trace_stats_reset(flx_ctx *c) { ENTER(c, CALL_OBSERVE); HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipMemsetAsync(c->stats, 0, FLX_NUM_TRACE_STATS * 8, c->stream)); return 0; }
int flx_scene_info(flx_ctx *c, uint32_t *out8) { NEED(c, out8, "flx_scene_info: null"); memcpy(out8, c->wideInfo, 32); return 0; }

// ---- test hooks
int flx_state_export(flx_ctx *c, float *out)
{
    ENTER(c, CALL_OBSERVE);
    HIPCHK(c, hipSetDevice(c->device));
    float *d = nullptr; size_t bytes = (size_t)FLX_NUM_COLS * c->numTasks * 4;
    HIPCHK(c, hipMalloc((void **)&d, bytes));
    launch_state_export(c->stream, c->st, d, c->haveParams ? 2.0f * c->params.worldRadius : 0.0f);
    hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(c, e);
    return 0;
}
int flx_state_import(flx_ctx *c, const float *in)
{
    ENTER(c, CALL_OBSERVE);
    HIPCHK(c, hipSetDevice(c->device));
    float *d = nullptr; size_t bytes = (size_t)FLX_NUM_COLS * c->numTasks * 4;
    HIPCHK(c, hipMalloc((void **)&d, bytes));
    hipError_t e = hipMemcpyAsync(d, in, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { launch_state_import(c->stream, c->st, d); e = hipStreamSynchronize(c->stream); }
    (void)hipFree(d);
    HIPCHK(c, e);
    return 0;
}
int flx_env_sample_table(flx_ctx *c, float *out)
{
    ENTER(c, CALL_QUIET);
    NEED(c, out, "flx_env_sample_table: null");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->sc.neeRec, (size_t)c->sc.envW * c->sc.envH * 32, hipMemcpyDeviceToHost));
    return 0;
}
int flx_math_probe(flx_ctx *c, int fn, const float *a, const float *b, uint32_t n, uint32_t *out_bits)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, a && b && out_bits && n && fn >= 0 && fn <= 15, "flx_math_probe: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    float *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, (size_t)n * 12));
    hipError_t e = hipMemcpyAsync(d, a, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + n, b, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { launch_math_probe(c->stream, fn, d, d + n, n, reinterpret_cast<uint32_t *>(d + 2 * (size_t)n)); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out_bits, d + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(c, e);
    return 0;
}
int flx_queue_read(flx_ctx *c, int q, uint32_t *out)
{
    NEED(c, q >= 0 && q < FLX_NUM_QUEUES, "bad queue id");
    ENTER(c, CALL_PEEK);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->qs.q[q], (size_t)c->numTasks * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int flx_queue_write(flx_ctx *c, int q, const uint32_t *in, uint32_t n)
{
    ENTER(c, CALL_OBSERVE);
    NEED(c, q >= 0 && q < FLX_NUM_QUEUES && n <= c->numTasks, "bad queue id / length");
    HIPCHK(c, hipSetDevice(c->device));
    if (n) HIPCHK(c, hipMemcpyAsync(c->qs.q[q], in, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int flx_set_counters(flx_ctx *c, const void *in32)
{
    ENTER(c, CALL_OBSERVE);
    c->qs.extPend = 0;                                  // the caller's counters are complete
    c->matQueuesEmpty = false; c->raygenQueueEmpty = false;   // ... and unknown here
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->qs.counters, in32, 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int flx_set_option(flx_ctx *c, const char *name, int value)
{
    ENTER(c, CALL_PEEK);
    if (name && strcmp(name, "xcd_remap") == 0) { c->xcdRemap = value; return 0; }
    if (name && strcmp(name, "fuse") == 0 && (value == 0 || value == 1)) { c->fuse = value; return 0; }
    if (name && strcmp(name, "ext_order") == 0 && value >= 0 && value <= 2) { c->extOrder = value; return 0; }
    if (name && strcmp(name, "regen") == 0 && (value == 0 || value == 1)) { if (value && optionBuffers(c, false, true)) return 1; c->regenOpt = value; return 0; }
    if (name && strcmp(name, "regen_prep") == 0 && (value == 0 || value == 1)) { c->prepOpt = value; return 0; }
    if (name && strcmp(name, "regroup") == 0 && value >= -1 && value <= 1) { c->regroupOpt = value; c->regroup = value >= 0 ? value : c->regroupAuto; return 0; }
    if (name && strcmp(name, "fuse_set") == 0 && (value == 1 || value == 31)) { c->fuseSet = value; return 0; }
    if (name && strcmp(name, "overlap") == 0 && value >= -1 && value <= 2) { ENTER(c, CALL_OBSERVE); c->overlapOpt = value; pickSchedule(c); return 0; }
    if (name && strcmp(name, "shadow_tree") == 0 && (value == 2 || value == 4)) { ENTER(c, CALL_OBSERVE); c->shadowTree = value; return 0; }
    if (name && strcmp(name, "extend_tree") == 0 && (value == 2 || value == 4)) { ENTER(c, CALL_OBSERVE); c->extendTree = value; return 0; }
    if (name && strcmp(name, "denoiser") == 0 && (value == 0 || value == 1)) {
        ENTER(c, CALL_OBSERVE);
        if (c->denoiser != value) { c->denoiser = value; HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipStreamSynchronize(c->stream)); return allocAov(c); }
        return 0;
    }
    if (name && strcmp(name, "refill_extend") == 0 && refill_value_ok(value)) { ENTER(c, CALL_OBSERVE); c->refillExt = value; return 0; }
    if (name && strcmp(name, "refill_shadow") == 0 && (value == -1 || refill_value_ok(value))) { ENTER(c, CALL_OBSERVE); c->refillShadowOpt = value; pickSchedule(c); return 0; }
    if (name && (strcmp(name, "refill_extend") == 0 || strcmp(name, "refill_shadow") == 0)) { c->err = "flx_set_option: refill value must be 0 or refillMin (1..64) | waitMax (0..64) << 8"; return 1; }
    if (name && strcmp(name, "shadow_split") == 0 && value >= 0 && (value & 0xFF) <= 255 && (value >> 8) <= 255 && ((value & 0xFF) > 0 || value == 0)) { ENTER(c, CALL_OBSERVE); if (value && optionBuffers(c, true, false)) return 1; c->shadowSplit = value; return 0; }
    if (name && strcmp(name, "shadow_split_limit") == 0 && value >= 0) { ENTER(c, CALL_OBSERVE); c->splitLimit = (uint32_t)value; return 0; }
    if (name && strcmp(name, "eager_bump") == 0 && (value == 0 || value == 1)) { c->eagerBump = value; return 0; }
    if (name && strcmp(name, "node_layout") == 0 && (value == 0 || value == 1)) { c->nodeLayout = value; return 0; }
    c->err = std::string("flx_set_option: unknown option ") + (name ? name : "(null)");
    return 1;
}
// the state machine's state as one number (read-only option "phase"; tests/test_gpu_fuzz.py reports which (phase, call) pairs it exercised):
// bits 0-2 enum Phase, bit 3 RAW hit records pending, bit 4 material queues known empty
static int phaseCode(const flx_ctx *c)
{
    return c->phase | (c->rawHits ? 8 : 0) | (c->matQueuesEmpty ? 16 : 0);
}
int flx_get_option(flx_ctx *c, const char *name, int *value)
{
    NEED(c, name && value, "flx_get_option: null");
    const struct { const char *n; int v; } tab[] = {
        {"xcd_remap", c->xcdRemap}, {"fuse", c->fuse}, {"overlap", c->overlap}, {"shadow_tree", c->shadowTree}, {"extend_tree", c->extendTree},
        {"denoiser", c->denoiser}, {"eager_bump", c->eagerBump}, {"node_layout", c->nodeLayout}, {"fuse_set", c->fuseSet}, {"ext_order", c->extOrder}, {"regen", c->regenOpt}, {"regroup", c->regroup}, {"regen_prep", c->prepOpt}, {"refill_extend", c->refillExt}, {"refill_shadow", c->refillShadow}, {"shadow_split", c->shadowSplit}, {"fused_queue_mask", (int)fused_queue_mask(fuseSetNow(c))}, {"fuse_set_now", fuseSetNow(c)}};
    for (const auto &t : tab) if (strcmp(name, t.n) == 0) { *value = t.v; return 0; }
    if (strcmp(name, "phase") == 0) { *value = phaseCode(c); return 0; }
    c->err = std::string("flx_get_option: unknown option ") + name;
    return 1;
}

} // extern "C"
