// hipcontext.hpp -- C++ device context over the C ABI of libfluctus_hip.so.
//
// Same method surface as the reference's `class CLContext` (reference: src/clcontext.hpp:31-79) for the
// wavefront path and the microkernel integrator, so the Tracer loop reads like the reference's; every method forwards to one flx_* entry
// point (include/fluctus_hip.h).  The HIP library is bound with dlopen at construction: libfluctus_host.so
// itself has no GPU dependency and there is NO CPU fallback -- a missing library or device throws.
#pragma once
#include <string>
#include <vector>
#include <cstdint>
#include "../../include/fluctus_hip.h"
#include "scene.hpp"
#include "bvh.hpp"
#include "envmap.hpp"

namespace fluctus {

typedef flx_render_params RenderParams;
typedef flx_queue_counters QueueCounters;
struct PerfNumbers { double primary = 0, extension = 0, shadow = 0, samples = 0, total = 0; };   // src/clcontext.hpp:19-24
struct RenderStats { uint64_t primaryRays = 0, extensionRays = 0, shadowRays = 0, samples = 0; }; // src/geom.h:254-260 (64-bit here)

class HipContext {
public:
    HipContext(int device, uint32_t numTasks, const std::string &libPath = "");
    ~HipContext();
    HipContext(const HipContext &) = delete;

    void uploadSceneData(BVH *bvh, Scene *scene);                 // src/clcontext.cpp:522-566
    void createEnvMap(EnvironmentMap *map);                       // src/clcontext.cpp:467-511
    void updateParams(const RenderParams &params);                // src/clcontext.cpp:703-707
    void enqueueWfResetKernel(const RenderParams &params);        // src/clcontext.cpp:765-770
    void enqueueWfRaygenKernel(const RenderParams &params);
    void enqueueWfExtRayKernel(const RenderParams &params);
    void enqueueWfShadowRayKernel(const RenderParams &params);
    void enqueueWfLogicKernel(const RenderParams &params, bool firstIteration);
    void enqueueWfMaterialKernels(const RenderParams &params);
    // microkernel integrator (src/clcontext.hpp:45-51)
    void enqueueResetKernel(const RenderParams &params);
    void enqueueRayGenKernel(const RenderParams &params);
    void enqueueNextVertexKernel(const RenderParams &params);
    void enqueueBsdfSampleKernel(const RenderParams &params);
    void enqueueSplatKernel(const RenderParams &params);
    void enqueueSplatPreviewKernel(const RenderParams &params);
    void fetchStatsAsync();                                       // src/clcontext.cpp:642-646; folded into statsAsync by finishQueue()
    void recompileKernels(bool useDenoiser);                      // src/clcontext.cpp:852-874: here only the denoiser-feature switch
    // kernel selection the reference makes through Settings + recompile (src/settings.cpp): "extend_tree" / "shadow_tree" 2 | 4, ...
    // (include/fluctus_hip.h, flx_set_option)
    void setOption(const std::string &name, int value);
    int getOption(const std::string &name);                    // flx_get_option: current value, e.g. what the upload picked for "fuse_set"
    void enqueueClearWfQueues();                                  // src/clcontext.cpp:877-883
    void enqueueGetCounters(QueueCounters *cnt);                  // async; valid after finishQueue()
    void enqueuePostprocessKernel(const RenderParams &params);
    void finishQueue();
    void updatePixelIndex(uint32_t numPixels, uint32_t numNewPaths);
    void resetPixelIndex();
    uint32_t getNumTasks() const;
    void setPartition(uint32_t rank, uint32_t nranks);
    // multi-GPU from ONE process (no counterpart in the reference, one cl::CommandQueue): the contexts become ranks 0..n-1 of an RCCL
    // group with the pixel-interleaved partition; gatherLocal collects the accumulation tiles on `root` (flx_group_init_local / flx_gather_local)
    static void groupInitLocal(const std::vector<HipContext *> &ranks);
    static void gatherLocal(const std::vector<HipContext *> &ranks, uint32_t root, std::vector<float> &rgba);
    uint32_t localPixels() const;

    // image export: raw accumulation (.pfm, float RGB = sum/count) or tonemapped preview (.ppm)
    // (reference: CLContext::saveImage via DevIL, src/clcontext.cpp:386-465)
    void saveImage(const std::string &filename, const RenderParams &params);
    void readPixels(int which, std::vector<float> &rgba);

    void updateRenderPerf(float deltaT);                          // src/clcontext.cpp:648-656
    const PerfNumbers getRenderPerf() const { return renderPerf; }
    const RenderStats getStats() const { return statsAsync; }
    void resetStats() { statsAsync = RenderStats(); }
    RenderStats statsAsync;                                       // Tracer adds queue lengths here (src/tracer.cpp:336-339)

    flx_ctx *raw() { return ctx; }

private:
    void check(int rc, const char *what);
    void foldMkStats();
    RenderParams lastParams {};
    uint32_t mkStats[4] = {0, 0, 0, 0};
    bool mkPending = false;
    void *dl = nullptr;
    flx_ctx *ctx = nullptr;
    PerfNumbers renderPerf;
    struct Api;
    Api *api = nullptr;
};

} // namespace fluctus
