// tracer.hpp -- headless render driver: the host loop that sequences the wavefront kernels.
//
// Restates the wavefront branch of the reference's Tracer (reference: src/tracer.hpp:31-43,
// src/tracer.cpp): resetParams (:38-52), init (:55-80), initHierarchy (:574-590, BVH cache keyed by a hash of
// the mesh), update() WF branch (:222-266, :302-340) and runBenchmark() WF body (:362-528, CSV schema
// `scene;time;primary;extension;shadow;total;samples` :393), plus the microkernel integrator's branches of the same
// functions and renderSingle (:95-187).  No window, no GL, no denoiser.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "hipcontext.hpp"

namespace fluctus {

class Tracer {
public:
    Tracer(int width, int height, int device = 0, uint32_t numTasks = 1u << 20);
    // one process driving several GPUs of a node: devices[0] is the root; every device runs the complete wavefront loop
    // on its own interleaved pixel subset with its own numTasks paths, scene replicated; tiles are gathered over RCCL for read-back.
    // The same device may be listed more than once (1-GPU stand-in for N ranks).
    Tracer(int width, int height, const std::vector<int> &devices, uint32_t numTasks);
    uint32_t numRanks() const { return 1u + (uint32_t)peers.size(); }
    // full-resolution accumulation image (rgb sum, sample count) assembled from all ranks
    void readAccumulation(std::vector<float> &rgba);
    ~Tracer();

    void init(int width, int height, const std::string &sceneFile);             // file path or "proc:<kind>:<tris>:<seed>"
    void setEnvMap(const std::string &hdrFile);                                   // Tracer::initEnvMap
    void update();                                                                // one frame (iteration 0 = 2-bounce preview x3)
    // benchmark-style iterations for `seconds` (reference: 30 s per scene) or exactly `iterations` if > 0;
    // returns the CSV text (header + one row per 0.5 s of wall time)
    std::string runBenchmark(double seconds, int iterations = 0);
    // final-frame render: exactly `spp` samples in every pixel (reference: src/tracer.cpp:95-187).  Switches to the
    // microkernel integrator and turns Russian roulette off, as the reference does; needs numTasks >= width*height.
    void renderSingle(int spp, bool denoise = false);                             // denoise: also fill the denoiser feature buffers
    void setDenoiser(bool on) { useDenoiser = on; for (auto *c : ranks()) c->recompileKernels(on); iteration = 0; }
    void toggleRenderer() { useWavefront = !useWavefront; iteration = 0; }        // src/tracer.cpp:881-886
    void setOption(const std::string &name, int value) { for (auto *c : ranks()) c->setOption(name, value); }   // every rank (HipContext::setOption)
    bool usesWavefront() const { return useWavefront; }
    void saveImage(const std::string &filename) { clctx->saveImage(filename, params); }

    RenderParams &getParams() { return params; }
    void paramsChanged() { paramsUpdatePending = true; }
    HipContext *getContext() { return clctx.get(); }
    Scene *getScene() { return scene.get(); }
    uint32_t getIteration() const { return iteration; }
    const QueueCounters &lastCounters() const { return lastCnt; }
    // on-disk caches in the reference's formats and names (src/tracer.cpp:573-590, 625-684): <dir>/hierarchy_<hash>.bin and
    // <dir>/state_<hash>.dat, hash = XXH64 of the scene file in decimal (of the triangle array for procedural scenes)
    std::string hierarchyCacheDir = "";                                           // empty = no on-disk BVH cache ("data/hierarchies" in the reference)
    std::string stateDir = "";                                                    // "data/states" in the reference
    bool saveState() const;                                                       // Tracer::saveState: camera, lights, sampling and post-processing parameters
    bool loadState();                                                             // false when there is no (complete) state file for this scene
    const std::string &getSceneHash() const { return sceneHash; }
    float cameraRotation[2] = {0.0f, 0.0f};                                       // UI state the file carries (src/tracer.hpp: cameraRotation, cameraSpeed)
    float cameraSpeed = 1.0f;

private:
    void resetParams(int width, int height);
    void initCamera();
    void initPostProcessing();
    void initAreaLight();
    void initHierarchy();
    void updateMicrokernel();

    RenderParams params;
    std::unique_ptr<Scene> scene;
    std::unique_ptr<EnvironmentMap> envMap;
    std::unique_ptr<HipContext> clctx;                                            // rank 0
    std::vector<std::unique_ptr<HipContext>> peers;                               // ranks 1..R-1 (multi-GPU wavefront path)
    std::vector<HipContext *> ranks() { std::vector<HipContext *> r{clctx.get()}; for (auto &p : peers) r.push_back(p.get()); return r; }
    BVH *bvh = nullptr;
    uint32_t iteration = 0;
    bool paramsUpdatePending = true;
    bool useDenoiser = false;                                                     // feature buffers only; the OptiX denoiser itself is out of scope
    bool useWavefront = true;                                                     // this library's default; the reference starts on MK (src/tracer.cpp:11)
    QueueCounters lastCnt {};
    std::string sceneName;
    std::string sceneHash;
};

} // namespace fluctus
