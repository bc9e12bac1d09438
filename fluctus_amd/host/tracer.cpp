// tracer.cpp -- see tracer.hpp.
#include "tracer.hpp"
#include <chrono>
#include <sstream>
#include <fstream>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <stdexcept>

namespace fluctus {

static flx_vec3 v3(float x, float y, float z, float w = 0.0f) { return flx_vec3{x, y, z, w}; }

Tracer::Tracer(int width, int height, int device, uint32_t numTasks)
{
    std::memset(&params, 0, sizeof(params));
    resetParams(width, height);
    initCamera();
    initPostProcessing();
    initAreaLight();
    scene.reset(new Scene());
    clctx.reset(new HipContext(device, numTasks));
}

Tracer::Tracer(int width, int height, const std::vector<int> &devices, uint32_t numTasks) : Tracer(width, height, devices.empty() ? 0 : devices[0], numTasks)
{
    for (size_t i = 1; i < devices.size(); i++) peers.emplace_back(new HipContext(devices[i], numTasks));
    if (!peers.empty()) HipContext::groupInitLocal(ranks());
}

Tracer::~Tracer() { delete bvh; }

void Tracer::readAccumulation(std::vector<float> &rgba)
{
    if (peers.empty()) { clctx->readPixels(0, rgba); return; }
    HipContext::gatherLocal(ranks(), 0, rgba);
}

// reference: src/tracer.cpp:38-52
void Tracer::resetParams(int width, int height)
{
    params.width = (uint32_t)width; params.height = (uint32_t)height;
    params.useEnvMap = 0; params.useAreaLight = 1; params.envMapStrength = 1.0f; params.maxBounces = 10;
    params.sampleImpl = 1; params.sampleExpl = 1; params.useRoulette = 0; params.wfSeparateQueues = 0;
}
// reference: src/tracer.cpp:760-776
void Tracer::initCamera()
{
    flx_camera &c = params.camera;
    c.pos = v3(0.0f, 1.0f, 3.5f); c.right = v3(1.0f, 0.0f, 0.0f); c.up = v3(0.0f, 1.0f, 0.0f); c.dir = v3(0.0f, 0.0f, -1.0f);
    c.fov = 60.0f; c.apertureSize = 0.0f; c.focalDist = 0.5f;
    paramsUpdatePending = true;
}
// reference: src/tracer.cpp:778-786
void Tracer::initPostProcessing() { params.exposure = 1.0f; params.tmOperator = 2; paramsUpdatePending = true; }
// reference: src/tracer.cpp:788-797
void Tracer::initAreaLight()
{
    flx_arealight &l = params.areaLight;
    l.E = v3(200.0f, 200.0f, 200.0f); l.right = v3(0.0f, 0.0f, -1.0f); l.up = v3(0.0f, 1.0f, 0.0f);
    l.N = v3(-1.0f, 0.0f, 0.0f, 0.0f); l.pos = v3(1.0f, 1.0f, 0.0f, 1.0f); l.size.x = 0.5f; l.size.y = 0.5f;
    paramsUpdatePending = true;
}

// reference: src/tracer.cpp:573-590 -- cached hierarchy if present, else SBVH (always SplitMode::SAH -> SBVH) + export
void Tracer::initHierarchy()
{
    delete bvh; bvh = new BVH();
    auto &tris = scene->getTriangles();
    params.n_tris = (uint32_t)tris.size();
    std::string cache;
    if (!hierarchyCacheDir.empty()) {
        cache = hierarchyCacheDir + "/hierarchy_" + sceneHash + ".bin";
        if (bvh->importFrom(cache)) return;
    }
    bvh->build(&tris, BVH::Mode::SBVH);
    if (!cache.empty()) bvh->exportTo(cache);
}

// reference: src/tracer.cpp:55-80
void Tracer::init(int width, int height, const std::string &sceneFile)
{
    resetParams(width, height);
    scene.reset(new Scene());
    sceneName = sceneFile;
    if (sceneFile.compare(0, 5, "proc:") == 0) {
        std::string kind; uint32_t tris = 100000, seed = 42;
        std::istringstream ss(sceneFile.substr(5)); std::string tok;
        if (std::getline(ss, tok, ':')) kind = tok;
        if (std::getline(ss, tok, ':')) tris = (uint32_t)std::stoul(tok);
        if (std::getline(ss, tok, ':')) seed = (uint32_t)std::stoul(tok);
        scene->generate(kind, tris, seed);
        sceneHash = std::to_string(xxh64(scene->getTriangles().data(), scene->getTriangles().size() * sizeof(flx_triangle)));   // no file to hash
    } else {
        scene->loadModel(sceneFile);
        sceneHash = std::to_string(fileHash(sceneFile));                 // Scene::hashString (src/scene.cpp:46-51, 95)
    }
    initHierarchy();
    params.worldRadius = bvh->worldRadius();                 // :66-67
    for (auto *c : ranks()) c->uploadSceneData(bvh, scene.get());    // replicated: read-only, 288 GB per GPU
    delete bvh; bvh = nullptr;                               // :72-73 data lives on the GPU now
    paramsUpdatePending = true;
    iteration = 0;
}

void Tracer::setEnvMap(const std::string &hdrFile)
{
    envMap.reset(new EnvironmentMap(hdrFile));
    for (auto *c : ranks()) c->createEnvMap(envMap.get());
    params.useEnvMap = 1;
    paramsUpdatePending = true;
}

// reference: src/tracer.cpp:625-684 (iterateStateItems): one item after the other in native byte order
namespace {
struct StateIO {
    std::fstream f; bool write;
    template <class T> void rw(T &v) { if (write) f.write((const char *)&v, sizeof(T)); else f.read((char *)&v, sizeof(T)); }
    void vec(flx_vec3 &v) { rw(v.x); rw(v.y); rw(v.z); }
};
}
static bool stateItems(const std::string &path, bool write, RenderParams &p, float *cameraRotation, float &cameraSpeed)
{
    StateIO io; io.write = write;
    io.f.open(path, std::ios::binary | (write ? std::ios::out : std::ios::in));
    if (!io.f.good()) return false;
    io.rw(cameraRotation[0]); io.rw(cameraRotation[1]); io.rw(cameraSpeed);
    io.rw(p.camera.fov); io.rw(p.camera.focalDist); io.rw(p.camera.apertureSize);
    io.vec(p.camera.dir); io.vec(p.camera.pos); io.vec(p.camera.right); io.vec(p.camera.up);
    io.vec(p.areaLight.N); io.vec(p.areaLight.pos); io.vec(p.areaLight.right); io.vec(p.areaLight.up); io.vec(p.areaLight.E);
    io.rw(p.areaLight.size.x); io.rw(p.areaLight.size.y); io.rw(p.envMapStrength);
    io.rw(p.maxBounces); io.rw(p.useAreaLight); io.rw(p.useEnvMap); io.rw(p.sampleExpl); io.rw(p.sampleImpl); io.rw(p.useRoulette);
    io.rw(p.exposure); io.rw(p.tmOperator);
    return io.f.good();
}
bool Tracer::saveState() const
{
    if (stateDir.empty()) return false;
    RenderParams p = params; float rot[2] = {cameraRotation[0], cameraRotation[1]}; float speed = cameraSpeed;
    return stateItems(stateDir + "/state_" + sceneHash + ".dat", true, p, rot, speed);
}
bool Tracer::loadState()
{
    if (stateDir.empty()) return false;
    RenderParams p = params; float rot[2] = {0.0f, 0.0f}; float speed = 1.0f;
    if (!stateItems(stateDir + "/state_" + sceneHash + ".dat", false, p, rot, speed)) return false;
    params = p; cameraRotation[0] = rot[0]; cameraRotation[1] = rot[1]; cameraSpeed = speed;
    paramsUpdatePending = true;
    return true;
}

// reference: src/tracer.cpp:95-187
void Tracer::renderSingle(int spp, bool denoise)
{
    if (!peers.empty()) throw std::runtime_error("renderSingle: the microkernel integrator is single-GPU");
    if (useWavefront) toggleRenderer();                                  // only MK guarantees the spp of every pixel (:99-101)
    if ((uint64_t)params.width * params.height > clctx->getNumTasks())
        throw std::runtime_error("renderSingle: width*height exceeds the context's numTasks (one path per pixel)");
    params.useRoulette = 0;                                              // :104-108
    if (denoise) setDenoiser(true);                                      // :110-114
    clctx->updateParams(params); paramsUpdatePending = false;
    clctx->enqueueResetKernel(params);
    for (int sample = 0; sample < spp; sample++) {
        clctx->enqueueRayGenKernel(params);
        for (uint32_t bounce = 0; bounce < params.maxBounces + 1; bounce++) {
            clctx->enqueueNextVertexKernel(params);
            clctx->enqueueBsdfSampleKernel(params);
        }
        clctx->enqueueSplatKernel(params);
        clctx->enqueuePostprocessKernel(params);
        clctx->fetchStatsAsync();
        clctx->finishQueue();
        iteration++;
    }
}

// reference: src/tracer.cpp:268-299, microkernel branch of update()
void Tracer::updateMicrokernel()
{
    if (iteration == 0) {                                                // interactive preview: two segments, splat incomplete paths
        clctx->enqueueResetKernel(params);
        clctx->enqueueRayGenKernel(params);
        clctx->enqueueNextVertexKernel(params);
        clctx->enqueueBsdfSampleKernel(params);
        clctx->enqueueNextVertexKernel(params);
        clctx->enqueueBsdfSampleKernel(params);
        clctx->enqueueSplatPreviewKernel(params);
    } else {                                                             // one state-machine step of every path per frame
        clctx->enqueueRayGenKernel(params);
        clctx->enqueueNextVertexKernel(params);
        clctx->enqueueBsdfSampleKernel(params);
        clctx->enqueueSplatKernel(params);
    }
    clctx->enqueuePostprocessKernel(params);
    clctx->fetchStatsAsync();                                            // :343-344
    clctx->finishQueue();
    iteration++;
}

// reference: src/tracer.cpp:189-358.  With several ranks every enqueue* call fans out over the devices (all asynchronous: one host
// thread keeps them busy); each rank has its own counters and pixel cursor.
void Tracer::update()
{
    auto R = ranks();
    if (paramsUpdatePending) { for (auto *c : R) c->updateParams(params); paramsUpdatePending = false; iteration = 0; }
    if (!useWavefront) {
        if (!peers.empty()) throw std::runtime_error("update: the microkernel integrator is single-GPU");
        updateMicrokernel(); return;
    }
    std::vector<QueueCounters> cnt(R.size());
    std::memset(cnt.data(), 0, cnt.size() * sizeof(QueueCounters));
    uint32_t maxBounces = params.maxBounces;
    int N = 1;
    if (iteration == 0) {
        params.maxBounces = std::min((uint32_t)2, maxBounces);           // 2-bounce preview
        N = 3;
        for (auto *c : R) {
            c->updateParams(params);
            c->resetPixelIndex();
            c->enqueueWfResetKernel(params);
            c->enqueueWfRaygenKernel(params);
            c->enqueueWfExtRayKernel(params);
            c->enqueueClearWfQueues();
        }
    }
    for (int i = 0; i < N; i++) {
        for (size_t r = 0; r < R.size(); r++) {
            HipContext *c = R[r];
            c->enqueueWfLogicKernel(params, iteration == 0);
            c->enqueueWfRaygenKernel(params);
            c->enqueueWfMaterialKernels(params);
            c->enqueueGetCounters(&cnt[r]);
            c->enqueueWfExtRayKernel(params);
            c->enqueueWfShadowRayKernel(params);
            c->enqueueClearWfQueues();
        }
    }
    if (iteration == 0) { params.maxBounces = maxBounces; for (auto *c : R) c->updateParams(params); }
    for (auto *c : R) c->enqueuePostprocessKernel(params);
    for (auto *c : R) c->finishQueue();
    QueueCounters sum; std::memset(&sum, 0, sizeof(sum));
    for (size_t r = 0; r < R.size(); r++) {
        R[r]->updatePixelIndex(R[r]->localPixels(), cnt[r].raygenQueue);
        sum.raygenQueue += cnt[r].raygenQueue; sum.extensionQueue += cnt[r].extensionQueue; sum.shadowQueue += cnt[r].shadowQueue;
        sum.diffuseQueue += cnt[r].diffuseQueue; sum.glossyQueue += cnt[r].glossyQueue; sum.ggxReflQueue += cnt[r].ggxReflQueue;
        sum.ggxRefrQueue += cnt[r].ggxRefrQueue; sum.deltaQueue += cnt[r].deltaQueue;
    }
    clctx->statsAsync.extensionRays += sum.extensionQueue;               // :336-339
    clctx->statsAsync.shadowRays += sum.shadowQueue;
    clctx->statsAsync.primaryRays += sum.raygenQueue;
    clctx->statsAsync.samples += (iteration > 0) ? sum.raygenQueue : 0;
    lastCnt = sum;
    iteration++;
}

// reference: src/tracer.cpp:362-528, wavefront body
std::string Tracer::runBenchmark(double seconds, int iterations)
{
    using clk = std::chrono::steady_clock;
    auto now = [] { return std::chrono::duration<double>(clk::now().time_since_epoch()).count(); };
    std::ostringstream csv;
    csv << "scene;time;primary;extension;shadow;total;samples\n";
    auto R = ranks();
    if (!useWavefront && !peers.empty()) throw std::runtime_error("runBenchmark: the microkernel integrator is single-GPU");
    // resetRenderer (:372-382)
    iteration = 0;
    paramsUpdatePending = false;
    for (auto *c : R) {
        c->updateParams(params);
        c->resetPixelIndex();
        c->enqueueWfResetKernel(params);
        c->enqueueClearWfQueues();
        if (peers.empty()) { c->enqueueResetKernel(params); c->fetchStatsAsync(); }   // :377; drains + zeroes the device-side MK counters
        c->finishQueue();
        c->resetStats();
    }
    double startT = now(), lastLog = startT, currT = startT;
    int it = 0;
    auto log = [&](double t) {
        RenderStats s = clctx->getStats(); clctx->resetStats();
        double dt = t - lastLog, sc = 1e6 * dt; lastLog = t;
        csv << sceneName << ";" << (t - startT) << ";" << s.primaryRays / sc << ";" << s.extensionRays / sc << ";" << s.shadowRays / sc << ";"
            << (s.primaryRays + s.extensionRays + s.shadowRays) / sc << ";" << s.samples / sc << "\n";
    };
    std::vector<QueueCounters> cnt(R.size());
    while (iterations > 0 ? it < iterations : currT - startT < seconds) {
        std::memset(cnt.data(), 0, cnt.size() * sizeof(QueueCounters));
        for (size_t r = 0; r < R.size(); r++) {
            HipContext *c = R[r];
            if (useWavefront) {
                c->enqueueWfLogicKernel(params, false);
                c->enqueueWfRaygenKernel(params);
                c->enqueueWfMaterialKernels(params);
                c->enqueueGetCounters(&cnt[r]);
                c->enqueueWfExtRayKernel(params);
                c->enqueueWfShadowRayKernel(params);
                c->enqueueClearWfQueues();
            } else {                                                     // :441-447
                c->enqueueRayGenKernel(params);
                c->enqueueNextVertexKernel(params);
                c->enqueueBsdfSampleKernel(params);
                c->enqueueSplatKernel(params);
                c->fetchStatsAsync();
            }
            c->enqueuePostprocessKernel(params);
        }
        for (auto *c : R) c->finishQueue();
        QueueCounters sum; std::memset(&sum, 0, sizeof(sum));
        for (size_t r = 0; r < R.size(); r++) {
            if (useWavefront) {
                clctx->statsAsync.extensionRays += cnt[r].extensionQueue;
                clctx->statsAsync.shadowRays += cnt[r].shadowQueue;
                clctx->statsAsync.primaryRays += cnt[r].raygenQueue;
                clctx->statsAsync.samples += (iteration > 0) ? cnt[r].raygenQueue : 0;
            }
            R[r]->updatePixelIndex(R[r]->localPixels(), cnt[r].raygenQueue);
            sum.raygenQueue += cnt[r].raygenQueue; sum.extensionQueue += cnt[r].extensionQueue; sum.shadowQueue += cnt[r].shadowQueue;
            sum.diffuseQueue += cnt[r].diffuseQueue; sum.glossyQueue += cnt[r].glossyQueue; sum.ggxReflQueue += cnt[r].ggxReflQueue;
            sum.ggxRefrQueue += cnt[r].ggxRefrQueue; sum.deltaQueue += cnt[r].deltaQueue;
        }
        lastCnt = sum;
        iteration++; it++;
        currT = now();
        if (currT - lastLog > 0.5) log(currT);
    }
    log(now());
    return csv.str();
}

} // namespace fluctus
