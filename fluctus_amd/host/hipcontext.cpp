// hipcontext.cpp -- see hipcontext.hpp.
#include "hipcontext.hpp"
#include <dlfcn.h>
#include <stdexcept>
#include <cstdio>
#include <cmath>
#include <fstream>

namespace fluctus {

struct HipContext::Api {
#define FN(name) decltype(&::name) name = nullptr;
    FN(flx_create) FN(flx_destroy) FN(flx_last_error) FN(flx_upload_scene) FN(flx_upload_envmap) FN(flx_set_params)
    FN(flx_wf_reset) FN(flx_wf_raygen) FN(flx_wf_extend) FN(flx_wf_shadow) FN(flx_wf_logic) FN(flx_wf_materials)
    FN(flx_clear_queues) FN(flx_get_counters_async) FN(flx_finish) FN(flx_pixel_index_update) FN(flx_pixel_index_reset)
    FN(flx_num_tasks) FN(flx_postprocess) FN(flx_read_pixels) FN(flx_set_partition) FN(flx_local_pixels)
    FN(flx_mk_reset) FN(flx_mk_raygen) FN(flx_mk_next_vertex) FN(flx_mk_sample_bsdf) FN(flx_mk_splat) FN(flx_mk_splat_preview)
    FN(flx_mk_stats_async) FN(flx_mk_stats_reset) FN(flx_set_option) FN(flx_get_option) FN(flx_group_init_local) FN(flx_gather_local)
#undef FN
};

static std::string defaultLibPath()
{
    Dl_info info;
    if (dladdr((void *)&defaultLibPath, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.find_last_of('/');
        return (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/libfluctus_hip.so";
    }
    return "libfluctus_hip.so";
}

HipContext::HipContext(int device, uint32_t numTasks, const std::string &libPath)
{
    std::string path = libPath.empty() ? defaultLibPath() : libPath;
    dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) throw std::runtime_error(std::string("HipContext: cannot load ") + path + ": " + dlerror() + " (the HIP library is required; there is no CPU fallback)");
    api = new Api();
#define BIND(name) api->name = (decltype(api->name))dlsym(dl, #name); if (!api->name) throw std::runtime_error("HipContext: missing symbol " #name);
    BIND(flx_create) BIND(flx_destroy) BIND(flx_last_error) BIND(flx_upload_scene) BIND(flx_upload_envmap) BIND(flx_set_params)
    BIND(flx_wf_reset) BIND(flx_wf_raygen) BIND(flx_wf_extend) BIND(flx_wf_shadow) BIND(flx_wf_logic) BIND(flx_wf_materials)
    BIND(flx_clear_queues) BIND(flx_get_counters_async) BIND(flx_finish) BIND(flx_pixel_index_update) BIND(flx_pixel_index_reset)
    BIND(flx_num_tasks) BIND(flx_postprocess) BIND(flx_read_pixels) BIND(flx_set_partition) BIND(flx_local_pixels)
    BIND(flx_mk_reset) BIND(flx_mk_raygen) BIND(flx_mk_next_vertex) BIND(flx_mk_sample_bsdf) BIND(flx_mk_splat) BIND(flx_mk_splat_preview)
    BIND(flx_mk_stats_async) BIND(flx_mk_stats_reset) BIND(flx_set_option) BIND(flx_get_option) BIND(flx_group_init_local) BIND(flx_gather_local)
#undef BIND
    if (api->flx_create(device, numTasks, &ctx) != 0)
        throw std::runtime_error(std::string("HipContext: ") + api->flx_last_error(nullptr));
}

HipContext::~HipContext()
{
    if (ctx && api) api->flx_destroy(ctx);
    delete api;
    if (dl) dlclose(dl);
}

// reference: CLContext::verify -> clt::check (src/clcontext.cpp:931-936); here an exception instead of exit()
void HipContext::check(int rc, const char *what)
{
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + api->flx_last_error(ctx));
}

void HipContext::uploadSceneData(BVH *bvh, Scene *scene)
{
    std::vector<flx_texdesc> descs; std::vector<uint8_t> blob;
    scene->packTextures(descs, blob);
    auto &tris = scene->getTriangles(); auto &mats = scene->getMaterials();
    check(api->flx_upload_scene(ctx, tris.data(), tris.size(), bvh->m_indices.data(), bvh->m_indices.size(), bvh->m_nodes.data(), bvh->m_nodes.size(),
                                mats.data(), mats.size(), descs.data(), descs.size(), blob.data(), blob.size()), "uploadSceneData");
}
void HipContext::createEnvMap(EnvironmentMap *m)
{
    check(api->flx_upload_envmap(ctx, m->getData(), m->getWidth(), m->getHeight(), m->getProbTable(), m->getAliasTable(), m->getPdfTable()), "createEnvMap");
}
void HipContext::updateParams(const RenderParams &p) { check(api->flx_set_params(ctx, &p), "updateParams"); lastParams = p; }
void HipContext::enqueueWfResetKernel(const RenderParams &) { check(api->flx_wf_reset(ctx), "wf_reset"); }
void HipContext::enqueueWfRaygenKernel(const RenderParams &) { check(api->flx_wf_raygen(ctx), "wf_raygen"); }
void HipContext::enqueueWfExtRayKernel(const RenderParams &) { check(api->flx_wf_extend(ctx), "wf_extension"); }
void HipContext::enqueueWfShadowRayKernel(const RenderParams &) { check(api->flx_wf_shadow(ctx), "wf_shadow"); }
void HipContext::enqueueWfLogicKernel(const RenderParams &, bool first) { check(api->flx_wf_logic(ctx, first ? 1 : 0), "wf_logic"); }
void HipContext::enqueueWfMaterialKernels(const RenderParams &) { check(api->flx_wf_materials(ctx), "wf_materials"); }
// microkernel integrator (src/clcontext.cpp:709-748)
void HipContext::enqueueResetKernel(const RenderParams &) { check(api->flx_mk_reset(ctx), "mk_reset"); }
void HipContext::enqueueRayGenKernel(const RenderParams &) { check(api->flx_mk_raygen(ctx), "mk_raygen"); }
void HipContext::enqueueNextVertexKernel(const RenderParams &) { check(api->flx_mk_next_vertex(ctx), "mk_next_vertex"); }
void HipContext::enqueueBsdfSampleKernel(const RenderParams &) { check(api->flx_mk_sample_bsdf(ctx), "mk_sample_bsdf"); }
void HipContext::enqueueSplatKernel(const RenderParams &) { check(api->flx_mk_splat(ctx), "mk_splat"); }
void HipContext::enqueueSplatPreviewKernel(const RenderParams &) { check(api->flx_mk_splat_preview(ctx), "mk_splat_preview"); }
// reference: CLContext::fetchStatsAsync (src/clcontext.cpp:642-646): the device counters since the last reset land in
// mkStats at the next finishQueue(); foldMkStats() then moves them into statsAsync (64-bit) and zeroes the device side,
// so the 32-bit device counters (src/geom.h:254-260) cannot wrap in a long render.
void HipContext::fetchStatsAsync() { check(api->flx_mk_stats_async(ctx, mkStats), "fetch stats"); check(api->flx_mk_stats_reset(ctx), "reset stats"); mkPending = true; }
void HipContext::foldMkStats()
{
    if (!mkPending) return;
    statsAsync.primaryRays += mkStats[0]; statsAsync.extensionRays += mkStats[1]; statsAsync.shadowRays += mkStats[2]; statsAsync.samples += mkStats[3];
    mkStats[0] = mkStats[1] = mkStats[2] = mkStats[3] = 0; mkPending = false;
}
// The reference rebuilds every kernel with / without -DUSE_OPTIX_DENOISER; here the kernels branch on the presence of
// the feature buffers, so "recompiling" is allocating or freeing them.
// one process, one context per GPU: RCCL group + partitions 0..n-1 (flx_group_init_local), gather of the tiles on `root`
void HipContext::groupInitLocal(const std::vector<HipContext *> &ranks)
{
    if (ranks.empty()) throw std::runtime_error("groupInitLocal: no contexts");
    std::vector<flx_ctx *> h; for (auto *r : ranks) h.push_back(r->ctx);
    ranks[0]->check(ranks[0]->api->flx_group_init_local(h.data(), (uint32_t)h.size()), "groupInitLocal");
}
void HipContext::gatherLocal(const std::vector<HipContext *> &ranks, uint32_t root, std::vector<float> &rgba)
{
    if (ranks.empty() || root >= ranks.size()) throw std::runtime_error("gatherLocal: bad arguments");
    std::vector<flx_ctx *> h; for (auto *r : ranks) h.push_back(r->ctx);
    rgba.resize((size_t)ranks[root]->lastParams.width * ranks[root]->lastParams.height * 4);
    ranks[root]->check(ranks[root]->api->flx_gather_local(h.data(), (uint32_t)h.size(), root, rgba.data()), "gatherLocal");
}
void HipContext::setOption(const std::string &name, int value) { check(api->flx_set_option(ctx, name.c_str(), value), "setOption"); }
int HipContext::getOption(const std::string &name) { int v = 0; check(api->flx_get_option(ctx, name.c_str(), &v), "getOption"); return v; }
void HipContext::recompileKernels(bool useDenoiser) { check(api->flx_set_option(ctx, "denoiser", useDenoiser ? 1 : 0), "recompileKernels"); }
void HipContext::enqueueClearWfQueues() { check(api->flx_clear_queues(ctx), "clear queues"); }
void HipContext::enqueueGetCounters(QueueCounters *cnt) { check(api->flx_get_counters_async(ctx, cnt), "get counters"); }
void HipContext::enqueuePostprocessKernel(const RenderParams &) { check(api->flx_postprocess(ctx), "postprocess"); }
void HipContext::finishQueue() { check(api->flx_finish(ctx), "finish"); foldMkStats(); }
void HipContext::updatePixelIndex(uint32_t n, uint32_t nnew) { check(api->flx_pixel_index_update(ctx, n, nnew), "updatePixelIndex"); }
void HipContext::resetPixelIndex() { check(api->flx_pixel_index_reset(ctx), "resetPixelIndex"); }
uint32_t HipContext::getNumTasks() const { return api->flx_num_tasks(ctx); }
void HipContext::setPartition(uint32_t r, uint32_t n) { check(api->flx_set_partition(ctx, r, n), "setPartition"); }
uint32_t HipContext::localPixels() const { return api->flx_local_pixels(ctx); }

void HipContext::readPixels(int which, std::vector<float> &rgba)
{
    rgba.resize((size_t)localPixels() * 4);
    check(api->flx_read_pixels(ctx, which, rgba.data()), "read pixels");
}

void HipContext::saveImage(const std::string &filename, const RenderParams &p)
{
    const bool raw = filename.size() > 4 && filename.compare(filename.size() - 4, 4, ".pfm") == 0;
    std::vector<float> px;
    if (!raw) { enqueuePostprocessKernel(p); }
    readPixels(raw ? 0 : 1, px);
    const uint32_t w = p.width, h = p.height;
    if ((size_t)w * h != px.size() / 4) throw std::runtime_error("saveImage: partitioned framebuffer; gather the tiles first");
    std::ofstream out(filename, std::ios::binary);
    if (raw) {
        out << "PF\n" << w << " " << h << "\n-1.0\n";                 // little-endian, bottom row first = our row 0 (y up)
        for (uint32_t i = 0; i < w * h; i++) {
            float c = px[i * 4 + 3] > 0 ? px[i * 4 + 3] : 1.0f;
            float rgb[3] = {px[i * 4] / c, px[i * 4 + 1] / c, px[i * 4 + 2] / c};
            out.write((const char *)rgb, 12);
        }
    } else {
        out << "P6\n" << w << " " << h << "\n255\n";
        for (int y = (int)h - 1; y >= 0; y--)
            for (uint32_t x = 0; x < w; x++) {
                unsigned char c[3];
                for (int k = 0; k < 3; k++) { float v = px[((size_t)y * w + x) * 4 + k]; v = v < 0 ? 0 : (v > 1 ? 1 : v); c[k] = (unsigned char)std::lround(v * 255.0f); }
                out.write((const char *)c, 3);
            }
    }
}

void HipContext::updateRenderPerf(float deltaT)
{
    double scale = 1e6 * deltaT;
    renderPerf.primary = statsAsync.primaryRays / scale;
    renderPerf.extension = statsAsync.extensionRays / scale;
    renderPerf.shadow = statsAsync.shadowRays / scale;
    renderPerf.samples = statsAsync.samples / scale;
    renderPerf.total = renderPerf.primary + renderPerf.extension + renderPerf.shadow;
}

} // namespace fluctus
