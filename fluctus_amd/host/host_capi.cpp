// host_capi.cpp -- C entry points of libfluctus_host.so (scene / BVH / env-map preparation).
// Pure CPU, no device code.  Consumed by the Python harness (tests, bench.py) through ctypes and
// by the C++ Tracer.  Every function returns 0 on success, non-zero on error (message via
// fh_last_error()), never throws across the boundary.
#include "scene.hpp"
#include "bvh.hpp"
#include "envmap.hpp"
#include "texture.hpp"
#include "../csrc/flx_wide.h"      // the 4-wide tree builder is plain host C++ (flx_upload_scene runs it); exposed here for CPU-side tests
#include <cstring>
#include <string>
#include <exception>
#include <stdexcept>

using namespace fluctus;

static thread_local std::string g_err;
#define FH_TRY try {
#define FH_CATCH } catch (const std::exception &e) { g_err = e.what(); return 1; } catch (...) { g_err = "unknown error"; return 1; } return 0;

extern "C" {

const char *fh_last_error() { return g_err.c_str(); }

// ---- Scene
int fh_scene_create(void **out) { FH_TRY *out = new Scene(); FH_CATCH }
int fh_scene_destroy(void *s) { delete (Scene *)s; return 0; }
int fh_scene_load(void *s, const char *path) { FH_TRY ((Scene *)s)->loadModel(path); FH_CATCH }
int fh_scene_world_up(void *s, float *out3) { flx_vec3 u = ((Scene *)s)->getWorldUp(); out3[0] = u.x; out3[1] = u.y; out3[2] = u.z; return 0; }
int fh_scene_generate(void *s, const char *kind, uint32_t targetTris, uint32_t seed) { FH_TRY ((Scene *)s)->generate(kind, targetTris, seed); FH_CATCH }
int fh_scene_counts(void *s, uint64_t *ntris, uint64_t *nmats, uint64_t *ntex, uint64_t *texbytes, uint32_t *typeBits)
{
    FH_TRY
    Scene *sc = (Scene *)s;
    *ntris = sc->getTriangles().size(); *nmats = sc->getMaterials().size(); *ntex = sc->getTextures().size();
    uint64_t b = 0; for (auto &t : sc->getTextures()) b += (uint64_t)t.width * t.height * 4;
    *texbytes = b; *typeBits = sc->getMaterialTypes();
    FH_CATCH
}
int fh_scene_get(void *s, void *tris, void *mats, void *texdesc, uint8_t *texdata)
{
    FH_TRY
    Scene *sc = (Scene *)s;
    if (tris) memcpy(tris, sc->getTriangles().data(), sc->getTriangles().size() * sizeof(flx_triangle));
    if (mats) memcpy(mats, sc->getMaterials().data(), sc->getMaterials().size() * sizeof(flx_material));
    std::vector<flx_texdesc> d; std::vector<uint8_t> blob;
    sc->packTextures(d, blob);
    if (texdesc && !d.empty()) memcpy(texdesc, d.data(), d.size() * sizeof(flx_texdesc));
    if (texdata && !blob.empty()) memcpy(texdata, blob.data(), blob.size());
    FH_CATCH
}
int fh_scene_add_material(void *s, const void *mat80, int *idx) { FH_TRY *idx = ((Scene *)s)->addMaterial(*(const flx_material *)mat80); FH_CATCH }
int fh_scene_set_tri_material(void *s, uint64_t first, uint64_t count, int matId)
{
    FH_TRY
    auto &t = ((Scene *)s)->getTriangles();
    for (uint64_t i = first; i < first + count && i < t.size(); i++) t[i].matId = matId;
    FH_CATCH
}

// PNG or JPEG by file signature (the name is historical); RGBA8, lower-left origin
int fh_png_load(const char *path, uint32_t *w, uint32_t *h, uint8_t *rgba /* null: query size */)
{
    FH_TRY
    Texture t = loadTexture(path);
    *w = t.width; *h = t.height;
    if (rgba) memcpy(rgba, t.rgba.data(), t.rgba.size());
    FH_CATCH
}

// JPEG from memory: RGB8, top-left origin (libjpeg scanline order); rgb == null queries the size
int fh_jpeg_decode(const uint8_t *data, uint64_t size, uint32_t *w, uint32_t *h, uint8_t *rgb, uint64_t cap)
{
    FH_TRY
    std::vector<uint8_t> out;
    decodeJPEG(data, (size_t)size, w, h, out);
    if (rgb) { if (out.size() > cap) throw std::runtime_error("fh_jpeg_decode: buffer too small"); memcpy(rgb, out.data(), out.size()); }
    FH_CATCH
}

// ---- BVH   mode: 0 = SBVH, 1 = SAH sweep, 2 = binned
int fh_bvh_build(const void *tris, uint64_t ntris, int mode, void **out)
{
    FH_TRY
    std::vector<flx_triangle> *copy = new std::vector<flx_triangle>((const flx_triangle *)tris, (const flx_triangle *)tris + ntris);
    BVH *b = new BVH();
    try { b->build(copy, mode == 0 ? BVH::Mode::SBVH : mode == 1 ? BVH::Mode::SAH : BVH::Mode::Binned); }
    catch (...) { delete copy; delete b; throw; }
    delete copy;
    *out = b;
    FH_CATCH
}
// threads: 0 = every core this process may use (affinity capped by the cgroup CPU quota), 1 = the serial recursion; jobSize 0 = default
int fh_bvh_build_ex(const void *tris, uint64_t ntris, int mode, int threads, uint64_t jobSize, void **out)
{
    FH_TRY
    std::vector<flx_triangle> *copy = new std::vector<flx_triangle>((const flx_triangle *)tris, (const flx_triangle *)tris + ntris);
    BVH *b = new BVH();
    b->sbvhThreads = threads; b->sbvhJobSize = (size_t)jobSize;
    try { b->build(copy, mode == 0 ? BVH::Mode::SBVH : mode == 1 ? BVH::Mode::SAH : BVH::Mode::Binned); }
    catch (...) { delete copy; delete b; throw; }
    delete copy;
    *out = b;
    FH_CATCH
}
// Build the 4-wide quantised tree of a scene on the CPU exactly as flx_upload_scene does and CHECK its invariants (tests/test_host.py):
//  every quantised child box contains the child's exact box (real arithmetic), every binary leaf is referenced exactly once and its
//  block carries the leaf's box, count and triangles, every inner record is referenced exactly once, unused slots point at the dummy
//  leaf.  out8 = {wide nodes, leaf data float4s, stack bound, nested, leaves, max children per node histogram packed: [5]=2-slot nodes,
//  [6]=3-slot, [7]=4-slot}.
static int wide_tree_check(const void *nodesv, uint64_t nnodes, const void *trisv, uint64_t ntris, const uint32_t *indices, uint64_t nidx, uint64_t *out8, double *areas2);
int fh_wide_tree_check(const void *nodesv, uint64_t nnodes, const void *trisv, uint64_t ntris, const uint32_t *indices, uint64_t nidx, uint64_t *out8)
{
    return wide_tree_check(nodesv, nnodes, trisv, ntris, indices, nidx, out8, nullptr);
}
// + areas2 = {sum of the EXACT surface areas of everything a wide-node slot refers to, sum of the slots' QUANTISED box areas}: their ratio
// is what the 8-bit boxes cost in expected visits (surface-area heuristic)
int fh_wide_tree_areas(const void *nodesv, uint64_t nnodes, const void *trisv, uint64_t ntris, const uint32_t *indices, uint64_t nidx, uint64_t *out8, double *areas2)
{
    return wide_tree_check(nodesv, nnodes, trisv, ntris, indices, nidx, out8, areas2);
}
static int wide_tree_check(const void *nodesv, uint64_t nnodes, const void *trisv, uint64_t ntris, const uint32_t *indices, uint64_t nidx, uint64_t *out8, double *areas2)
{
    FH_TRY
    double areaExact = 0.0, areaQuant = 0.0;
    const flx_node *nodes = (const flx_node *)nodesv;
    const flx_triangle *tris = (const flx_triangle *)trisv;
    flxw::WideTree w; const char *err = nullptr;
    if (!flxw::build_wide(nodes, nnodes, tris, ntris, indices, nidx, w, &err)) throw std::runtime_error(err ? err : "build_wide failed");
    // leaf blocks by offset
    std::vector<uint32_t> leafOfOffset(w.leafdata.size(), 0xFFFFFFFFu);
    uint64_t nleaves = 0;
    {
        size_t off = 5;                                                   // after the dummy leaf
        for (size_t i = 0; i < nnodes; i++) {
            if (!nodes[i].nPrims) continue;
            nleaves++;
            if (off + 2 + 3 * (size_t)nodes[i].nPrims > w.leafdata.size()) throw std::runtime_error("leaf data too short");
            const flxw::F4 &h0 = w.leafdata[off], &h1 = w.leafdata[off + 1];
            int cnt; memcpy(&cnt, &h0.w, 4);
            if (cnt != nodes[i].nPrims || h0.x != nodes[i].bmin.x || h0.y != nodes[i].bmin.y || h0.z != nodes[i].bmin.z ||
                h1.x != nodes[i].bmax.x || h1.y != nodes[i].bmax.y || h1.z != nodes[i].bmax.z) throw std::runtime_error("leaf header differs from the binary leaf");
            for (uint32_t k = 0; k < nodes[i].nPrims; k++) {
                const flxw::F4 &a = w.leafdata[off + 2 + 3 * k];
                int ti; memcpy(&ti, &a.w, 4);
                if ((uint32_t)ti != indices[nodes[i].iStartOrRight + k] || a.x != tris[ti].v0.p.x) throw std::runtime_error("leaf triangle order differs from the index list");
            }
            leafOfOffset[off] = (uint32_t)i;
            off += 2 + 3 * (size_t)nodes[i].nPrims;
        }
        if (off != w.leafdata.size()) throw std::runtime_error("leaf data size");
    }
    uint64_t hist[5] = {0, 0, 0, 0, 0};
    if (w.rootRef & FLX_WIDE_LEAF_BIT) { out8[0] = w.nodes.size(); out8[1] = w.leafdata.size(); out8[2] = w.maxStack; out8[3] = w.nested; out8[4] = nleaves; out8[5] = out8[6] = out8[7] = 0; return 0; }
    // children are numbered after their parent, so one backwards sweep sees every child before its parent: exact[i] = union of the
    // exact leaf boxes below wide node i, and every slot's REAL-arithmetic planes must enclose the exact box of what hangs there
    std::vector<uint8_t> leafSeen(nnodes, 0), nodeSeen(w.nodes.size(), 0);
    std::vector<Box> exact(w.nodes.size());
    nodeSeen[0] = 1;
    for (size_t wi = w.nodes.size(); wi-- > 0;) {
        const flxw::WNode &n = w.nodes[wi];
        const uint32_t refs[4] = {n.c0, n.c1, n.c2, n.c3};
        const uint32_t ql[3] = {n.qlox, n.qloy, n.qloz}, qh[3] = {n.qhix, n.qhiy, n.qhiz};
        const long double o[3] = {n.ox, n.oy, n.oz}, sc[3] = {n.sx, n.sy, n.sz};
        int used = 0;
        for (int c = 0; c < 4; c++) {
            if (refs[c] == FLX_WIDE_EMPTY) {
                for (int a = 0; a < 3; a++) if (((ql[a] >> (8 * c)) & 255u) != 255u || ((qh[a] >> (8 * c)) & 255u) != 0u) throw std::runtime_error("unused slot without an inverted box");
                continue;
            }
            used++;
            Box sub;
            if (refs[c] & FLX_WIDE_LEAF_BIT) {
                const uint32_t off = refs[c] & FLX_WIDE_OFF_MASK;
                if (off >= leafOfOffset.size() || leafOfOffset[off] == 0xFFFFFFFFu) throw std::runtime_error("leaf ref does not point at a leaf block");
                const uint32_t li = leafOfOffset[off];
                if (leafSeen[li]++) throw std::runtime_error("leaf referenced twice");
                sub.expand(&nodes[li].bmin.x); sub.expand(&nodes[li].bmax.x);
            } else {
                if (refs[c] >= w.nodes.size() || refs[c] <= wi) throw std::runtime_error("inner ref out of range or not after its parent");
                if (nodeSeen[refs[c]]++) throw std::runtime_error("wide node referenced twice");
                sub = exact[refs[c]];
            }
            for (int a = 0; a < 3; a++) {
                const long double lo = o[a] + (long double)((ql[a] >> (8 * c)) & 255u) * sc[a], hi = o[a] + (long double)((qh[a] >> (8 * c)) & 255u) * sc[a];
                if (lo > (long double)sub.mn[a] || hi < (long double)sub.mx[a]) throw std::runtime_error("quantised box does not contain the exact box of its subtree");
            }
            {
                double q[3];
                for (int a = 0; a < 3; a++) q[a] = (double)(((qh[a] >> (8 * c)) & 255u) - (long double)((ql[a] >> (8 * c)) & 255u)) * (double)sc[a];
                areaQuant += q[0] * q[1] + q[1] * q[2] + q[2] * q[0];
                const double e0 = (double)sub.mx[0] - sub.mn[0], e1 = (double)sub.mx[1] - sub.mn[1], e2 = (double)sub.mx[2] - sub.mn[2];
                areaExact += e0 * e1 + e1 * e2 + e2 * e0;
            }
            exact[wi].expand(sub);
        }
        if (used < 2) throw std::runtime_error("wide node with fewer than two children");
        hist[used]++;
        for (int a = 0; a < 3; a++) if ((long double)exact[wi].mn[a] < o[a]) throw std::runtime_error("grid origin above the subtree's min corner");
    }
    {   // the root's exact box is the binary root's
        const flx_node &r = nodes[0];
        if (exact[0].mn[0] != r.bmin.x || exact[0].mn[1] != r.bmin.y || exact[0].mn[2] != r.bmin.z || exact[0].mx[0] != r.bmax.x || exact[0].mx[1] != r.bmax.y || exact[0].mx[2] != r.bmax.z)
            throw std::runtime_error("union of the leaf boxes differs from the binary root's box");
    }
    for (size_t i = 0; i < nnodes; i++) if (nodes[i].nPrims && leafSeen[i] != 1) throw std::runtime_error("binary leaf not referenced exactly once");
    for (size_t i = 0; i < w.nodes.size(); i++) if (!nodeSeen[i]) throw std::runtime_error("wide node unreachable");
    out8[0] = w.nodes.size(); out8[1] = w.leafdata.size(); out8[2] = w.maxStack; out8[3] = w.nested; out8[4] = nleaves;
    out8[5] = hist[2]; out8[6] = hist[3]; out8[7] = hist[4];
    if (areas2) { areas2[0] = areaExact; areas2[1] = areaQuant; }
    FH_CATCH
}

int fh_usable_threads() { return BVH::usableThreads(); }
int fh_bvh_destroy(void *b) { delete (BVH *)b; return 0; }
int fh_bvh_counts(void *b, uint64_t *nnodes, uint64_t *nidx, uint32_t *metrics4)
{
    FH_TRY
    BVH *v = (BVH *)b; *nnodes = v->m_nodes.size(); *nidx = v->m_indices.size();
    if (metrics4) { metrics4[0] = v->metrics.depth; metrics4[1] = v->metrics.splits; metrics4[2] = v->metrics.duplicates; metrics4[3] = v->metrics.spatialSplits; }
    FH_CATCH
}
int fh_bvh_get(void *b, void *nodes, uint32_t *indices, float *worldRadius)
{
    FH_TRY
    BVH *v = (BVH *)b;
    if (nodes) memcpy(nodes, v->m_nodes.data(), v->m_nodes.size() * sizeof(flx_node));
    if (indices) memcpy(indices, v->m_indices.data(), v->m_indices.size() * 4);
    if (worldRadius) *worldRadius = v->worldRadius();
    FH_CATCH
}
int fh_bvh_export(void *b, const char *path) { FH_TRY ((BVH *)b)->exportTo(path); FH_CATCH }
int fh_bvh_import(const char *path, void **out)
{
    FH_TRY
    BVH *b = new BVH();
    if (!b->importFrom(path)) { delete b; throw std::runtime_error(std::string("cannot import BVH from ") + path); }
    *out = b;
    FH_CATCH
}

// ---- Environment map
int fh_envmap_load(const char *path, void **out) { FH_TRY *out = new EnvironmentMap(path); FH_CATCH }
int fh_envmap_from_memory(int w, int h, const float *rgb, void **out) { FH_TRY *out = new EnvironmentMap(w, h, rgb); FH_CATCH }
int fh_envmap_destroy(void *e) { delete (EnvironmentMap *)e; return 0; }
int fh_envmap_dims(void *e, int *w, int *h) { *w = ((EnvironmentMap *)e)->getWidth(); *h = ((EnvironmentMap *)e)->getHeight(); return 0; }
int fh_envmap_get(void *e, float *rgb, float *prob, int *alias, float *pdf)
{
    FH_TRY
    EnvironmentMap *m = (EnvironmentMap *)e; size_t n = (size_t)m->getWidth() * m->getHeight();
    if (rgb) memcpy(rgb, m->getData(), n * 3 * 4);
    if (prob) memcpy(prob, m->getProbTable(), n * 4);
    if (alias) memcpy(alias, m->getAliasTable(), n * 4);
    if (pdf) memcpy(pdf, m->getPdfTable(), n * 4);
    FH_CATCH
}

} // extern "C"

// ---- headless Tracer (C++ render driver over HipContext / libfluctus_hip.so)
#include "tracer.hpp"
extern "C" {
int fh_tracer_create(int width, int height, int device, uint32_t numTasks, void **out) { FH_TRY *out = new Tracer(width, height, device, numTasks); FH_CATCH }
int fh_tracer_create_multi(int width, int height, const int *devices, int ndev, uint32_t numTasks, void **out)
{
    FH_TRY *out = new Tracer(width, height, std::vector<int>(devices, devices + ndev), numTasks); FH_CATCH
}
int fh_tracer_num_ranks(void *t) { return (int)((Tracer *)t)->numRanks(); }
int fh_tracer_read_accumulation(void *t, float *out, uint64_t capFloats)
{
    FH_TRY
    std::vector<float> px; ((Tracer *)t)->readAccumulation(px);
    if (px.size() > capFloats) throw std::runtime_error("fh_tracer_read_accumulation: buffer too small");
    memcpy(out, px.data(), px.size() * sizeof(float));
    FH_CATCH
}
int fh_tracer_destroy(void *t) { delete (Tracer *)t; return 0; }
int fh_tracer_init(void *t, int width, int height, const char *scene) { FH_TRY ((Tracer *)t)->init(width, height, scene); FH_CATCH }
int fh_tracer_set_envmap(void *t, const char *hdr) { FH_TRY ((Tracer *)t)->setEnvMap(hdr); FH_CATCH }
int fh_tracer_params(void *t, void *out240, const void *in240)
{
    FH_TRY
    Tracer *tr = (Tracer *)t;
    if (in240) { memcpy(&tr->getParams(), in240, 240); tr->paramsChanged(); }
    if (out240) memcpy(out240, &tr->getParams(), 240);
    FH_CATCH
}
int fh_tracer_update(void *t, void *counters32) { FH_TRY ((Tracer *)t)->update(); if (counters32) memcpy(counters32, &((Tracer *)t)->lastCounters(), 32); FH_CATCH }
int fh_tracer_set_cache_dirs(void *t, const char *hierarchies, const char *states) { ((Tracer *)t)->hierarchyCacheDir = hierarchies ? hierarchies : ""; ((Tracer *)t)->stateDir = states ? states : ""; return 0; }
int fh_tracer_save_state(void *t) { return ((Tracer *)t)->saveState() ? 0 : 1; }
int fh_tracer_load_state(void *t) { return ((Tracer *)t)->loadState() ? 0 : 1; }
int fh_tracer_scene_hash(void *t, char *out, uint64_t cap) { const std::string &h = ((Tracer *)t)->getSceneHash(); if (h.size() + 1 > cap) return 1; memcpy(out, h.c_str(), h.size() + 1); return 0; }
uint64_t fh_xxh64(const void *data, uint64_t len, uint64_t seed) { return xxh64(data, (size_t)len, seed); }
int fh_tracer_render_single(void *t, int spp, int denoise) { FH_TRY ((Tracer *)t)->renderSingle(spp, denoise != 0); FH_CATCH }
int fh_tracer_set_option(void *t, const char *name, int value) { FH_TRY ((Tracer *)t)->setOption(name ? name : "", value); FH_CATCH }
int fh_tracer_set_denoiser(void *t, int on) { FH_TRY ((Tracer *)t)->setDenoiser(on != 0); FH_CATCH }
int fh_tracer_toggle_renderer(void *t) { FH_TRY ((Tracer *)t)->toggleRenderer(); FH_CATCH }
int fh_tracer_uses_wavefront(void *t) { return ((Tracer *)t)->usesWavefront() ? 1 : 0; }
int fh_tracer_stats(void *t, uint64_t *out4)
{
    FH_TRY RenderStats s = ((Tracer *)t)->getContext()->getStats(); out4[0] = s.primaryRays; out4[1] = s.extensionRays; out4[2] = s.shadowRays; out4[3] = s.samples; FH_CATCH
}
int fh_tracer_run_benchmark(void *t, double seconds, int iterations, char *csv, uint64_t cap)
{
    FH_TRY
    std::string s = ((Tracer *)t)->runBenchmark(seconds, iterations);
    if (csv && cap) { size_t n = s.size() < cap - 1 ? s.size() : cap - 1; memcpy(csv, s.data(), n); csv[n] = 0; }
    FH_CATCH
}
int fh_tracer_read_pixels(void *t, int which, float *out, uint64_t capFloats)
{
    FH_TRY
    std::vector<float> px; ((Tracer *)t)->getContext()->readPixels(which, px);
    if (px.size() > capFloats) throw std::runtime_error("fh_tracer_read_pixels: buffer too small");
    memcpy(out, px.data(), px.size() * 4);
    FH_CATCH
}
int fh_tracer_save_image(void *t, const char *path) { FH_TRY ((Tracer *)t)->saveImage(path); FH_CATCH }
}
