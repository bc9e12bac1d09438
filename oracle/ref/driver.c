/*
 * driver.c -- sequential NDRange driver for the reference's own wavefront kernels.
 *
 * TEST INFRASTRUCTURE (oracle/_ref build, this container only; /root/reference is read, never
 * copied).  The kernels of /root/reference/src/wf_*.cl and mk_postprocess.cl are compiled for
 * x86-64 by clang's OpenCL front end (see Makefile) and called here as ordinary C functions, one
 * work-item at a time in ascending global id, with the launch ranges and buffer set-up of
 * reference src/clcontext.cpp:116-141,467-566,765-895.  The entry points mirror
 * include/fluctus_hip.h (prefix ref_) so one Python driver loop runs reference, oracle and device.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "../../include/fluctus_wire.h"

extern __thread size_t ref_current_gid;
typedef struct { int width, height; const float *rgba; } ref_image;
typedef unsigned int uint;

/* kernels (argument lists: reference src/wf_*.cl, src/mk_postprocess.cl) */
void reset(float *tasks, float *pixels, float *denAlbedo, float *denNormal, void *queueLens, uint *raygenQueue, void *params, uint numTasks);
void genRays(float *tasks, void *params, void *queueLens, uint *raygenQueue, uint *extensionQueue, uint *currPixelIdx, uint numTasks);
void traceExtension(float *tasks, void *queueLens, uint *extensionQueue, void *tris, void *nodes, uint *indices, void *params, uint numTasks);
void traceShadow(float *tasks, void *queueLens, uint *shadowQueue, void *tris, void *nodes, uint *indices, void *params, uint numTasks);
typedef void (*logic_fn)(float *tasks, float *pixels, float *denNormal, float *denAlbedo, void *queueLens,
                         uint *extQ, uint *shadowQ, uint *raygenQ, uint *diffQ, uint *glossyQ, uint *ggxReflQ, uint *ggxRefrQ, uint *deltaQ,
                         void *tris, void *nodes, uint *indices, const ref_image *envMap, float *probTable, int *aliasTable, float *pdfTable,
                         void *materials, uint8_t *texData, void *textures, void *params, uint numTasks, uint firstIteration);
#define LOGIC_DECL(n) void logic_v##n(float *, float *, float *, float *, void *, uint *, uint *, uint *, uint *, uint *, uint *, uint *, uint *, void *, void *, uint *, const ref_image *, float *, int *, float *, void *, uint8_t *, void *, void *, uint, uint);
LOGIC_DECL(0) LOGIC_DECL(1) LOGIC_DECL(2) LOGIC_DECL(3) LOGIC_DECL(4) LOGIC_DECL(5) LOGIC_DECL(6) LOGIC_DECL(7)
LOGIC_DECL(8) LOGIC_DECL(9) LOGIC_DECL(10) LOGIC_DECL(11) LOGIC_DECL(12) LOGIC_DECL(13) LOGIC_DECL(14) LOGIC_DECL(15)
LOGIC_DECL(16) LOGIC_DECL(17) LOGIC_DECL(18) LOGIC_DECL(19) LOGIC_DECL(20) LOGIC_DECL(21) LOGIC_DECL(22) LOGIC_DECL(23)
LOGIC_DECL(24) LOGIC_DECL(25) LOGIC_DECL(26) LOGIC_DECL(27) LOGIC_DECL(28) LOGIC_DECL(29) LOGIC_DECL(30) LOGIC_DECL(31)
#undef LOGIC_DECL
#define LOGIC_DECL(n) void logic_d##n(float *, float *, float *, float *, void *, uint *, uint *, uint *, uint *, uint *, uint *, uint *, uint *, void *, void *, uint *, const ref_image *, float *, int *, float *, void *, uint8_t *, void *, void *, uint, uint);
LOGIC_DECL(0) LOGIC_DECL(1) LOGIC_DECL(2) LOGIC_DECL(3) LOGIC_DECL(4) LOGIC_DECL(5) LOGIC_DECL(6) LOGIC_DECL(7)
LOGIC_DECL(8) LOGIC_DECL(9) LOGIC_DECL(10) LOGIC_DECL(11) LOGIC_DECL(12) LOGIC_DECL(13) LOGIC_DECL(14) LOGIC_DECL(15)
LOGIC_DECL(16) LOGIC_DECL(17) LOGIC_DECL(18) LOGIC_DECL(19) LOGIC_DECL(20) LOGIC_DECL(21) LOGIC_DECL(22) LOGIC_DECL(23)
LOGIC_DECL(24) LOGIC_DECL(25) LOGIC_DECL(26) LOGIC_DECL(27) LOGIC_DECL(28) LOGIC_DECL(29) LOGIC_DECL(30) LOGIC_DECL(31)
static const logic_fn logic_variants_den[32] = {                    /* built with -DUSE_OPTIX_DENOISER */
    logic_d0, logic_d1, logic_d2, logic_d3, logic_d4, logic_d5, logic_d6, logic_d7, logic_d8, logic_d9, logic_d10, logic_d11,
    logic_d12, logic_d13, logic_d14, logic_d15, logic_d16, logic_d17, logic_d18, logic_d19, logic_d20, logic_d21, logic_d22,
    logic_d23, logic_d24, logic_d25, logic_d26, logic_d27, logic_d28, logic_d29, logic_d30, logic_d31};
static const logic_fn logic_variants[32] = {
    logic_v0, logic_v1, logic_v2, logic_v3, logic_v4, logic_v5, logic_v6, logic_v7, logic_v8, logic_v9, logic_v10, logic_v11,
    logic_v12, logic_v13, logic_v14, logic_v15, logic_v16, logic_v17, logic_v18, logic_v19, logic_v20, logic_v21, logic_v22,
    logic_v23, logic_v24, logic_v25, logic_v26, logic_v27, logic_v28, logic_v29, logic_v30, logic_v31};
typedef void (*mat_fn)(float *tasks, void *queueLens, uint *matQueue, uint *extensionQueue, void *materials, uint8_t *texData, void *textures, void *params, uint numTasks);
void wavefrontDiffuse(float *, void *, uint *, uint *, void *, uint8_t *, void *, void *, uint);
void wavefrontGlossy(float *, void *, uint *, uint *, void *, uint8_t *, void *, void *, uint);
void wavefrontGGXReflection(float *, void *, uint *, uint *, void *, uint8_t *, void *, void *, uint);
void wavefrontGGXRefraction(float *, void *, uint *, uint *, void *, uint8_t *, void *, void *, uint);
void wavefrontDelta(float *, void *, uint *, uint *, void *, uint8_t *, void *, void *, uint);
void wavefrontAllMaterials(float *, void *, uint *, uint *, void *, uint8_t *, void *, void *, uint);
/* microkernel integrator (argument lists: reference src/mk_*.cl) */
void mk_reset(float *tasks, float *pixels, float *denAlbedo, float *denNormal, void *params, uint numTasks);
void genCameraRays(float *tasks, void *params, uint numTasks);
void nextVertex(float *tasks, void *materials, uint8_t *texData, void *textures, float *denNormal, void *tris, void *nodes, uint *indices, void *params,
                void *stats, const ref_image *envMap, float *pdfTable, uint numTasks);
void sampleBsdf(float *tasks, float *denAlbedo, void *materials, uint8_t *texData, void *textures, const ref_image *envMap, float *probTable, int *aliasTable,
                float *pdfTable, void *tris, void *nodes, uint *indices, void *params, void *stats, uint numTasks);
void splat(float *tasks, float *pixels, void *params, void *stats, uint numTasks);
void splatPreview(float *tasks, float *pixels, void *params, uint numTasks);
void process(float *pixelsRaw, float *denAlbedo, float *denNormal, float *pixelsPreview, float *denAlbedoGL, float *denNormalGL, void *params, uint numTasks);
/* -DUSE_OPTIX_DENOISER builds of the kernels that write / resolve the denoiser feature buffers */
void process_d(float *pixelsRaw, float *denAlbedo, float *denNormal, float *pixelsPreview, float *denAlbedoGL, float *denNormalGL, void *params, uint numTasks);
void nextVertex_d(float *tasks, void *materials, uint8_t *texData, void *textures, float *denNormal, void *tris, void *nodes, uint *indices, void *params,
                  void *stats, const ref_image *envMap, float *pdfTable, uint numTasks);
void sampleBsdf_d(float *tasks, float *denAlbedo, void *materials, uint8_t *texData, void *textures, const ref_image *envMap, float *probTable, int *aliasTable,
                  float *pdfTable, void *tris, void *nodes, uint *indices, void *params, void *stats, uint numTasks);

typedef struct {
    uint numTasks;
    float *tasks;
    uint *queues[FLX_NUM_QUEUES];
    flx_queue_counters counters __attribute__((aligned(64)));
    uint currPixelIdx, hostPixelIdx;
    flx_render_params params __attribute__((aligned(64)));   /* kernels load float3 members with aligned 16-byte moves */
    float *pixels, *preview, *denAlbedo, *denNormal, *denAlbedoGL, *denNormalGL;
    int denoiser;
    int threads;
    size_t npix;
    void *tris; size_t ntris; uint *indices; size_t nidx; void *nodes; size_t nnodes;
    void *materials; size_t nmat; void *texdesc; size_t ntex; uint8_t *texdata; size_t texbytes;
    ref_image env; float *envRGBA; float *prob, *pdf; int *alias;
    uint32_t mkStats[4] __attribute__((aligned(16)));
} ref_ctx;

static void *dup(const void *src, size_t bytes) { void *p = NULL; if (posix_memalign(&p, 64, bytes ? bytes : 64)) return NULL; if (src && bytes) memcpy(p, src, bytes); return p; }

int ref_create(uint32_t num_tasks, ref_ctx **out)
{
    ref_ctx *c = NULL;
    if (posix_memalign((void **)&c, 64, sizeof(ref_ctx))) return 1;
    memset(c, 0, sizeof(ref_ctx));
    c->numTasks = num_tasks;
    c->tasks = (float *)calloc((size_t)FLX_NUM_COLS * num_tasks, 4);
    for (int q = 0; q < FLX_NUM_QUEUES; q++) c->queues[q] = (uint *)calloc(num_tasks, 4);
    c->envRGBA = (float *)calloc(4, 4);                      /* dummy 1x1 env map (clcontext.cpp:513-518) */
    c->env.width = c->env.height = 1; c->env.rgba = c->envRGBA;
    c->prob = (float *)calloc(1, 4); c->pdf = (float *)calloc(1, 4); c->alias = (int *)calloc(1, 4);
    c->prob[0] = 1.0f; c->pdf[0] = 1.0f;
    *out = c;
    return 0;
}
int ref_destroy(ref_ctx *c)
{
    free(c->tasks); for (int q = 0; q < FLX_NUM_QUEUES; q++) free(c->queues[q]);
    free(c->pixels); free(c->preview); free(c->denAlbedo); free(c->denNormal); free(c->denAlbedoGL); free(c->denNormalGL);
    free(c->tris); free(c->indices); free(c->nodes); free(c->materials); free(c->texdesc); free(c->texdata);
    free(c->envRGBA); free(c->prob); free(c->pdf); free(c->alias); free(c);
    return 0;
}
int ref_upload_scene(ref_ctx *c, const void *tris, size_t ntris, const uint32_t *indices, size_t nidx, const void *nodes, size_t nnodes,
                     const void *materials, size_t nmat, const void *texdesc, size_t ntex, const uint8_t *texdata, size_t texbytes)
{
    free(c->tris); free(c->indices); free(c->nodes); free(c->materials); free(c->texdesc); free(c->texdata);
    c->tris = dup(tris, ntris * 160); c->indices = (uint *)dup(indices, nidx * 4); c->nodes = dup(nodes, nnodes * 48);
    c->materials = dup(materials, nmat * 80); c->texdesc = dup(texdesc, ntex * 12); c->texdata = (uint8_t *)dup(texdata, texbytes);
    c->ntris = ntris; c->nidx = nidx; c->nnodes = nnodes; c->nmat = nmat; c->ntex = ntex; c->texbytes = texbytes;
    return 0;
}
int ref_upload_envmap(ref_ctx *c, const float *rgb, int w, int h, const float *prob, const int *alias, const float *pdf)
{
    size_t n = (size_t)w * h;
    free(c->envRGBA); free(c->prob); free(c->pdf); free(c->alias);
    c->envRGBA = (float *)malloc(n * 16);
    for (size_t i = 0; i < n; i++) { c->envRGBA[i * 4] = rgb[i * 3]; c->envRGBA[i * 4 + 1] = rgb[i * 3 + 1]; c->envRGBA[i * 4 + 2] = rgb[i * 3 + 2]; c->envRGBA[i * 4 + 3] = 1.0f; }
    c->env.width = w; c->env.height = h; c->env.rgba = c->envRGBA;
    c->prob = (float *)dup(prob, n * 4); c->alias = (int *)dup(alias, n * 4); c->pdf = (float *)dup(pdf, n * 4);
    return 0;
}
int ref_set_params(ref_ctx *c, const void *p240)
{
    memcpy(&c->params, p240, 240);
    size_t npix = (size_t)c->params.width * c->params.height;
    if (npix != c->npix) {
        free(c->pixels); free(c->preview); free(c->denAlbedo); free(c->denNormal); free(c->denAlbedoGL); free(c->denNormalGL);
        c->pixels = (float *)calloc(npix, 16); c->preview = (float *)calloc(npix, 16);
        c->denAlbedo = (float *)calloc(npix, 16); c->denNormal = (float *)calloc(npix, 16);
        c->denAlbedoGL = (float *)calloc(npix, 16); c->denNormalGL = (float *)calloc(npix, 16);
        c->npix = npix;
    }
    return 0;
}

/* NDRange: ascending ids on one thread (default; deterministic queue order for the parity tests) or chunks of 64
 * work-items over c->threads host threads, as a CPU OpenCL device would schedule work-groups */
#define RANGE(n, call) do { \
    const long n_ = (long)(n); \
    if (c->threads > 1) { \
        _Pragma("omp parallel for schedule(dynamic, 64) num_threads(c->threads)") \
        for (long g_ = 0; g_ < n_; g_++) { ref_current_gid = (size_t)g_; call; } \
    } else for (long g_ = 0; g_ < n_; g_++) { ref_current_gid = (size_t)g_; call; } \
} while (0)

int ref_wf_reset(ref_ctx *c)
{
    size_t n = c->numTasks > c->npix ? c->numTasks : c->npix;      /* clcontext.cpp:767 */
    RANGE(n, reset(c->tasks, c->pixels, c->denAlbedo, c->denNormal, &c->counters, c->queues[FLX_Q_RAYGEN], &c->params, c->numTasks));
    return 0;
}
int ref_wf_raygen(ref_ctx *c)
{
    RANGE(c->numTasks, genRays(c->tasks, &c->params, &c->counters, c->queues[FLX_Q_RAYGEN], c->queues[FLX_Q_EXTENSION], &c->currPixelIdx, c->numTasks));
    return 0;
}
int ref_wf_extend(ref_ctx *c)
{
    RANGE(c->numTasks, traceExtension(c->tasks, &c->counters, c->queues[FLX_Q_EXTENSION], c->tris, c->nodes, c->indices, &c->params, c->numTasks));
    return 0;
}
int ref_wf_shadow(ref_ctx *c)
{
    RANGE(c->numTasks, traceShadow(c->tasks, &c->counters, c->queues[FLX_Q_SHADOW], c->tris, c->nodes, c->indices, &c->params, c->numTasks));
    return 0;
}
int ref_wf_logic(ref_ctx *c, int first)
{
    const flx_render_params *p = &c->params;      /* build flags: kernel_impl.hpp:49-67 */
    int v = (p->useAreaLight ? 1 : 0) | (p->useEnvMap ? 2 : 0) | (p->sampleExpl ? 4 : 0) | (p->sampleImpl ? 8 : 0) | (!p->wfSeparateQueues ? 16 : 0);
    logic_fn f = c->denoiser ? logic_variants_den[v] : logic_variants[v];
    size_t n = ((size_t)(c->numTasks - 1) / 32 + 1) * 32;            /* clcontext.cpp:792 */
    RANGE(n, f(c->tasks, c->pixels, c->denNormal, c->denAlbedo, &c->counters, c->queues[FLX_Q_EXTENSION], c->queues[FLX_Q_SHADOW],
               c->queues[FLX_Q_RAYGEN], c->queues[FLX_Q_DIFFUSE], c->queues[FLX_Q_GLOSSY], c->queues[FLX_Q_GGX_REFL], c->queues[FLX_Q_GGX_REFR],
               c->queues[FLX_Q_DELTA], c->tris, c->nodes, c->indices, &c->env, c->prob, c->alias, c->pdf, c->materials, c->texdata, c->texdesc,
               &c->params, c->numTasks, (uint)first));
    return 0;
}
static void runMat(ref_ctx *c, mat_fn f, int q)
{
    RANGE(c->numTasks, f(c->tasks, &c->counters, c->queues[q], c->queues[FLX_Q_EXTENSION], c->materials, c->texdata, c->texdesc, &c->params, c->numTasks));
}
int ref_wf_materials(ref_ctx *c)
{
    if (c->params.wfSeparateQueues) {                /* clcontext.cpp:796-813 */
        runMat(c, wavefrontDiffuse, FLX_Q_DIFFUSE); runMat(c, wavefrontGlossy, FLX_Q_GLOSSY);
        runMat(c, wavefrontGGXReflection, FLX_Q_GGX_REFL); runMat(c, wavefrontGGXRefraction, FLX_Q_GGX_REFR);
        runMat(c, wavefrontDelta, FLX_Q_DELTA);
    } else runMat(c, wavefrontAllMaterials, FLX_Q_DIFFUSE);
    return 0;
}
int ref_postprocess(ref_ctx *c)
{
    if (c->denoiser) { RANGE(c->npix, process_d(c->pixels, c->denAlbedo, c->denNormal, c->preview, c->denAlbedoGL, c->denNormalGL, &c->params, c->numTasks)); }
    else { RANGE(c->npix, process(c->pixels, c->denAlbedo, c->denNormal, c->preview, c->denAlbedoGL, c->denNormalGL, &c->params, c->numTasks)); }
    return 0;
}
int ref_clear_queues(ref_ctx *c) { memset(&c->counters, 0, 32); return 0; }
int ref_get_counters(ref_ctx *c, void *out32) { memcpy(out32, &c->counters, 32); return 0; }
int ref_set_counters(ref_ctx *c, const void *in32) { memcpy(&c->counters, in32, 32); return 0; }
int ref_pixel_index_update(ref_ctx *c, uint32_t npix, uint32_t nnew) { c->hostPixelIdx = (c->hostPixelIdx + nnew) % npix; c->currPixelIdx = c->hostPixelIdx; return 0; }
int ref_pixel_index_reset(ref_ctx *c) { c->hostPixelIdx = 0; c->currPixelIdx = 0; return 0; }
int ref_read_pixels(ref_ctx *c, int which, float *out)
{
    const float *src[6] = {c->pixels, c->preview, c->denAlbedoGL, c->denNormalGL, c->denAlbedo, c->denNormal};
    if (which < 0 || which > 5) return 1;
    memcpy(out, src[which], c->npix * 16);
    return 0;
}
int ref_set_threads(ref_ctx *c, int n) { c->threads = n < 1 ? 1 : n; return 0; }
int ref_set_option(ref_ctx *c, const char *name, int value) { if (name && strcmp(name, "denoiser") == 0) { c->denoiser = value != 0; return 0; } return 1; }
int ref_state_export(ref_ctx *c, float *out) { memcpy(out, c->tasks, (size_t)FLX_NUM_COLS * c->numTasks * 4); return 0; }
int ref_state_import(ref_ctx *c, const float *in) { memcpy(c->tasks, in, (size_t)FLX_NUM_COLS * c->numTasks * 4); return 0; }
int ref_queue_read(ref_ctx *c, int q, uint32_t *out) { memcpy(out, c->queues[q], (size_t)c->numTasks * 4); return 0; }
int ref_queue_write(ref_ctx *c, int q, const uint32_t *in, uint32_t n) { memcpy(c->queues[q], in, (size_t)n * 4); return 0; }

/* ---- microkernel integrator: launch ranges of src/clcontext.cpp:709-750 (reset/splat: width x height; the rest: NUM_TASKS) */
int ref_mk_reset(ref_ctx *c) { RANGE(c->npix, mk_reset(c->tasks, c->pixels, c->denAlbedo, c->denNormal, &c->params, c->numTasks)); return 0; }
int ref_mk_raygen(ref_ctx *c) { RANGE(c->numTasks, genCameraRays(c->tasks, &c->params, c->numTasks)); return 0; }
int ref_mk_next_vertex(ref_ctx *c)
{
    if (c->denoiser) { RANGE(c->numTasks, nextVertex_d(c->tasks, c->materials, c->texdata, c->texdesc, c->denNormal, c->tris, c->nodes, c->indices, &c->params, c->mkStats, &c->env, c->pdf, c->numTasks)); }
    else { RANGE(c->numTasks, nextVertex(c->tasks, c->materials, c->texdata, c->texdesc, c->denNormal, c->tris, c->nodes, c->indices, &c->params, c->mkStats, &c->env, c->pdf, c->numTasks)); }
    return 0;
}
int ref_mk_sample_bsdf(ref_ctx *c)
{
    if (c->denoiser) { RANGE(c->numTasks, sampleBsdf_d(c->tasks, c->denAlbedo, c->materials, c->texdata, c->texdesc, &c->env, c->prob, c->alias, c->pdf, c->tris, c->nodes, c->indices,
                                                       &c->params, c->mkStats, c->numTasks)); }
    else { RANGE(c->numTasks, sampleBsdf(c->tasks, c->denAlbedo, c->materials, c->texdata, c->texdesc, &c->env, c->prob, c->alias, c->pdf, c->tris, c->nodes, c->indices,
                                         &c->params, c->mkStats, c->numTasks)); }
    return 0;
}
int ref_mk_splat(ref_ctx *c) { RANGE(c->npix, splat(c->tasks, c->pixels, &c->params, c->mkStats, c->numTasks)); return 0; }
int ref_mk_splat_preview(ref_ctx *c) { RANGE(c->npix, splatPreview(c->tasks, c->pixels, &c->params, c->numTasks)); return 0; }
int ref_mk_stats(ref_ctx *c, uint32_t *out4, int reset) { memcpy(out4, c->mkStats, 16); if (reset) memset(c->mkStats, 0, 16); return 0; }
