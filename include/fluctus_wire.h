/*
 * fluctus_wire.h -- wire-format structs of the wavefront path-tracing hot path.
 *
 * These are the byte layouts that cross the drop-in boundary (host <-> device
 * library).  They restate, with our own names, the shared host/device ABI of the
 * reference (reference: src/geom.h:52-252, src/bxdf_types.h:4-11; host mirrors
 * src/triangle.hpp:18-49, src/bvhnode.hpp:50-59; host float3 is 16 bytes,
 * include/math/float3.hpp:31-36).  Sizes and offsets are pinned by static
 * asserts; a host that fills the reference's own structs can hand the same bytes
 * to flx_upload_scene()/flx_set_params() unchanged.
 *
 * Plain C, usable from C, C++ and HIP.
 */
#ifndef FLUCTUS_WIRE_H
#define FLUCTUS_WIRE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 16-byte vector slot: x,y,z + pad (reference: cl_float3 / FireRays::float3). */
typedef struct { float x, y, z, w; } flx_vec3;
typedef struct { float x, y; } flx_vec2;

/* reference: geom.h:82-95 (Vertex, Triangle) / triangle.hpp:6-27 */
typedef struct {
    flx_vec3 p;   /* position            */
    flx_vec3 n;   /* shading normal      */
    flx_vec3 t;   /* texcoord (x,y used) */
} flx_vertex;     /* 48 B */

typedef struct {
    flx_vertex v0, v1, v2;
    int32_t    matId;
    int32_t    _pad[3];
} flx_triangle;   /* 160 B */

/* reference: geom.h:65-80 (AABB, GPUNode) / bvhnode.hpp:50-59 (Node).
 * Left child of an inner node is always index+1; nPrims==0 marks inner nodes. */
typedef struct {
    flx_vec3 bmin;
    flx_vec3 bmax;
    int32_t  parent;
    uint32_t iStartOrRight;  /* leaf: first slot in the index list; inner: right child */
    uint8_t  nPrims;
    uint8_t  _pad[7];
} flx_node;       /* 48 B */

/* reference: geom.h:113-124 */
typedef struct {
    flx_vec3 Kd, Ks, Ke;
    float    Ns;
    float    Ni;
    int32_t  map_Kd, map_Ks, map_N;
    int32_t  type;           /* FLX_BXDF_* */
    int32_t  _pad[2];
} flx_material;   /* 80 B */

/* reference: geom.h:126-131 */
typedef struct { uint32_t offset, width, height; } flx_texdesc; /* 12 B */

/* reference: geom.h:104-111 */
typedef struct {
    flx_vec3 right, up, N, pos, E;
    flx_vec2 size;           /* half extents */
    float    _pad[2];
} flx_arealight;  /* 96 B */

/* reference: geom.h:146-155 */
typedef struct {
    flx_vec3 pos, dir, up, right;
    float    fov;            /* degrees */
    float    apertureSize;
    float    focalDist;
    float    _pad;
} flx_camera;     /* 80 B */

/* reference: geom.h:157-180 */
typedef struct {
    flx_arealight areaLight;      /*   0 */
    flx_camera    camera;         /*  96 */
    float         exposure;       /* 176 ppParams.exposure   */
    uint32_t      tmOperator;     /* 180 ppParams.tmOperator */
    uint32_t      width;          /* 184 */
    uint32_t      height;         /* 188 */
    uint32_t      n_tris;         /* 192 */
    uint32_t      useEnvMap;      /* 196 */
    uint32_t      useAreaLight;   /* 200 */
    float         envMapStrength; /* 204 */
    uint32_t      maxBounces;     /* 208 */
    uint32_t      sampleImpl;     /* 212 */
    uint32_t      sampleExpl;     /* 216 */
    uint32_t      useRoulette;    /* 220 */
    uint32_t      wfSeparateQueues; /* 224 */
    float         worldRadius;    /* 228 */
    uint32_t      _pad[2];
} flx_render_params; /* 240 B */

/* reference: geom.h:240-252 */
typedef struct {
    uint32_t raygenQueue;
    uint32_t extensionQueue;
    uint32_t shadowQueue;
    uint32_t diffuseQueue;
    uint32_t glossyQueue;
    uint32_t ggxReflQueue;
    uint32_t ggxRefrQueue;
    uint32_t deltaQueue;
} flx_queue_counters; /* 32 B */

/* reference: bxdf_types.h:4-11 */
enum {
    FLX_BXDF_DIFFUSE              = 1 << 1,
    FLX_BXDF_GLOSSY               = 1 << 2,
    FLX_BXDF_GGX_ROUGH_REFLECTION = 1 << 3,
    FLX_BXDF_IDEAL_REFLECTION     = 1 << 4,
    FLX_BXDF_GGX_ROUGH_DIELECTRIC = 1 << 5,
    FLX_BXDF_IDEAL_DIELECTRIC     = 1 << 6,
    FLX_BXDF_EMISSIVE             = 1 << 7
};
#define FLX_BXDF_IS_SINGULAR(t) (((t) & (FLX_BXDF_IDEAL_REFLECTION | FLX_BXDF_IDEAL_DIELECTRIC)) != 0)

/*
 * Reference path-state layout (GPUTaskState, geom.h:199-236), used ONLY by the
 * test hooks flx_state_export/flx_state_import and by the oracle: SoA of 64
 * 4-byte columns, element (col, gid) at word col*numTasks + gid.  The device
 * library keeps its own packed layout and converts on demand.
 */
enum {
    FLX_COL_ORIG = 0, FLX_COL_DIR = 4, FLX_COL_SHADOW_ORIG = 8, FLX_COL_SHADOW_DIR = 12,
    FLX_COL_T = 16, FLX_COL_EI = 20, FLX_COL_LAST_BSDF = 24, FLX_COL_LAST_EMISSION = 28,
    FLX_COL_LAST_T = 32, FLX_COL_P = 36, FLX_COL_N = 40, FLX_COL_UV = 44,
    FLX_COL_PHASE = 46, FLX_COL_LAST_PDF_W = 47, FLX_COL_PATH_LEN = 48, FLX_COL_SEED = 49,
    FLX_COL_LAST_SPECULAR = 50, FLX_COL_SHADOW_BLOCKED = 51, FLX_COL_BACKFACE = 52,
    FLX_COL_PIXEL_INDEX = 53, FLX_COL_FIRST_DIFFUSE = 54, FLX_COL_LAST_PDF_DIRECT = 55,
    FLX_COL_LAST_PDF_IMPLICIT = 56, FLX_COL_LAST_COS_TH = 57, FLX_COL_LAST_PICK_PROB = 58,
    FLX_COL_SHADOW_LEN = 59, FLX_COL_HIT_T = 60, FLX_COL_HIT_I = 61,
    FLX_COL_AREA_LIGHT_HIT = 62, FLX_COL_MAT_ID = 63, FLX_NUM_COLS = 64
};

/* queue ids for flx_queue_read() */
enum {
    FLX_Q_RAYGEN = 0, FLX_Q_EXTENSION = 1, FLX_Q_SHADOW = 2, FLX_Q_DIFFUSE = 3,
    FLX_Q_GLOSSY = 4, FLX_Q_GGX_REFL = 5, FLX_Q_GGX_REFR = 6, FLX_Q_DELTA = 7, FLX_NUM_QUEUES = 8
};

#ifdef __cplusplus
}
static_assert(sizeof(flx_vertex) == 48, "Vertex");
static_assert(sizeof(flx_triangle) == 160 && offsetof(flx_triangle, matId) == 144, "Triangle");
static_assert(sizeof(flx_node) == 48 && offsetof(flx_node, parent) == 32 &&
              offsetof(flx_node, iStartOrRight) == 36 && offsetof(flx_node, nPrims) == 40, "GPUNode");
static_assert(sizeof(flx_material) == 80 && offsetof(flx_material, Ns) == 48 &&
              offsetof(flx_material, type) == 68, "Material");
static_assert(sizeof(flx_texdesc) == 12, "TexDescriptor");
static_assert(sizeof(flx_arealight) == 96 && offsetof(flx_arealight, size) == 80, "AreaLight");
static_assert(sizeof(flx_camera) == 80 && offsetof(flx_camera, fov) == 64, "Camera");
static_assert(sizeof(flx_render_params) == 240 && offsetof(flx_render_params, camera) == 96 &&
              offsetof(flx_render_params, exposure) == 176 && offsetof(flx_render_params, width) == 184 &&
              offsetof(flx_render_params, worldRadius) == 228, "RenderParams");
static_assert(sizeof(flx_queue_counters) == 32, "QueueCounters");
#endif

#endif /* FLUCTUS_WIRE_H */
