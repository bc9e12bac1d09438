/*
 * wf_oracle.cpp -- CPU ORACLE for the wavefront (wf_*) path-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The shipped
 * library (fluctus_amd/csrc -> libfluctus_hip.so) never links, loads or calls it.
 *
 * It restates, kernel by kernel and in the reference's own structure, the
 * algorithm of the reference's OpenCL wavefront kernels, operating directly on
 * the reference's path-state layout (GPUTaskState SoA, geom.h:199-236; element
 * (col,gid) at word col*numTasks+gid).  Work-items are executed SEQUENTIALLY in
 * ascending global id, kernels in issue order: that defines the canonical queue
 * order the HIP path reproduces with a stable compaction.
 *
 * Arithmetic: IEEE binary32, no contraction (-ffp-contract=off), transcendental
 * functions from include/flx_math.h (shared arithmetic contract, see its header).
 *
 * Pinning: the reference ships no tests/golden vectors for this path.
 * This oracle is pinned against the reference's own kernels compiled for x86-64
 * (oracle/ref -> oracle/_ref/libfluctus_ref.so, this container only) by
 * tests/test_oracle_vs_ref.py and against tests/golden/ fixtures generated from
 * that build by scripts/make_golden.py.
 *
 * Each function cites the reference file:line it follows.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "../include/fluctus_wire.h"
#include "../include/flx_math.h"

using namespace flx;

namespace {

struct Hit {            /* reference: geom.h:133-142 */
    f3 P, N; f2 uv; float t; int i; int areaLightHit; int matId;
};

struct Ctx {
    uint32_t numTasks = 0;
    std::vector<float> state;             /* 64 * numTasks words */
    std::vector<uint32_t> queues[FLX_NUM_QUEUES];
    flx_queue_counters counters {};
    uint32_t currPixelIdx = 0;
    uint32_t hostPixelIdx = 0;
    flx_render_params params {};
    std::vector<float> pixels;            /* float4 per pixel */
    std::vector<float> preview;
    /* denoiser feature buffers (reference: USE_OPTIX_DENOISER; clcontext.cpp:337-338): accumulators + normalised outputs */
    bool denoiser = false;
    std::vector<float> aovAlbedo, aovNormal, aovAlbedoOut, aovNormalOut;
    /* scene */
    std::vector<flx_triangle> tris;
    std::vector<uint32_t> indices;
    std::vector<flx_node> nodes;
    std::vector<flx_material> materials;
    std::vector<flx_texdesc> texdesc;
    std::vector<uint8_t> texdata;
    /* env map */
    int envW = 1, envH = 1;
    std::vector<float> envRGBA;           /* float4 texels; dummy 1x1 zero (clcontext.cpp:513-518) */
    std::vector<float> probTable, pdfTable;
    std::vector<int> aliasTable;
    /* partition of the framebuffer across ranks (multi-GPU mirror; rank 0 of 1 = reference) */
    uint32_t rank = 0, nranks = 1;
    /* traversal statistics (algorithmic-bytes model) */
    uint64_t stat[6] = {0, 0, 0, 0, 0, 0}; /* ext: rays, inner, tri, hits; shadow: inner, tri */
    uint64_t statShadowRays = 0;
    int threads = 1;
    uint32_t mkStats[4] = {0, 0, 0, 0};   /* RenderStats: primaryRays, extensionRays, shadowRays, samples (geom.h:254-260) */
};

/* --- SoA access (reference: geom.h:38-49) -------------------------------- */
inline float &F(Ctx &c, int col, uint32_t gid) { return c.state[(size_t)col * c.numTasks + gid]; }
inline uint32_t &U(Ctx &c, int col, uint32_t gid) { return reinterpret_cast<uint32_t &>(c.state[(size_t)col * c.numTasks + gid]); }
inline int32_t &I(Ctx &c, int col, uint32_t gid) { return reinterpret_cast<int32_t &>(c.state[(size_t)col * c.numTasks + gid]); }
inline f3 R3(Ctx &c, int col, uint32_t gid) { return mk3(F(c, col, gid), F(c, col + 1, gid), F(c, col + 2, gid)); }
inline void W3(Ctx &c, int col, uint32_t gid, f3 v) { F(c, col, gid) = v.x; F(c, col + 1, gid) = v.y; F(c, col + 2, gid) = v.z; }
inline f3 V(const flx_vec3 &v) { return mk3(v.x, v.y, v.z); }

/* reference: utils.cl:202-225 */
void writeHit(Ctx &c, uint32_t gid, const Hit &h)
{
    W3(c, FLX_COL_P, gid, h.P); W3(c, FLX_COL_N, gid, h.N);
    F(c, FLX_COL_UV, gid) = h.uv.x; F(c, FLX_COL_UV + 1, gid) = h.uv.y;
    F(c, FLX_COL_HIT_T, gid) = h.t; I(c, FLX_COL_HIT_I, gid) = h.i;
    I(c, FLX_COL_AREA_LIGHT_HIT, gid) = h.areaLightHit; I(c, FLX_COL_MAT_ID, gid) = h.matId;
}
Hit readHit(Ctx &c, uint32_t gid)
{
    Hit h;
    h.P = R3(c, FLX_COL_P, gid); h.N = R3(c, FLX_COL_N, gid);
    h.uv = mk2(F(c, FLX_COL_UV, gid), F(c, FLX_COL_UV + 1, gid));
    h.t = F(c, FLX_COL_HIT_T, gid); h.i = I(c, FLX_COL_HIT_I, gid);
    h.areaLightHit = I(c, FLX_COL_AREA_LIGHT_HIT, gid); h.matId = I(c, FLX_COL_MAT_ID, gid);
    return h;
}
Hit emptyHit(float tmax)  /* reference: geom.h:144 EMPTY_HIT */
{
    Hit h; h.P = mk3(0.0f); h.N = mk3(0.0f); h.uv = mk2(0.0f, 0.0f); h.t = tmax; h.i = -1; h.areaLightHit = 0; h.matId = -1;
    return h;
}

/* ------------------------------------------------------------------------ */
/* intersect.cl                                                             */
/* ------------------------------------------------------------------------ */

/* reference: intersect.cl:41-60.  native_recip restated as IEEE 1/x. */
bool intersectAABB(f3 orig, f3 dir, const flx_vec3 &bmin, const flx_vec3 &bmax, float *tminRet, float tMaxPrev)
{
    f3 dinv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    f3 tmp = (V(bmin) - orig) * dinv;
    f3 tmaxv = (V(bmax) - orig) * dinv;
    f3 tminv = min3(tmp, tmaxv);
    tmaxv = max3(tmp, tmaxv);
    float tmin = fmaxf_(fmaxf_(tminv.x, tminv.y), tminv.z);
    float tmax = fminf_(fminf_(tmaxv.x, tmaxv.y), tmaxv.z);
    if (tmax < 0.0f) return false;
    if (tmin > tmax) return false;
    *tminRet = tmin;
    return tmin < tMaxPrev;
}

/* reference: intersect.cl:62-93 (Moller-Trumbore, EPSILON 1e-12) */
bool intersectTriangle(f3 orig, f3 dir, f3 p0, f3 p1, f3 p2, float *tret, float *uret, float *vret)
{
    f3 s1 = p1 - p0;
    f3 s2 = p2 - p0;
    f3 pvec = cross(dir, s2);
    float det = dot(s1, pvec);
    if (absf(det) < 1e-12f) return false;
    float iDet = 1.0f / det;
    f3 tvec = orig - p0;
    float u = dot(tvec, pvec) * iDet;
    if (u < 0.0f || u > 1.0f) return false;
    f3 qvec = cross(tvec, s1);
    float v = dot(dir, qvec) * iDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(s2, qvec) * iDet;
    if (t < 0.0f) return false;
    *tret = t; *uret = u; *vret = v;
    return true;
}

/* reference: intersect.cl:96-121 */
bool intersectTriangleLocal(f3 orig, f3 dir, f3 p0, f3 p1, f3 p2, float *tres)
{
    float t, u, v;
    if (!intersectTriangle(orig, dir, p0, p1, p2, &t, &u, &v)) return false;
    if (t > *tres) return false;
    *tres = t;
    return true;
}

/* reference: intersect.cl:124-155 */
void intersectLight(Hit *hit, f3 orig, f3 dir, const flx_render_params &p)
{
    const flx_arealight &L = p.areaLight;
    if (dot(dir, V(L.N)) > 0.0f) return;
    f3 pos = V(L.pos), right = V(L.right), up = V(L.up);
    f3 tl = pos + L.size.x * right + L.size.y * up;
    f3 tr = pos - L.size.x * right + L.size.y * up;
    f3 bl = pos + L.size.x * right - L.size.y * up;
    f3 br = pos - L.size.x * right - L.size.y * up;
    bool first = intersectTriangleLocal(orig, dir, tl, bl, br, &hit->t);
    bool second = intersectTriangleLocal(orig, dir, tl, br, tr, &hit->t);
    if (first || second) {
        hit->areaLightHit = 1;
        hit->P = orig + hit->t * dir;
        hit->N = V(L.N);
        hit->i = 0;
        hit->matId = 0;
    }
}

/* ------------------------------------------------------------------------ */
/* bvh.cl (stack variant, :232-374)                                          */
/* ------------------------------------------------------------------------ */

/* reference: bvh.cl:234-310 */
void bvh_intersect(Ctx &c, f3 orig, f3 dir, Hit *hit, uint64_t *nInner, uint64_t *nTri)
{
    uint32_t stack[64];
    int sp = 0;
    stack[0] = 0;
    while (sp >= 0) {
        uint32_t ni = stack[sp--];
        const flx_node &n = c.nodes[ni];
        if (n.nPrims != 0) {
            float tmin = FLX_FLT_MAX, umin = 0.0f, vmin = 0.0f;
            int imin = -1;
            for (uint32_t i = n.iStartOrRight; i < n.iStartOrRight + n.nPrims; i++) {
                const flx_triangle &tr = c.tris[c.indices[i]];
                float t, u, v;
                ++*nTri;
                if (intersectTriangle(orig, dir, V(tr.v0.p), V(tr.v1.p), V(tr.v2.p), &t, &u, &v)) {
                    if (t > 0.0f && t < tmin) { imin = (int)i; tmin = t; umin = u; vmin = v; }
                }
            }
            if (imin != -1 && tmin < hit->t) {
                const flx_triangle &tr = c.tris[c.indices[imin]];
                hit->i = (int)c.indices[imin];
                hit->matId = tr.matId;
                hit->t = tmin;
                hit->P = orig + tmin * dir;
                hit->N = normalize(bary(umin, vmin, V(tr.v0.n), V(tr.v1.n), V(tr.v2.n)));
                f3 uv = bary(umin, vmin, V(tr.v0.t), V(tr.v1.t), V(tr.v2.t));
                hit->uv = mk2(uv.x, uv.y);
            }
        } else {
            ++*nInner;
            float lnear = 0.0f, rnear = 0.0f;
            uint32_t left = ni + 1, right = n.iStartOrRight;
            bool lh = intersectAABB(orig, dir, c.nodes[left].bmin, c.nodes[left].bmax, &lnear, hit->t);
            bool rh = intersectAABB(orig, dir, c.nodes[right].bmin, c.nodes[right].bmax, &rnear, hit->t);
            if (lh && rh) {
                uint32_t closer = left, farther = right;
                if (rnear < lnear) std::swap(closer, farther);
                stack[++sp] = farther;
                stack[++sp] = closer;
            } else if (lh) stack[++sp] = left;
            else if (rh) stack[++sp] = right;
        }
    }
}

/* reference: bvh.cl:312-373 */
bool bvh_occluded(Ctx &c, f3 orig, f3 dir, float maxDist, uint64_t *nInner, uint64_t *nTri)
{
    uint32_t stack[64];
    int sp = 0;
    stack[0] = 0;
    while (sp >= 0) {
        uint32_t ni = stack[sp--];
        const flx_node &n = c.nodes[ni];
        if (n.nPrims != 0) {
            for (uint32_t i = n.iStartOrRight; i < n.iStartOrRight + n.nPrims; i++) {
                const flx_triangle &tr = c.tris[c.indices[i]];
                float t, u, v;
                ++*nTri;
                if (intersectTriangle(orig, dir, V(tr.v0.p), V(tr.v1.p), V(tr.v2.p), &t, &u, &v) && t > 0.0f && t < maxDist)
                    return true;
            }
        } else {
            ++*nInner;
            float lnear = 0.0f, rnear = 0.0f;
            uint32_t left = ni + 1, right = n.iStartOrRight;
            bool lh = intersectAABB(orig, dir, c.nodes[left].bmin, c.nodes[left].bmax, &lnear, maxDist);
            bool rh = intersectAABB(orig, dir, c.nodes[right].bmin, c.nodes[right].bmax, &rnear, maxDist);
            if (lh && rh) {
                uint32_t closer = left, farther = right;
                if (rnear < lnear) std::swap(closer, farther);
                stack[++sp] = farther;
                stack[++sp] = closer;
            } else if (lh) stack[++sp] = left;
            else if (rh) stack[++sp] = right;
        }
    }
    return false;
}

/* ------------------------------------------------------------------------ */
/* utils.cl                                                                  */
/* ------------------------------------------------------------------------ */

/* reference: utils.cl:30-33 */
f3 reflect(f3 dir, f3 n) { return dir - 2.0f * dot(dir, n) * n; }

/* reference: utils.cl:36-43 */
f3 refract(f3 wi, f3 n, float eta)
{
    float iDotN = dot(-wi, n);
    float sin2ThetaI = fmaxf_(0.0f, 1.0f - iDotN * iDotN);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    float cosThetaT = sqrtf(fmaxf_(0.0f, 1.0f - sin2ThetaT));
    return wi * eta + n * (eta * iDotN - cosThetaT);
}

/* reference: utils.cl:50-59 */
void makeOrthoBasis(f3 N, f3 *a, f3 *b)
{
    if (N.x != N.y || N.x != N.z) *a = mk3(N.z - N.y, N.x - N.z, N.y - N.x);
    else                          *a = mk3(N.z - N.y, N.x + N.z, -N.y - N.x);
    *a = normalize(*a);
    *b = cross(N, *a);
}

/* reference: utils.cl:75-80 */
f2 uniformSampleDisk(uint32_t *seed)
{
    float sqrt_r = sqrtf(rand01(seed));
    float th = FLX_2PI * rand01(seed);
    float s, co; sincosf_(th, &s, &co);
    return mk2(sqrt_r * co, sqrt_r * s);
}

/* reference: utils.cl:83-112 */
f3 cosSampleHemisphere(f3 n, uint32_t *seed, float *p)
{
    float r1 = 2.0f * FLX_PI * rand01(seed);
    float r2 = rand01(seed);
    float r2s = sqrtf(r2);
    f3 w = n, u;
    if (absf(w.x) > 0.1f) u = cross(mk3(0.0f, 1.0f, 0.0f), w);
    else                  u = cross(mk3(1.0f, 0.0f, 0.0f), w);
    u = normalize(u);
    f3 v = cross(w, u);
    float s, co; sincosf_(r1, &s, &co);
    u = u * (co * r2s);
    v = v * (s * r2s);
    w = w * sqrtf(1.0f - r2);
    f3 dir = u + v + w;
    float costh = dot(n, dir);
    *p = costh / FLX_PI;
    return dir;
}

/* reference: utils.cl:114-133 */
f3 readTexture(Ctx &c, f2 uv, const flx_texdesc &tex)
{
    float ux = uv.x * (float)tex.width, uy = uv.y * (float)tex.height;
    int w = (int)tex.width, h = (int)tex.height;
    int tx = (((int)floorf(ux)) % w + w) % w;
    int ty = (((int)floorf(uy)) % h + h) % h;
    /* (int2)(tx + frac) truncates toward zero; frac in [0,1) so the value is tx (or tx+1 when rounding up) */
    int cx = (int)((float)tx + ux - floorf(ux));
    int cy = (int)((float)ty + uy - floorf(uy));
    cx = cx < 0 ? 0 : (cx > w - 1 ? w - 1 : cx);
    cy = cy < 0 ? 0 : (cy > h - 1 ? h - 1 : cy);
    const uint8_t *pix = c.texdata.data() + tex.offset + (size_t)cx * 4 + (size_t)cy * tex.width * 4;
    return mk3((float)pix[0], (float)pix[1], (float)pix[2]) / 255.0f;
}

/* reference: utils.cl:136-146 */
f3 matGetFloat3(Ctx &c, f3 fallback, f2 uv, int idx) { return idx != -1 ? readTexture(c, uv, c.texdesc[idx]) : fallback; }
f3 matGetAlbedo(Ctx &c, f3 fallback, f2 uv, int idx) { return pow3(matGetFloat3(c, fallback, uv, idx), 2.2f); }

/* reference: utils.cl:149-182 */
f3 tangentSpaceNormal(Ctx &c, const Hit &hit, const flx_material &mat)
{
    if (mat.map_N == -1) return hit.N;
    f3 texNormal = matGetFloat3(c, mk3(0.5f, 0.5f, 1.0f), hit.uv, mat.map_N);
    texNormal = 2.0f * texNormal - mk3(1.0f, 1.0f, 1.0f);
    const flx_triangle &t = c.tris[hit.i];
    f3 e1 = V(t.v1.p) - V(t.v0.p), e2 = V(t.v2.p) - V(t.v0.p);
    f3 t1 = V(t.v1.t) - V(t.v0.t), t2 = V(t.v2.t) - V(t.v0.t);
    float det = t1.x * t2.y - t1.y * t2.x;
    if (det == 0.0f) return hit.N;
    float invDet = 1.0f / det;
    f3 T = normalize(invDet * (e1 * t2.y - e2 * t1.y));
    f3 B = normalize(invDet * (e2 * t1.x - e1 * t2.x));
    f3 N;
    N.x = T.x * texNormal.x + B.x * texNormal.y + hit.N.x * texNormal.z;
    N.y = T.y * texNormal.x + B.y * texNormal.y + hit.N.y * texNormal.z;
    N.z = T.z * texNormal.x + B.z * texNormal.y + hit.N.z * texNormal.z;
    return normalize(N);
}

/* reference: utils.cl:197-200 */
float pdfAtoW(float pdf, float dist, float cosine) { return pdf * (dist * dist) / absf(cosine); }

/* reference: utils.cl:227-236 */
void sampleAreaLight(const flx_arealight &L, float *pdf, f3 *p, uint32_t *seed)
{
    *pdf = 1.0f / (4.0f * L.size.x * L.size.y);
    *p = V(L.pos);
    float r1 = 2.0f * rand01(seed) - 1.0f;
    float r2 = 2.0f * rand01(seed) - 1.0f;
    *p = *p + r1 * L.size.x * V(L.right);
    *p = *p + r2 * L.size.y * V(L.up);
}

/* reference: utils.cl:239-242 */
float luminance(f3 v) { return 0.212671f * v.x + 0.715160f * v.y + 0.072169f * v.z; }

/* ------------------------------------------------------------------------ */
/* env_map.cl                                                                */
/* ------------------------------------------------------------------------ */

/* reference: env_map.cl:14-24 */
f2 directionToUV(f3 dir)
{
    if (dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) return mk2(0.0f, 0.0f);
    float u = 1.0f + atan2f_(dir.x, -dir.z) / FLX_PI;
    float r = clampf(dir.y / length(dir), -1.0f, 1.0f);
    float v = acosf_(r) / FLX_PI;
    return mk2(u * 0.5f, v);
}

/* reference: env_map.cl:28-37 */
f3 UVToDirection(float u, float v)
{
    float phi = v * FLX_PI;
    float theta = (u * 2.0f - 1.0f) * FLX_PI;
    float sinPhi, cosPhi, sinTh, cosTh;
    sincosf_(phi, &sinPhi, &cosPhi);
    sincosf_(theta, &sinTh, &cosTh);
    return mk3(sinPhi * sinTh, cosPhi, -sinPhi * cosTh);
}

/* reference: env_map.cl:10,39-43 -- read_imagef with CLK_NORMALIZED_COORDS_TRUE |
 * CLK_ADDRESS_CLAMP_TO_EDGE | CLK_FILTER_LINEAR, restated per OpenCL 1.2 s8.2:
 * u = s*w, i0 = floor(u-0.5), a = frac(u-0.5), texel indices clamped to the edge. */
f3 evalEnvMapUV(Ctx &c, f2 uv)
{
    int w = c.envW, h = c.envH;
    float u = uv.x * (float)w, v = uv.y * (float)h;
    float fu = u - 0.5f, fv = v - 0.5f;
    float flu = floorf(fu), flv = floorf(fv);
    float a = fu - flu, b = fv - flv;
    int i0 = (int)flu, j0 = (int)flv, i1 = i0 + 1, j1 = j0 + 1;
    i0 = i0 < 0 ? 0 : (i0 > w - 1 ? w - 1 : i0); i1 = i1 < 0 ? 0 : (i1 > w - 1 ? w - 1 : i1);
    j0 = j0 < 0 ? 0 : (j0 > h - 1 ? h - 1 : j0); j1 = j1 < 0 ? 0 : (j1 > h - 1 ? h - 1 : j1);
    const float *T = c.envRGBA.data();
    auto tx = [&](int i, int j) { const float *p = T + ((size_t)j * w + i) * 4; return mk3(p[0], p[1], p[2]); };
    f3 r = (1.0f - a) * (1.0f - b) * tx(i0, j0) + a * (1.0f - b) * tx(i1, j0)
         + (1.0f - a) * b * tx(i0, j1) + a * b * tx(i1, j1);
    return r;
}
f3 evalEnvMapDir(Ctx &c, f3 dir) { return evalEnvMapUV(c, directionToUV(dir)); }

/* reference: env_map.cl:65-92 */
void sampleEnvMapAlias(Ctx &c, float rnd, f3 *L, float *pdfW)
{
    int width = c.envW, height = c.envH;
    float r = rnd * (float)width * (float)height;
    int i = std::min((int)floorf(r), width * height - 1);
    float mProb = c.probTable[i];
    int uvInd = (r - (float)i < mProb) ? i : c.aliasTable[i];
    float pdf_uv = c.pdfTable[uvInd];
    int uInd = uvInd % width, vInd = uvInd / width;
    float u = ((float)uInd + 0.5f) / (float)width;
    float v = ((float)vInd + 0.5f) / (float)height;
    *L = UVToDirection(u, v);
    float sinTh = sinf_(FLX_PI * v);
    float directPdfUV = pdf_uv * 1.0f;
    if (sinTh != 0.0f) *pdfW = directPdfUV / (2.0f * FLX_PI * FLX_PI * sinTh);
    else               *pdfW = 0.0f;
}

/* reference: env_map.cl:95-107 */
float envMapPdf(Ctx &c, f3 direction)
{
    int width = c.envW, height = c.envH;
    f2 uv = directionToUV(direction);
    float sinTh = sinf_(uv.y * FLX_PI);
    if (sinTh == 0.0f) return 0.0f;
    int iu = std::min((int)floorf(uv.x * (float)width), width - 1);
    int iv = std::min((int)floorf(uv.y * (float)height), height - 1);
    return c.pdfTable[iv * width + iu] / (FLX_2PI * FLX_PI * sinTh);
}

/* ------------------------------------------------------------------------ */
/* BSDFs                                                                     */
/* ------------------------------------------------------------------------ */

/* reference: fresnel.cl:5-20 */
float fresnelDielectric(float cosThI, float etaI, float etaT)
{
    float sinThetaI = sqrtf(fmaxf_(0.0f, 1.0f - cosThI * cosThI));
    float sinThetaT = etaI / etaT * sinThetaI;
    float cosThetaT = sqrtf(fmaxf_(0.0f, 1.0f - sinThetaT * sinThetaT));
    if (sinThetaT >= 1.0f) return 1.0f;
    float parl = ((etaT * cosThI) - (etaI * cosThetaT)) / ((etaT * cosThI) + (etaI * cosThetaT));
    float perp = ((etaI * cosThI) - (etaT * cosThetaT)) / ((etaI * cosThI) + (etaT * cosThetaT));
    return 0.5f * (parl * parl + perp * perp);
}

/* reference: diffuse.cl:9-26 */
f3 sampleDiffuse(Ctx &c, const Hit &hit, const flx_material &mat, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    *dirOut = cosSampleHemisphere(hit.N, seed, pdfW);
    f3 Kd = matGetAlbedo(c, V(mat.Kd), hit.uv, mat.map_Kd);
    return Kd * FLX_INV_PI;
}
f3 evalDiffuse(Ctx &c, const Hit &hit, const flx_material &mat)
{
    f3 Kd = matGetAlbedo(c, V(mat.Kd), hit.uv, mat.map_Kd);
    return Kd * FLX_INV_PI;
}
float pdfDiffuse(const Hit &hit, f3 dirOut) { return dot(hit.N, dirOut) * FLX_INV_PI; }

/* reference: ggx.cl:12-15 */
float toRoughness(float shininess) { return sqrtf(2.0f / (2.0f + shininess)); }

/* reference: ggx.cl:19-36 (native_sin/native_cos restated as sin/cos) */
f3 ggxSampleLobe(float alpha, f3 N, uint32_t *seed)
{
    f3 X, Y, Z = N;
    makeOrthoBasis(Z, &X, &Y);
    float rx = rand01(seed);
    float ry = rand01(seed);
    float theta = atan2f_(alpha * sqrtf(rx), sqrtf(1.0f - rx));
    float phi = FLX_2PI * ry;
    float sinTheta, cosTheta, sinPhi, cosPhi;
    sincosf_(theta, &sinTheta, &cosTheta);
    sincosf_(phi, &sinPhi, &cosPhi);
    return normalize(X * sinTheta * cosPhi + Y * sinTheta * sinPhi + Z * cosTheta);
}

/* reference: ggx.cl:40-53 */
float ggxG1(float alpha, f3 v, f3 n, f3 m)
{
    float mDotV = dot(m, v), nDotV = dot(n, v);
    if (nDotV * mDotV <= 0.0f) return 0.0f;
    float cosThSq = nDotV * nDotV;
    float tanSq = (cosThSq > 0.0f) ? ((1.0f - cosThSq) / cosThSq) : 0.0f;
    return 2.0f / (1.0f + sqrtf(1.0f + alpha * alpha * tanSq));
}
/* reference: ggx.cl:56-60 */
float ggxG(float alpha, f3 dirIn, f3 dirOut, f3 n, f3 m) { return ggxG1(alpha, dirIn, n, m) * ggxG1(alpha, dirOut, n, m); }

/* reference: ggx.cl:64-78 */
float ggxD(float alpha, f3 n, f3 m)
{
    float nDotM = dot(n, m);
    if (nDotM <= 0.0f) return 0.0f;
    float nDotMSq = nDotM * nDotM;
    float tanSq = nDotM != 0.0f ? ((1.0f - nDotMSq) / nDotMSq) : 0.0f;
    float aSq = alpha * alpha;
    float denom = FLX_PI * nDotMSq * nDotMSq * (aSq + tanSq) * (aSq + tanSq);
    return denom > 0.0f ? (aSq / denom) : 0.0f;
}

/* reference: ggx.cl:81-87 */
float ggxPdfReflect(float alpha, f3 dirOut, f3 N, f3 H)
{
    float nDotH = absf(dot(N, H));
    float oDotH = absf(dot(dirOut, H));
    float jInv = 4.0f * oDotH;
    return jInv == 0.0f ? 0.0f : ggxD(alpha, N, H) * nDotH / jInv;
}

/* reference: ggx.cl:89-113 */
f3 sampleGGXReflect(Ctx &c, const Hit &hit, const flx_material &mat, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    dirIn = dirIn * -1.0f;
    float alpha = toRoughness(mat.Ns);
    f3 H = ggxSampleLobe(alpha, hit.N, seed);
    *dirOut = reflect(-dirIn, H);
    *pdfW = ggxPdfReflect(alpha, *dirOut, hit.N, H);
    float iDotN = dot(dirIn, hit.N);
    float oDotN = dot(*dirOut, hit.N);
    float Fr = (mat.Ni > 1.0f) ? fresnelDielectric(iDotN, 1.0f, mat.Ni) : 1.0f;
    f3 Ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
    float D = ggxD(alpha, hit.N, H);
    float G = ggxG(alpha, dirIn, *dirOut, hit.N, H);
    float den = 4.0f * iDotN * oDotN;
    return (den != 0.0f) ? (Ks * Fr * G * D / den) : mk3(0.0f);
}

/* reference: ggx.cl:115-136 */
f3 evalGGXReflect(Ctx &c, const Hit &hit, const flx_material &mat, f3 dirIn, f3 dirOut)
{
    dirIn = dirIn * -1.0f;
    float alpha = toRoughness(mat.Ns);
    f3 H = normalize(dirIn + dirOut);
    float iDotN = dot(dirIn, hit.N);
    float oDotN = dot(dirOut, hit.N);
    float Fr = (mat.Ni > 1.0f) ? fresnelDielectric(iDotN, 1.0f, mat.Ni) : 1.0f;
    f3 Ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
    float D = ggxD(alpha, hit.N, H);
    float G = ggxG(alpha, dirIn, dirOut, hit.N, H);
    float den = 4.0f * iDotN * oDotN;
    return (den != 0.0f) ? (Ks * Fr * G * D / den) : mk3(0.0f);
}

/* reference: ggx.cl:138-144 */
float pdfGGXReflect(const Hit &hit, const flx_material &mat, f3 dirIn, f3 dirOut)
{
    dirIn = dirIn * -1.0f;
    float alpha = toRoughness(mat.Ns);
    f3 H = normalize(dirIn + dirOut);
    return ggxPdfReflect(alpha, dirOut, hit.N, H);
}

/* reference: ggx.cl:147-154 */
float ggxPdfRefract(float alpha, float etaI, float etaO, f3 dirIn, f3 dirOut, f3 N, f3 H)
{
    float nDotH = absf(dot(N, H));
    float iDotH = absf(dot(dirIn, H));
    float oDotH = absf(dot(dirOut, H));
    float sqrtJInv = etaI * iDotH + etaO * oDotH;
    return sqrtJInv == 0.0f ? 0.0f : ggxD(alpha, N, H) * nDotH * oDotH * etaO * etaO / (sqrtJInv * sqrtJInv);
}

/* reference: ggx.cl:156-221 */
f3 sampleGGXRefract(Ctx &c, const Hit &hit, const flx_material &mat, bool backface, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    dirIn = dirIn * -1.0f;
    float raylen = length(dirIn);
    float alpha = toRoughness(mat.Ns);
    float etaI = 1.0f, etaO = mat.Ni;
    if (backface) std::swap(etaI, etaO);
    float iDotN = dot(normalize(dirIn), hit.N);
    f3 H = ggxSampleLobe(alpha, hit.N, seed);
    float Fr = fresnelDielectric(iDotN, etaI, etaO);
    if (rand01(seed) < Fr) {
        *dirOut = raylen * reflect(normalize(-dirIn), H);
        *pdfW = ggxPdfReflect(alpha, *dirOut, hit.N, H);
        float oDotN = dot(*dirOut, hit.N);
        float D = ggxD(alpha, hit.N, H);
        float G = ggxG(alpha, dirIn, *dirOut, hit.N, H);
        float den = 4.0f * iDotN * oDotN;
        return (den != 0.0f) ? mk3(Fr * G * D / den) : mk3(0.0f);
    } else {
        float eta = etaI / etaO;
        *dirOut = raylen * refract(normalize(-dirIn), hit.N, eta);
        H = normalize(-(dirIn * etaI + *dirOut * etaO));
        f3 Nn = backface ? -hit.N : hit.N;
        *pdfW = ggxPdfRefract(alpha, etaI, etaO, dirIn, *dirOut, Nn, H);
        f3 bsdf = mk3(eta * eta);
        f3 Ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
        bsdf = bsdf * Ks;
        float iDotH = absf(dot(normalize(dirIn), H));
        float oDotH = absf(dot(*dirOut, H));
        float oDotN = dot(*dirOut, hit.N);
        float focusTermDenom = iDotN * oDotN * (etaI * iDotH + etaO * oDotH) * (etaI * iDotH + etaO * oDotH);
        if (focusTermDenom == 0.0f) return mk3(0.0f);
        float focusTerm = etaO * etaO * iDotH * oDotH / focusTermDenom;
        float D = ggxD(alpha, Nn, H);
        float G = ggxG(alpha, dirIn, *dirOut, Nn, H);
        return (1.0f - Fr) * bsdf * D * G * focusTerm;
    }
}

/* reference: ggx.cl:223-271 */
f3 evalGGXRefract(Ctx &c, const Hit &hit, const flx_material &mat, bool backface, f3 dirIn, f3 dirOut)
{
    dirIn = dirIn * -1.0f;
    float alpha = toRoughness(mat.Ns);
    float etaI = 1.0f, etaO = mat.Ni;
    if (backface) std::swap(etaI, etaO);
    float iDotN = dot(normalize(dirIn), hit.N);
    float oDotN = dot(normalize(dirOut), hit.N);
    float Fr = fresnelDielectric(iDotN, etaI, etaO);
    if (!backface) {
        f3 H = normalize(dirIn + dirOut);
        float D = ggxD(alpha, hit.N, H);
        float G = ggxG(alpha, dirIn, dirOut, hit.N, H);
        float den = 4.0f * iDotN * oDotN;
        return (den != 0.0f) ? mk3(Fr * G * D / den) : mk3(0.0f);
    } else {
        f3 H = normalize(-(dirIn * etaI + dirOut * etaO));
        float eta = etaI / etaO;
        f3 bsdf = mk3(eta * eta);
        f3 Ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
        bsdf = bsdf * Ks;
        float iDotH = absf(dot(normalize(dirIn), H));
        float oDotH = absf(dot(normalize(dirOut), H));
        float focusTermDenom = iDotN * oDotN * (etaI * iDotH + etaO * oDotH) * (etaI * iDotH + etaO * oDotH);
        if (focusTermDenom == 0.0f) return mk3(0.0f);
        float focusTerm = etaO * etaO * iDotH * oDotH / focusTermDenom;
        float D = ggxD(alpha, -hit.N, H);
        float G = ggxG(alpha, dirIn, dirOut, -hit.N, H);
        return (1.0f - Fr) * bsdf * D * G * focusTerm;
    }
}

/* reference: ggx.cl:273-292 */
float pdfGGXRefract(const Hit &hit, const flx_material &mat, bool backface, f3 dirIn, f3 dirOut)
{
    dirIn = dirIn * -1.0f;
    float alpha = toRoughness(mat.Ns);
    float etaI = 1.0f, etaO = mat.Ni;
    if (!backface) {
        f3 H = normalize(dirIn + dirOut);
        return ggxPdfReflect(alpha, dirOut, hit.N, H);
    } else {
        std::swap(etaI, etaO);
        f3 H = normalize(-(dirIn * etaI + dirOut * etaO));
        return ggxPdfRefract(alpha, etaI, etaO, dirIn, dirOut, -hit.N, H);
    }
}

/* reference: glossy.cl:12-22 */
f3 etaToKs(float eta) { float r = (eta > 0.0f) ? ((eta - 1.0f) / (eta + 1.0f)) : 0.0f; return mk3(r * r); }
float ksToEta(f3 Ks)
{
    float k = clampf((Ks.x + Ks.y + Ks.z) / 3.0f, 0.0f, 0.99f);
    return (sqrtf(k) + 1.0f) / (1.0f - sqrtf(k));
}

/* reference: glossy.cl:24-64.  NB: when dot(N,dirOut) < 1e-5 the reference returns
 * without writing *pdfW (uninitialised in the caller); we define it as 0. */
f3 sampleGlossy(Ctx &c, const Hit &hit, const flx_material &mat, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    flx_material m = mat;
    f3 Ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
    m.Ni = (mat.Ni > 0.0f) ? mat.Ni : ksToEta(Ks);
    if (is_zero(Ks)) Ks = etaToKs(m.Ni);
    m.Ks.x = Ks.x; m.Ks.y = Ks.y; m.Ks.z = Ks.z;
    float cosTh = dot(normalize(-dirIn), hit.N);
    float Fr = fresnelDielectric(cosTh, 1.0f, m.Ni);
    float basePdf, coatingPdf;
    f3 baseBrdf, coatingBrdf;
    if (rand01(seed) < Fr) {
        coatingBrdf = sampleGGXReflect(c, hit, m, dirIn, dirOut, &coatingPdf, seed);
        baseBrdf = evalDiffuse(c, hit, m);
        basePdf = pdfDiffuse(hit, *dirOut);
    } else {
        baseBrdf = sampleDiffuse(c, hit, m, dirOut, &basePdf, seed);
        coatingBrdf = evalGGXReflect(c, hit, m, dirIn, *dirOut);
        coatingPdf = pdfGGXReflect(hit, m, dirIn, *dirOut);
    }
    if (dot(hit.N, *dirOut) < 1e-5f) return mk3(0.0f);
    *pdfW = (1.0f - Fr) * basePdf + Fr * coatingPdf;
    return baseBrdf * (1.0f - Fr) + coatingBrdf;
}

/* reference: glossy.cl:66-85 */
f3 evalGlossy(Ctx &c, const Hit &hit, const flx_material &mat, f3 dirIn, f3 dirOut)
{
    flx_material m = mat;
    f3 Ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
    m.Ni = (mat.Ni > 0.0f) ? mat.Ni : ksToEta(Ks);
    if (length(Ks) == 0.0f) Ks = etaToKs(m.Ni);
    m.Ks.x = Ks.x; m.Ks.y = Ks.y; m.Ks.z = Ks.z;
    f3 baseBrdf = evalDiffuse(c, hit, m);
    f3 coatingBrdf = evalGGXReflect(c, hit, m, dirIn, dirOut);
    float cosTh = dot(normalize(-dirIn), hit.N);
    float Fr = fresnelDielectric(cosTh, 1.0f, m.Ni);
    return baseBrdf * (1.0f - Fr) + coatingBrdf;
}

/* reference: glossy.cl:87-101 */
float pdfGlossy(Ctx &c, const Hit &hit, const flx_material &mat, f3 dirIn, f3 dirOut)
{
    f3 Ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
    float Ni = (mat.Ni > 0.0f) ? mat.Ni : ksToEta(Ks);
    float basePdf = pdfDiffuse(hit, dirOut);
    float coatingPdf = pdfGGXReflect(hit, mat, dirIn, dirOut);
    float cosTh = dot(normalize(-dirIn), hit.N);
    float Fr = fresnelDielectric(cosTh, 1.0f, Ni);
    return (1.0f - Fr) * basePdf + Fr * coatingPdf;
}

/* reference: ideal_reflection.cl:9-22 */
f3 sampleIdealReflection(Ctx &c, const Hit &hit, const flx_material &mat, f3 dirIn, f3 *dirOut, float *pdfW)
{
    float len = length(dirIn);
    *dirOut = len * reflect(normalize(dirIn), hit.N);
    *pdfW = 1.0f;
    f3 ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
    float cosO = dot(normalize(*dirOut), hit.N);
    return (cosO != 0.0f) ? ks / cosO : mk3(0.0f);
}

/* reference: ideal_dielectric.cl:10-45 */
f3 sampleIdealDielectric(Ctx &c, const Hit &hit, const flx_material &mat, bool backface, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    float raylen = length(dirIn);
    f3 bsdf = mk3(1.0f);
    float cosI = dot(normalize(-dirIn), hit.N);
    float n1 = 1.0f, n2 = mat.Ni;
    if (backface) std::swap(n1, n2);
    float eta = n1 / n2;
    float fr = fresnelDielectric(cosI, n1, n2);
    if (rand01(seed) < fr) {
        *dirOut = raylen * reflect(normalize(dirIn), hit.N);
    } else {
        *dirOut = raylen * refract(normalize(dirIn), hit.N, eta);
        bsdf = bsdf * (eta * eta);
        f3 Ks = matGetFloat3(c, V(mat.Ks), hit.uv, mat.map_Ks);
        bsdf = bsdf * Ks;
    }
    *pdfW = 1.0f;
    float cosO = dot(normalize(*dirOut), hit.N);
    return bsdf / cosO;
}

/* reference: bxdf_partial.cl:19-61 */
f3 bxdfSample(Ctx &c, const Hit &hit, const flx_material &mat, bool backface, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    switch (mat.type) {
    case FLX_BXDF_DIFFUSE:              return sampleDiffuse(c, hit, mat, dirOut, pdfW, seed);
    case FLX_BXDF_GLOSSY:               return sampleGlossy(c, hit, mat, dirIn, dirOut, pdfW, seed);
    case FLX_BXDF_GGX_ROUGH_REFLECTION: return sampleGGXReflect(c, hit, mat, dirIn, dirOut, pdfW, seed);
    case FLX_BXDF_IDEAL_REFLECTION:     return sampleIdealReflection(c, hit, mat, dirIn, dirOut, pdfW);
    case FLX_BXDF_GGX_ROUGH_DIELECTRIC: return sampleGGXRefract(c, hit, mat, backface, dirIn, dirOut, pdfW, seed);
    case FLX_BXDF_IDEAL_DIELECTRIC:     return sampleIdealDielectric(c, hit, mat, backface, dirIn, dirOut, pdfW, seed);
    }
    return mk3(0.0f);
}
/* reference: bxdf_partial.cl:64-106 */
f3 bxdfEval(Ctx &c, const Hit &hit, const flx_material &mat, bool backface, f3 dirIn, f3 dirOut)
{
    switch (mat.type) {
    case FLX_BXDF_DIFFUSE:              return evalDiffuse(c, hit, mat);
    case FLX_BXDF_GLOSSY:               return evalGlossy(c, hit, mat, dirIn, dirOut);
    case FLX_BXDF_GGX_ROUGH_REFLECTION: return evalGGXReflect(c, hit, mat, dirIn, dirOut);
    case FLX_BXDF_GGX_ROUGH_DIELECTRIC: return evalGGXRefract(c, hit, mat, backface, dirIn, dirOut);
    }
    return mk3(0.0f);
}
/* reference: bxdf_partial.cl:109-151 */
float bxdfPdf(Ctx &c, const Hit &hit, const flx_material &mat, bool backface, f3 dirIn, f3 dirOut)
{
    switch (mat.type) {
    case FLX_BXDF_DIFFUSE:              return pdfDiffuse(hit, dirOut);
    case FLX_BXDF_GLOSSY:               return pdfGlossy(c, hit, mat, dirIn, dirOut);
    case FLX_BXDF_GGX_ROUGH_REFLECTION: return pdfGGXReflect(hit, mat, dirIn, dirOut);
    case FLX_BXDF_GGX_ROUGH_DIELECTRIC: return pdfGGXRefract(hit, mat, backface, dirIn, dirOut);
    }
    return 0.0f;
}

/* ------------------------------------------------------------------------ */
/* kernels                                                                   */
/* ------------------------------------------------------------------------ */

/* add_float4 (utils.cl:341-358) on a host buffer */
inline void addFloat4(float *px, f3 v, float w, bool atomic)
{
    if (!atomic) { px[0] += v.x; px[1] += v.y; px[2] += v.z; px[3] += w; return; }
#pragma omp atomic
    px[0] += v.x;
#pragma omp atomic
    px[1] += v.y;
#pragma omp atomic
    px[2] += v.z;
#pragma omp atomic
    px[3] += w;
}
/* first-hit normal in camera space: rotation rows right, up, -dir (wf_logic.cl:189-196, mk_next_vertex.cl:63-67) */
inline f3 cameraSpaceNormal(const flx_render_params &p, f3 N)
{
    f3 r1 = V(p.camera.right), r2 = V(p.camera.up), r3 = V(p.camera.dir) * -1.0f;
    return mk3(dot(r1, N), dot(r2, N), dot(r3, N));
}

/* reference: wf_reset.cl:5-66; launch range clcontext.cpp:765-770 */
void k_reset(Ctx &c)
{
    const flx_render_params &p = c.params;
    uint32_t npix = p.width * p.height;
    uint32_t n = std::max(c.numTasks, npix);
    for (uint32_t gid = 0; gid < n; gid++) {
        if (gid < npix) {
            float *px = &c.pixels[(size_t)gid * 4]; px[0] = px[1] = px[2] = px[3] = 0.0f;
            float *nr = &c.aovNormal[(size_t)gid * 4]; nr[0] = nr[1] = nr[2] = nr[3] = 0.0f;             /* wf_reset.cl:22 */
            float *al = &c.aovAlbedo[(size_t)gid * 4]; al[0] = al[1] = al[2] = 0.1f; al[3] = 0.0f;       /* wf_reset.cl:23-24 */
        }
        if (gid >= c.numTasks) continue;
        W3(c, FLX_COL_EI, gid, mk3(0.0f));
        W3(c, FLX_COL_T, gid, mk3(1.0f));
        U(c, FLX_COL_PATH_LEN, gid) = 0;
        U(c, FLX_COL_LAST_SPECULAR, gid) = 1;
        F(c, FLX_COL_LAST_PDF_W, gid) = 1.0f;
        F(c, FLX_COL_LAST_PDF_DIRECT, gid) = 0.0f;
        F(c, FLX_COL_LAST_PDF_IMPLICIT, gid) = 0.0f;
        F(c, FLX_COL_LAST_COS_TH, gid) = 0.0f;
        F(c, FLX_COL_LAST_PICK_PROB, gid) = 1.0f;
        F(c, FLX_COL_SHADOW_LEN, gid) = 2.0f * p.worldRadius;
        U(c, FLX_COL_BACKFACE, gid) = 0;
        U(c, FLX_COL_SHADOW_BLOCKED, gid) = 1;
        U(c, FLX_COL_PIXEL_INDEX, gid) = 0;
        U(c, FLX_COL_FIRST_DIFFUSE, gid) = 0;
        W3(c, FLX_COL_LAST_EMISSION, gid, mk3(0.0f));
        W3(c, FLX_COL_LAST_BSDF, gid, mk3(0.0f));
        writeHit(c, gid, emptyHit(FLX_FLT_MAX));
        U(c, FLX_COL_SEED, gid) = gid;
        c.queues[FLX_Q_RAYGEN][gid] = gid;
        if (gid == 0) c.counters.raygenQueue = c.numTasks;
    }
}

/* reference: wf_raygen.cl:4-97.
 * Multi-GPU extension (not in the reference): with nranks > 1 the cursor runs over the
 * rank's local pixels and local index p maps to global pixel p*nranks + rank. */
void k_raygen(Ctx &c)
{
    const flx_render_params &p = c.params;
    uint32_t qlen = c.counters.raygenQueue;
    uint32_t numPixelsGlobal = p.width * p.height;
    uint32_t numPixels = (numPixelsGlobal - c.rank + c.nranks - 1) / c.nranks;
    const uint32_t extBase = c.counters.extensionQueue;
#pragma omp parallel for schedule(static) num_threads(c.threads)
    for (int64_t gd = 0; gd < (int64_t)std::min(qlen, c.numTasks); gd++) {
        uint32_t gid = c.queues[FLX_Q_RAYGEN][gd];
        uint32_t seed = U(c, FLX_COL_SEED, gid);
        uint32_t localIdx = (c.currPixelIdx + gd) % numPixels;
        uint32_t pixelIdx = localIdx * c.nranks + c.rank;
        U(c, FLX_COL_PIXEL_INDEX, gid) = localIdx;
        float x = (float)(pixelIdx % p.width);
        float y = (float)(pixelIdx / p.width);
        x += rand01(&seed);
        y += rand01(&seed);
        float NDCx = x / (float)p.width;
        float NDCy = y / (float)p.height;
        float SCRx = 2.0f * NDCx - 1.0f;
        float SCRy = 2.0f * NDCy - 1.0f;
        SCRx *= (float)p.width / (float)p.height;
        float scale = tanf_(0.5f * p.camera.fov * FLX_PI / 180.0f);
        SCRx *= scale;
        SCRy *= scale;
        f3 rayOrig = V(p.camera.pos);
        f3 rayTarget = rayOrig + V(p.camera.right) * SCRx + V(p.camera.up) * SCRy + V(p.camera.dir);
        f3 rayDirection = normalize(rayTarget - rayOrig);
        f3 fp = V(p.camera.pos) + rayDirection * p.camera.focalDist;
        f2 rnd = uniformSampleDisk(&seed);
        rayOrig = rayOrig + p.worldRadius * p.camera.apertureSize * (V(p.camera.right) * rnd.x + V(p.camera.up) * rnd.y);
        rayDirection = normalize(fp - rayOrig);
        W3(c, FLX_COL_ORIG, gid, rayOrig);
        W3(c, FLX_COL_DIR, gid, rayDirection);
        c.queues[FLX_Q_EXTENSION][extBase + (uint32_t)gd] = gid;   /* slot = old counter value, gid order */
        U(c, FLX_COL_SEED, gid) = seed;
        W3(c, FLX_COL_EI, gid, mk3(0.0f));
        W3(c, FLX_COL_T, gid, mk3(1.0f));
        U(c, FLX_COL_PATH_LEN, gid) = 0;
        U(c, FLX_COL_FIRST_DIFFUSE, gid) = 0;
        U(c, FLX_COL_LAST_SPECULAR, gid) = 1;
        F(c, FLX_COL_LAST_PDF_W, gid) = 1.0f;
        F(c, FLX_COL_LAST_PDF_DIRECT, gid) = 0.0f;
        F(c, FLX_COL_LAST_PDF_IMPLICIT, gid) = 0.0f;
        F(c, FLX_COL_LAST_COS_TH, gid) = 0.0f;
        F(c, FLX_COL_LAST_PICK_PROB, gid) = 1.0f;
        F(c, FLX_COL_SHADOW_LEN, gid) = 2.0f * p.worldRadius;
        U(c, FLX_COL_BACKFACE, gid) = 0;
        U(c, FLX_COL_SHADOW_BLOCKED, gid) = 1;
        W3(c, FLX_COL_LAST_EMISSION, gid, mk3(0.0f));
        W3(c, FLX_COL_LAST_BSDF, gid, mk3(0.0f));
        writeHit(c, gid, emptyHit(FLX_FLT_MAX));
    }
    c.counters.extensionQueue = extBase + std::min(qlen, c.numTasks);
}

/* reference: wf_extrays.cl:5-36 */
void k_extend(Ctx &c)
{
    const flx_render_params &p = c.params;
    uint32_t qlen = c.counters.extensionQueue;
    uint64_t sInner = 0, sTri = 0, sHit = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : sInner, sTri, sHit) num_threads(c.threads)
    for (int64_t gd = 0; gd < (int64_t)qlen; gd++) {
        uint32_t gid = c.queues[FLX_Q_EXTENSION][gd];
        f3 orig = R3(c, FLX_COL_ORIG, gid), dir = R3(c, FLX_COL_DIR, gid);
        Hit hit = emptyHit(FLX_FLT_MAX);
        uint64_t a = 0, b = 0;
        bvh_intersect(c, orig, dir, &hit, &a, &b);
        sInner += a; sTri += b; sHit += (hit.i >= 0);
        if (p.sampleImpl && p.useAreaLight) intersectLight(&hit, orig, dir, p);
        U(c, FLX_COL_PATH_LEN, gid) += 1;
        writeHit(c, gid, hit);
    }
    c.stat[0] += qlen; c.stat[1] += sInner; c.stat[2] += sTri; c.stat[3] += sHit;
}

/* reference: wf_shadowrays.cl:6-38 */
void k_shadow(Ctx &c)
{
    const flx_render_params &p = c.params;
    uint32_t qlen = c.counters.shadowQueue;
    uint64_t sInner = 0, sTri = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : sInner, sTri) num_threads(c.threads)
    for (int64_t gd = 0; gd < (int64_t)qlen; gd++) {
        uint32_t gid = c.queues[FLX_Q_SHADOW][gd];
        f3 orig = R3(c, FLX_COL_SHADOW_ORIG, gid), dir = R3(c, FLX_COL_SHADOW_DIR, gid);
        float lenL = F(c, FLX_COL_SHADOW_LEN, gid);
        Hit hitL = emptyHit(lenL);
        if (p.useAreaLight) intersectLight(&hitL, orig, dir, p);
        uint64_t a = 0, b = 0;
        bool occluded = (hitL.i > -1) || bvh_occluded(c, orig, dir, lenL, &a, &b);
        sInner += a; sTri += b;
        U(c, FLX_COL_SHADOW_BLOCKED, gid) = occluded ? 1u : 0u;
    }
    c.statShadowRays += qlen; c.stat[4] += sInner; c.stat[5] += sTri;
}

/* Queue appends.  Work-items run in ascending gid; with threads > 1 the range is cut into
 * contiguous chunks, each chunk appends to private lists, and the lists are concatenated in chunk
 * order afterwards -- the result is the sequential (canonical) queue order for any thread count. */
struct Appender {
    std::vector<uint32_t> l[FLX_NUM_QUEUES];
    void push(int q, uint32_t gid) { l[q].push_back(gid); }
};
void mergeAppenders(Ctx &c, std::vector<Appender> &apps)
{
    uint32_t *len[FLX_NUM_QUEUES] = {&c.counters.raygenQueue, &c.counters.extensionQueue, &c.counters.shadowQueue, &c.counters.diffuseQueue,
                                     &c.counters.glossyQueue, &c.counters.ggxReflQueue, &c.counters.ggxRefrQueue, &c.counters.deltaQueue};
    for (auto &a : apps)
        for (int q = 0; q < FLX_NUM_QUEUES; q++)
            for (uint32_t g : a.l[q]) c.queues[q][(*len[q])++] = g;
}

/* reference: wf_logic.cl:322-372 (addToMaterialQueueNaive; WF_SINGLE_MAT_QUEUE iff !wfSeparateQueues,
 * kernel_impl.hpp:49-67) */
void addToMaterialQueue(Ctx &c, Appender &app, uint32_t gid, const flx_material &mat)
{
    int q;
    if (!c.params.wfSeparateQueues) q = FLX_Q_DIFFUSE;
    else switch (mat.type) {
        case FLX_BXDF_DIFFUSE:              q = FLX_Q_DIFFUSE;  break;
        case FLX_BXDF_GLOSSY:               q = FLX_Q_GLOSSY;   break;
        case FLX_BXDF_GGX_ROUGH_REFLECTION: q = FLX_Q_GGX_REFL; break;
        case FLX_BXDF_GGX_ROUGH_DIELECTRIC: q = FLX_Q_GGX_REFR; break;
        case FLX_BXDF_IDEAL_REFLECTION:
        case FLX_BXDF_IDEAL_DIELECTRIC:     q = FLX_Q_DELTA;    break;
        default: return;
    }
    app.push(q, gid);
}

/* reference: wf_logic.cl:14-314 */
void k_logic(Ctx &c, int firstIteration)
{
    const flx_render_params &p = c.params;
    uint32_t maxId = firstIteration ? std::min(p.width * p.height, c.numTasks) : c.numTasks;
    const int nthr = c.threads;
    std::vector<Appender> apps(nthr);
    const uint32_t chunk = (maxId + nthr - 1) / nthr;
#pragma omp parallel for schedule(static, 1) num_threads(nthr)
    for (int tchunk = 0; tchunk < nthr; tchunk++) {
    Appender &app = apps[tchunk];
    const uint32_t g0 = (uint32_t)tchunk * chunk, g1 = std::min(maxId, g0 + chunk);
    for (uint32_t gid = g0; gid < g1; gid++) {
        uint32_t seed = U(c, FLX_COL_SEED, gid);
        uint32_t len = U(c, FLX_COL_PATH_LEN, gid);
        Hit hit = readHit(c, gid);
        f3 rayOrig = R3(c, FLX_COL_ORIG, gid), rayDir = R3(c, FLX_COL_DIR, gid);
        f3 T = R3(c, FLX_COL_T, gid);

        /* :60-69 russian roulette */
        bool terminate = (len >= p.maxBounces + 1);
        if (terminate && p.useRoulette) {
            float contProb = clampf(luminance(T), 0.01f, 0.5f);
            terminate = (rand01(&seed) > contProb);
            T = T / contProb;
            W3(c, FLX_COL_T, gid, T);
        }
        /* :72-73 */
        if (is_zero(T) || F(c, FLX_COL_LAST_PDF_W, gid) == 0.0f) terminate = true;

        /* :84-107 implicit environment sample */
        if (hit.i < 0 && !terminate) {
            float weight = 1.0f;
            bool lastSpecular = U(c, FLX_COL_LAST_SPECULAR, gid) != 0;
            f3 bg = mk3(0.0f);
            if (p.useEnvMap && (len == 1 || p.sampleImpl))
                bg = evalEnvMapDir(c, rayDir) * p.envMapStrength;
            if (p.sampleImpl && p.sampleExpl && p.useEnvMap && len > 1 && !lastSpecular) {
                float lightPickProb = F(c, FLX_COL_LAST_PICK_PROB, gid);
                float directPdfW = envMapPdf(c, rayDir);
                float actualPdfW = F(c, FLX_COL_LAST_PDF_W, gid);
                weight = (actualPdfW * lightPickProb) / (actualPdfW * lightPickProb + directPdfW);
            }
            f3 newEi = R3(c, FLX_COL_EI, gid) + weight * T * bg;
            W3(c, FLX_COL_EI, gid, newEi);
            terminate = true;
        }
        /* :111-131 implicit area light sample (compiled iff useAreaLight) */
        else if (p.useAreaLight && hit.areaLightHit && !terminate) {
            float misWeight = 1.0f;
            bool lastSpecular = U(c, FLX_COL_LAST_SPECULAR, gid) != 0;
            if (p.sampleExpl && len > 1 && !lastSpecular) {
                float directPdfA = 1.0f / (4.0f * p.areaLight.size.x * p.areaLight.size.y);
                float directPdfW = pdfAtoW(directPdfA, length(hit.P - rayOrig), dot(normalize(-rayDir), hit.N));
                float lightPickProb = F(c, FLX_COL_LAST_PICK_PROB, gid);
                float lastPdfW = F(c, FLX_COL_LAST_PDF_W, gid);
                misWeight = lastPdfW / (lastPdfW + directPdfW * lightPickProb);
            }
            f3 newEi = R3(c, FLX_COL_EI, gid) + T * misWeight * V(p.areaLight.E);
            W3(c, FLX_COL_EI, gid, newEi);
            terminate = true;
        }

        /* :135-156 consume the previous vertex' light sample if its shadow ray was unblocked */
        bool blocked = U(c, FLX_COL_SHADOW_BLOCKED, gid) != 0;
        if (!blocked) {
            f3 emission = R3(c, FLX_COL_LAST_EMISSION, gid);
            f3 bsdf = R3(c, FLX_COL_LAST_BSDF, gid);
            float cosTh = F(c, FLX_COL_LAST_COS_TH, gid);
            float directPdfW = F(c, FLX_COL_LAST_PDF_DIRECT, gid);
            float bsdfPdfW = F(c, FLX_COL_LAST_PDF_IMPLICIT, gid);
            float lightPickProb = F(c, FLX_COL_LAST_PICK_PROB, gid);
            float weight = 1.0f;
            if (p.sampleImpl) weight = (directPdfW * lightPickProb) / (directPdfW * lightPickProb + bsdfPdfW);
            f3 lastT = R3(c, FLX_COL_LAST_T, gid);
            f3 contrib = bsdf * lastT * emission * weight * cosTh / (lightPickProb * directPdfW);
            f3 newEi = R3(c, FLX_COL_EI, gid) + contrib;
            W3(c, FLX_COL_EI, gid, newEi);
        }

        /* :163-177 splat + regenerate */
        if (terminate) {
            if (len > 0) {
                uint32_t pixIdx = U(c, FLX_COL_PIXEL_INDEX, gid);
                f3 Ei = R3(c, FLX_COL_EI, gid);
                float *px = &c.pixels[(size_t)pixIdx * 4];
                if (nthr == 1) { px[0] += Ei.x; px[1] += Ei.y; px[2] += Ei.z; px[3] += 1.0f; }
                else {
#pragma omp atomic
                    px[0] += Ei.x;
#pragma omp atomic
                    px[1] += Ei.y;
#pragma omp atomic
                    px[2] += Ei.z;
#pragma omp atomic
                    px[3] += 1.0f;
                }
            }
            app.push(FLX_Q_RAYGEN, gid);
            U(c, FLX_COL_SEED, gid) = seed;
            continue;
        }

        /* :180-184 */
        flx_material mat = c.materials[hit.matId];
        hit.N = tangentSpaceNormal(c, hit, mat);
        bool backface = dot(hit.N, rayDir) > 0.0f;
        if (backface) hit.N = hit.N * -1.0f;
        f3 orig = hit.P - 1e-3f * rayDir;

        /* :186-209 denoiser features (USE_OPTIX_DENOISER) */
        if (c.denoiser) {
            uint32_t pixIdx = U(c, FLX_COL_PIXEL_INDEX, gid);
            if (len == 1) addFloat4(&c.aovNormal[(size_t)pixIdx * 4], cameraSpaceNormal(p, hit.N), 1.0f, nthr > 1);
            bool isDiffuse = !FLX_BXDF_IS_SINGULAR(mat.type);
            if (isDiffuse && !U(c, FLX_COL_FIRST_DIFFUSE, gid)) {
                U(c, FLX_COL_FIRST_DIFFUSE, gid) = 1;
                f3 albedo = matGetFloat3(c, V(mat.Kd), hit.uv, mat.map_Kd);                                /* not gamma-corrected */
                addFloat4(&c.aovAlbedo[(size_t)pixIdx * 4], albedo, 1.0f, nthr > 1);
            }
        }

        /* :212-213 */
        writeHit(c, gid, hit);
        U(c, FLX_COL_BACKFACE, gid) = backface ? 1u : 0u;

        /* :217-302 next event estimation */
        if (p.sampleExpl && !FLX_BXDF_IS_SINGULAR(mat.type)) {
            uint32_t den = p.useEnvMap + p.useAreaLight; if (den < 1u) den = 1u;
            float envMapProb = (float)p.useEnvMap / (float)den;
            bool useEnvMap = rand01(&seed) < envMapProb;
            bool useAreaLight = !useEnvMap && p.useAreaLight;
            if (useEnvMap && p.useEnvMap) {           /* :226-256 (compiled iff USE_ENV_MAP) */
                float lightPickProb = envMapProb;
                f3 L; float directPdfW = 0.0f;
                sampleEnvMapAlias(c, rand01(&seed), &L, &directPdfW);
                float lenL = 2.0f * p.worldRadius;
                L = normalize(L);
                float cosTh = fmaxf_(0.0f, dot(L, hit.N));
                f3 envMapLi = evalEnvMapDir(c, L) * p.envMapStrength;
                W3(c, FLX_COL_SHADOW_ORIG, gid, orig);
                W3(c, FLX_COL_SHADOW_DIR, gid, L);
                F(c, FLX_COL_SHADOW_LEN, gid) = lenL;
                F(c, FLX_COL_LAST_PDF_DIRECT, gid) = directPdfW;
                F(c, FLX_COL_LAST_COS_TH, gid) = cosTh;
                F(c, FLX_COL_LAST_PICK_PROB, gid) = lightPickProb;
                W3(c, FLX_COL_LAST_EMISSION, gid, envMapLi);
                app.push(FLX_Q_SHADOW, gid);
            }
            if (useAreaLight) {                        /* :261-300 */
                float lightPickProb = 1.0f - envMapProb;
                float directPdfA; f3 posL;
                sampleAreaLight(p.areaLight, &directPdfA, &posL, &seed);
                f3 L = posL - orig;
                float lenL = length(L) * 0.995f;
                L = normalize(L);
                float cosLight = fmaxf_(dot(V(p.areaLight.N), -L), 0.0f);
                if (cosLight > 0.0f) {
                    float directPdfW = pdfAtoW(directPdfA, lenL, cosLight);
                    float cosTh = fmaxf_(0.0f, dot(L, hit.N));
                    W3(c, FLX_COL_SHADOW_ORIG, gid, orig);
                    W3(c, FLX_COL_SHADOW_DIR, gid, L);
                    F(c, FLX_COL_SHADOW_LEN, gid) = lenL;
                    F(c, FLX_COL_LAST_PDF_DIRECT, gid) = directPdfW;
                    F(c, FLX_COL_LAST_COS_TH, gid) = cosTh;
                    F(c, FLX_COL_LAST_PICK_PROB, gid) = lightPickProb;
                    W3(c, FLX_COL_LAST_EMISSION, gid, V(p.areaLight.E));
                    app.push(FLX_Q_SHADOW, gid);
                } else {
                    U(c, FLX_COL_SHADOW_BLOCKED, gid) = 1;
                }
            }
        }
        U(c, FLX_COL_SEED, gid) = seed;
        addToMaterialQueue(c, app, gid, mat);
    }
    }
    mergeAppenders(c, apps);
}

/* reference: wf_mat_diffuse.cl:7-67 (and its glossy/ggx_refl/ggx_refr/delta/all twins) */
void k_material_queue(Ctx &c, int q, uint32_t qlen)
{
    const int nthr = c.threads;
    std::vector<Appender> apps(nthr);
    const uint32_t chunk = (qlen + nthr - 1) / nthr;
#pragma omp parallel for schedule(static, 1) num_threads(nthr)
    for (int tchunk = 0; tchunk < nthr; tchunk++) {
    Appender &app = apps[tchunk];
    const uint32_t g0 = (uint32_t)tchunk * chunk, g1 = std::min(qlen, g0 + chunk);
    for (uint32_t gd = g0; gd < g1; gd++) {
        uint32_t gid = c.queues[q][gd];
        uint32_t seed = U(c, FLX_COL_SEED, gid);
        Hit hit = readHit(c, gid);
        const flx_material &mat = c.materials[hit.matId];
        bool backface = U(c, FLX_COL_BACKFACE, gid) != 0;
        f3 dirIn = R3(c, FLX_COL_DIR, gid);
        f3 L = R3(c, FLX_COL_SHADOW_DIR, gid);
        f3 bsdfNEE = bxdfEval(c, hit, mat, backface, dirIn, L);
        float bsdfPdfW = fmaxf_(0.0f, bxdfPdf(c, hit, mat, backface, dirIn, L));
        W3(c, FLX_COL_LAST_BSDF, gid, bsdfNEE);
        F(c, FLX_COL_LAST_PDF_IMPLICIT, gid) = bsdfPdfW;
        float pdfW = 0.0f;
        f3 newDir = mk3(0.0f);
        f3 bsdf = bxdfSample(c, hit, mat, backface, dirIn, &newDir, &pdfW, &seed);
        float costh = dot(hit.N, normalize(newDir));
        f3 oldT = R3(c, FLX_COL_T, gid);
        f3 newT;
        if (pdfW == 0.0f || is_zero(bsdf)) newT = mk3(0.0f);
        else newT = oldT * bsdf * costh / pdfW;
        f3 orig = hit.P + 1e-4f * newDir;
        W3(c, FLX_COL_LAST_T, gid, oldT);
        W3(c, FLX_COL_T, gid, newT);
        W3(c, FLX_COL_ORIG, gid, orig);
        W3(c, FLX_COL_DIR, gid, newDir);
        F(c, FLX_COL_LAST_PDF_W, gid) = pdfW;
        U(c, FLX_COL_SEED, gid) = seed;
        U(c, FLX_COL_LAST_SPECULAR, gid) = FLX_BXDF_IS_SINGULAR(mat.type) ? 1u : 0u;
        app.push(FLX_Q_EXTENSION, gid);
    }
    }
    mergeAppenders(c, apps);
}

/* reference: clcontext.cpp:796-813 */
void k_materials(Ctx &c)
{
    if (c.params.wfSeparateQueues) {
        k_material_queue(c, FLX_Q_DIFFUSE, c.counters.diffuseQueue);
        k_material_queue(c, FLX_Q_GLOSSY, c.counters.glossyQueue);
        k_material_queue(c, FLX_Q_GGX_REFL, c.counters.ggxReflQueue);
        k_material_queue(c, FLX_Q_GGX_REFR, c.counters.ggxRefrQueue);
        k_material_queue(c, FLX_Q_DELTA, c.counters.deltaQueue);
    } else {
        k_material_queue(c, FLX_Q_DIFFUSE, c.counters.diffuseQueue);
    }
}


/* ------------------------------------------------------------------------ */
/* microkernel integrator (mk_*.cl) -- SURVEY 8(f) N3                         */
/* one path per pixel, `phase` state machine (geom.h:183-192)                */
/* ------------------------------------------------------------------------ */
enum { MK_RT_NEXT_VERTEX = 0, MK_SAMPLE_BSDF = 1, MK_SPLAT_SAMPLE = 4, MK_GENERATE_CAMERA_RAY = 5 };

uint32_t mkLimit(Ctx &c) { return std::min(c.params.width * c.params.height, c.numTasks); }

/* reference: mk_reset.cl:3-43 */
void k_mk_reset(Ctx &c)
{
    uint32_t limit = mkLimit(c);
    for (uint32_t gid = 0; gid < limit; gid++) {
        float *px = &c.pixels[(size_t)gid * 4]; px[0] = px[1] = px[2] = px[3] = 0.0f;
        float *nr = &c.aovNormal[(size_t)gid * 4]; nr[0] = nr[1] = nr[2] = nr[3] = 0.0f;                 /* mk_reset.cl:24-25 */
        float *al = &c.aovAlbedo[(size_t)gid * 4]; al[0] = al[1] = al[2] = 0.1f; al[3] = 0.0f;
        I(c, FLX_COL_PHASE, gid) = MK_GENERATE_CAMERA_RAY;
        W3(c, FLX_COL_EI, gid, mk3(0.0f)); W3(c, FLX_COL_T, gid, mk3(1.0f));
        U(c, FLX_COL_PATH_LEN, gid) = 0; U(c, FLX_COL_LAST_SPECULAR, gid) = 1; F(c, FLX_COL_LAST_PDF_W, gid) = 1.0f;
        U(c, FLX_COL_FIRST_DIFFUSE, gid) = 0; U(c, FLX_COL_SEED, gid) = gid;
    }
}

/* reference: mk_raygen.cl:4-63 */
void k_mk_raygen(Ctx &c)
{
    const flx_render_params &p = c.params;
    uint32_t limit = mkLimit(c);
    for (uint32_t gid = 0; gid < limit; gid++) {
        if (I(c, FLX_COL_PHASE, gid) != MK_GENERATE_CAMERA_RAY) continue;
        uint32_t seed = U(c, FLX_COL_SEED, gid);
        float x = (float)(gid % p.width), y = (float)(gid / p.width);
        x += rand01(&seed); y += rand01(&seed);
        float NDCx = x / (float)p.width, NDCy = y / (float)p.height;
        float SCRx = 2.0f * NDCx - 1.0f, SCRy = 2.0f * NDCy - 1.0f;
        SCRx *= (float)p.width / (float)p.height;
        float scale = tanf_(0.5f * p.camera.fov * FLX_PI / 180.0f);
        SCRx *= scale; SCRy *= scale;
        f3 rayOrig = V(p.camera.pos);
        f3 rayTarget = rayOrig + V(p.camera.right) * SCRx + V(p.camera.up) * SCRy + V(p.camera.dir);
        f3 rayDirection = normalize(rayTarget - rayOrig);
        f3 fp = V(p.camera.pos) + rayDirection * p.camera.focalDist;
        f2 rnd = uniformSampleDisk(&seed);
        rayOrig = rayOrig + p.worldRadius * p.camera.apertureSize * (V(p.camera.right) * rnd.x + V(p.camera.up) * rnd.y);
        rayDirection = normalize(fp - rayOrig);
        W3(c, FLX_COL_ORIG, gid, rayOrig); W3(c, FLX_COL_DIR, gid, rayDirection);
        U(c, FLX_COL_SEED, gid) = seed;
        I(c, FLX_COL_PHASE, gid) = MK_RT_NEXT_VERTEX;
    }
}

/* reference: mk_next_vertex.cl:7-123 */
void k_mk_next_vertex(Ctx &c)
{
    const flx_render_params &p = c.params;
    uint32_t limit = mkLimit(c);
    uint32_t prim = 0, ext = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : prim, ext) num_threads(c.threads)
    for (int64_t g = 0; g < (int64_t)limit; g++) {
        uint32_t gid = (uint32_t)g;
        if (I(c, FLX_COL_PHASE, gid) != MK_RT_NEXT_VERTEX) continue;
        f3 rayOrig = R3(c, FLX_COL_ORIG, gid), rayDir = R3(c, FLX_COL_DIR, gid);
        Hit hit = emptyHit(FLX_FLT_MAX);
        uint64_t a = 0, b = 0;
        bvh_intersect(c, rayOrig, rayDir, &hit, &a, &b);
        if (p.sampleImpl && p.useAreaLight) intersectLight(&hit, rayOrig, rayDir, p);
        writeHit(c, gid, hit);
        uint32_t len = U(c, FLX_COL_PATH_LEN, gid);
        if (len == 0) prim++; else ext++;
        len += 1;
        U(c, FLX_COL_PATH_LEN, gid) = len;
        if (c.denoiser && len == 1) addFloat4(&c.aovNormal[(size_t)gid * 4], cameraSpaceNormal(p, hit.N), 1.0f, false);   /* mk_next_vertex.cl:59-69 */
        if (hit.i < 0) {
            f3 bg = mk3(0.0f);
            if (p.useEnvMap && (len == 1 || p.sampleImpl)) bg = evalEnvMapDir(c, rayDir) * p.envMapStrength;
            float weight = 1.0f;
            bool lastSpecular = U(c, FLX_COL_LAST_SPECULAR, gid) != 0;
            if (p.sampleImpl && p.sampleExpl && p.useEnvMap && len > 1 && !lastSpecular) {
                const float lightPickProb = 1.0f;
                float directPdfW = envMapPdf(c, rayDir);
                float actualPdfW = F(c, FLX_COL_LAST_PDF_W, gid);
                weight = (actualPdfW * lightPickProb) / (actualPdfW * lightPickProb + directPdfW);
            }
            f3 T = R3(c, FLX_COL_T, gid);
            W3(c, FLX_COL_EI, gid, R3(c, FLX_COL_EI, gid) + weight * T * bg);
            I(c, FLX_COL_PHASE, gid) = MK_SPLAT_SAMPLE;
        } else if (hit.areaLightHit) {
            float misWeight = 1.0f;
            bool lastSpecular = U(c, FLX_COL_LAST_SPECULAR, gid) != 0;
            if (p.sampleExpl && len > 1 && !lastSpecular) {
                float directPdfA = 1.0f / (4.0f * p.areaLight.size.x * p.areaLight.size.y);
                float directPdfW = pdfAtoW(directPdfA, length(hit.P - rayOrig), dot(normalize(-rayDir), hit.N));
                const float lightPickProb = 1.0f;
                float lastPdfW = F(c, FLX_COL_LAST_PDF_W, gid);
                misWeight = lastPdfW / (lastPdfW + directPdfW * lightPickProb);
            }
            f3 T = R3(c, FLX_COL_T, gid);
            W3(c, FLX_COL_EI, gid, R3(c, FLX_COL_EI, gid) + T * misWeight * V(p.areaLight.E));
            I(c, FLX_COL_PHASE, gid) = MK_SPLAT_SAMPLE;
        } else {
            I(c, FLX_COL_PHASE, gid) = MK_SAMPLE_BSDF;
        }
    }
    c.mkStats[0] += prim; c.mkStats[1] += ext;
}

/* reference: mk_sample_bsdf.cl:8-197 */
void k_mk_sample_bsdf(Ctx &c)
{
    const flx_render_params &p = c.params;
    uint32_t limit = mkLimit(c);
    uint32_t shadow = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : shadow) num_threads(c.threads)
    for (int64_t g = 0; g < (int64_t)limit; g++) {
        uint32_t gid = (uint32_t)g;
        if (I(c, FLX_COL_PHASE, gid) != MK_SAMPLE_BSDF) continue;
        uint32_t seed = U(c, FLX_COL_SEED, gid);
        f3 rayDir = R3(c, FLX_COL_DIR, gid);
        Hit hit = readHit(c, gid);
        const flx_material &mat = c.materials[hit.matId];
        hit.N = tangentSpaceNormal(c, hit, mat);
        bool backface = dot(hit.N, rayDir) > 0.0f;
        if (backface) hit.N = hit.N * -1.0f;
        f3 orig = hit.P - 1e-3f * rayDir;
        if (c.denoiser && !FLX_BXDF_IS_SINGULAR(mat.type) && !U(c, FLX_COL_FIRST_DIFFUSE, gid)) {             /* mk_sample_bsdf.cl:56-66 */
            U(c, FLX_COL_FIRST_DIFFUSE, gid) = 1;
            addFloat4(&c.aovAlbedo[(size_t)gid * 4], matGetFloat3(c, V(mat.Kd), hit.uv, mat.map_Kd), 1.0f, false);
        }
        uint64_t a = 0, b = 0;
        if (p.sampleExpl && !FLX_BXDF_IS_SINGULAR(mat.type)) {
            const float lightPickProb = 1.0f;
            if (p.useEnvMap) {
                f3 L; float directPdfW = 0.0f;
                sampleEnvMapAlias(c, rand01(&seed), &L, &directPdfW);
                float lenL = 2.0f * p.worldRadius;
                L = normalize(L);
                Hit hitL = emptyHit(lenL);
                if (p.useAreaLight) intersectLight(&hitL, orig, L, p);
                bool occluded = (hitL.i > -1) || bvh_occluded(c, orig, L, lenL, &a, &b);
                shadow++;
                if (!occluded && directPdfW != 0.0f) {
                    f3 brdf = bxdfEval(c, hit, mat, backface, rayDir, L);
                    float cosTh = fmaxf_(0.0f, dot(L, hit.N));
                    float bsdfPdfW = fmaxf_(0.0f, bxdfPdf(c, hit, mat, backface, rayDir, L));
                    float weight = 1.0f;
                    if (p.sampleImpl) weight = (directPdfW * lightPickProb) / (directPdfW * lightPickProb + bsdfPdfW);
                    f3 T = R3(c, FLX_COL_T, gid);
                    f3 envMapLi = evalEnvMapDir(c, L) * p.envMapStrength;
                    f3 contrib = brdf * T * envMapLi * weight * cosTh / (lightPickProb * directPdfW);
                    W3(c, FLX_COL_EI, gid, R3(c, FLX_COL_EI, gid) + contrib);
                }
            }
            if (p.useAreaLight) {
                float directPdfA; f3 posL;
                sampleAreaLight(p.areaLight, &directPdfA, &posL, &seed);
                f3 L = posL - orig;
                float lenL = length(L);
                L = normalize(L);
                bool occluded = bvh_occluded(c, orig, L, lenL, &a, &b);
                shadow++;
                float cosLight = fmaxf_(dot(V(p.areaLight.N), -L), 0.0f);
                if (!occluded && cosLight > 0.0f) {
                    f3 brdf = bxdfEval(c, hit, mat, backface, rayDir, L);
                    float cosTh = fmaxf_(0.0f, dot(L, hit.N));
                    float directPdfW = pdfAtoW(directPdfA, lenL, cosLight);
                    float bsdfPdfW = fmaxf_(0.0f, bxdfPdf(c, hit, mat, backface, rayDir, L));
                    float weight = 1.0f;
                    if (p.sampleImpl) weight = (directPdfW * lightPickProb) / (directPdfW * lightPickProb + bsdfPdfW);
                    f3 T = R3(c, FLX_COL_T, gid);
                    f3 contrib = brdf * T * V(p.areaLight.E) * weight * cosTh / (lightPickProb * directPdfW);
                    W3(c, FLX_COL_EI, gid, R3(c, FLX_COL_EI, gid) + contrib);
                }
            }
        }
        float contProb = 1.0f;
        uint32_t len = U(c, FLX_COL_PATH_LEN, gid);
        bool terminate = (len - 1 >= p.maxBounces);
        if (terminate && p.useRoulette) {
            contProb = clampf(luminance(R3(c, FLX_COL_T, gid)), 0.01f, 0.5f);
            terminate = (rand01(&seed) > contProb);
        }
        float pdfW = 0.0f; f3 newDir = mk3(0.0f);   /* uninitialised in the reference; see sampleGlossy */
        f3 bsdf = bxdfSample(c, hit, mat, backface, rayDir, &newDir, &pdfW, &seed);
        float costh = dot(hit.N, normalize(newDir));
        pdfW *= contProb;
        if (pdfW == 0.0f || is_zero(bsdf)) terminate = true;
        f3 newT = R3(c, FLX_COL_T, gid) * bsdf * costh / pdfW;
        orig = hit.P + 1e-4f * newDir;
        W3(c, FLX_COL_T, gid, newT); W3(c, FLX_COL_ORIG, gid, orig); W3(c, FLX_COL_DIR, gid, newDir);
        F(c, FLX_COL_LAST_PDF_W, gid) = pdfW;
        U(c, FLX_COL_SEED, gid) = seed;
        U(c, FLX_COL_LAST_SPECULAR, gid) = FLX_BXDF_IS_SINGULAR(mat.type) ? 1u : 0u;
        I(c, FLX_COL_PHASE, gid) = terminate ? MK_SPLAT_SAMPLE : MK_RT_NEXT_VERTEX;
    }
    c.mkStats[2] += shadow;
}

/* reference: mk_splat.cl:4-41 */
void k_mk_splat(Ctx &c)
{
    uint32_t limit = mkLimit(c);
    for (uint32_t gid = 0; gid < limit; gid++) {
        if (I(c, FLX_COL_PHASE, gid) != MK_SPLAT_SAMPLE) continue;
        f3 Ei = R3(c, FLX_COL_EI, gid);
        float *px = &c.pixels[(size_t)gid * 4];
        float col[4] = {Ei.x, Ei.y, Ei.z, 1.0f};
        if (px[3] > 0.0f) for (int k = 0; k < 4; k++) col[k] += px[k];
        for (int k = 0; k < 4; k++) px[k] = col[k];
        c.mkStats[3]++;
        W3(c, FLX_COL_EI, gid, mk3(0.0f)); W3(c, FLX_COL_T, gid, mk3(1.0f));
        U(c, FLX_COL_PATH_LEN, gid) = 0; U(c, FLX_COL_FIRST_DIFFUSE, gid) = 0;
        I(c, FLX_COL_PHASE, gid) = MK_GENERATE_CAMERA_RAY;
    }
}

/* reference: mk_splat_preview.cl:3-25 */
void k_mk_splat_preview(Ctx &c)
{
    uint32_t limit = mkLimit(c);
    for (uint32_t gid = 0; gid < limit; gid++) {
        f3 Ei = R3(c, FLX_COL_EI, gid);
        float *px = &c.pixels[(size_t)gid * 4];
        px[0] = Ei.x; px[1] = Ei.y; px[2] = Ei.z; px[3] = 0.0f;
        W3(c, FLX_COL_EI, gid, mk3(0.0f)); W3(c, FLX_COL_T, gid, mk3(1.0f));
        U(c, FLX_COL_PATH_LEN, gid) = 0;
        I(c, FLX_COL_PHASE, gid) = MK_GENERATE_CAMERA_RAY;
    }
}

/* reference: tonemap.cl:3-26 */
f3 uc2TonemapFunc(f3 x)
{
    const float A = 0.22f, B = 0.30f, C = 0.10f, D = 0.20f, E = 0.01f, Fq = 0.30f;
    return ((x * (A * x + mk3(C * B)) + mk3(D * E)) / (x * (A * x + mk3(B)) + mk3(D * Fq))) - mk3(E / Fq);
}
f3 uncharted2Tonemap(f3 x) { return uc2TonemapFunc(2.0f * x) / uc2TonemapFunc(mk3(11.2f)); }
f3 reinhardTonemap(f3 col) { return col / (mk3(1.0f) + col); }

/* reference: mk_postprocess.cl:7-55 */
void k_postprocess(Ctx &c)
{
    const flx_render_params &p = c.params;
    uint32_t limit = p.width * p.height;
    for (uint32_t gid = 0; gid < limit; gid++) {
        const float *in = &c.pixels[(size_t)gid * 4];
        f3 col = mk3(in[0], in[1], in[2]); float w = in[3];
        if (w > 0.0f) { col = col / w; w = w / w; }
        col = col * p.exposure;
        if (p.tmOperator == 1) col = reinhardTonemap(col);
        if (p.tmOperator == 2) col = uncharted2Tonemap(col);
        col = pow3(col, 1.0f / 2.2f);
        float *out = &c.preview[(size_t)gid * 4];
        out[0] = col.x; out[1] = col.y; out[2] = col.z; out[3] = w;
        if (c.denoiser) {                                                   /* mk_postprocess.cl:49-54 */
            for (int which = 0; which < 2; which++) {
                const float *a = which ? &c.aovAlbedo[(size_t)gid * 4] : &c.aovNormal[(size_t)gid * 4];
                float *o = which ? &c.aovAlbedoOut[(size_t)gid * 4] : &c.aovNormalOut[(size_t)gid * 4];
                const bool nrm = a[3] > 1.0f;
                for (int k = 0; k < 4; k++) o[k] = nrm ? a[k] / a[3] : a[k];
            }
        }
    }
}

} /* namespace */

/* ------------------------------------------------------------------------ */
/* C entry points (mirror include/fluctus_hip.h one-to-one, prefix orc_)      */
/* ------------------------------------------------------------------------ */
extern "C" {

typedef struct orc_ctx orc_ctx;
#define CTX(p) (*reinterpret_cast<Ctx *>(p))

int orc_create(uint32_t num_tasks, orc_ctx **out)
{
    Ctx *c = new Ctx();
    c->numTasks = num_tasks;
    c->state.assign((size_t)FLX_NUM_COLS * num_tasks, 0.0f);
    for (auto &q : c->queues) q.assign(num_tasks, 0u);
    c->envRGBA.assign(4, 0.0f);
    c->probTable.assign(1, 1.0f); c->pdfTable.assign(1, 1.0f); c->aliasTable.assign(1, 0);
    *out = reinterpret_cast<orc_ctx *>(c);
    return 0;
}
int orc_destroy(orc_ctx *p) { delete &CTX(p); return 0; }
int orc_set_threads(orc_ctx *p, int n) { CTX(p).threads = n < 1 ? 1 : n; return 0; }

int orc_upload_scene(orc_ctx *p, const void *tris, size_t ntris, const uint32_t *indices, size_t nidx,
                     const void *nodes, size_t nnodes, const void *materials, size_t nmat,
                     const void *texdesc, size_t ntex, const uint8_t *texdata, size_t texbytes)
{
    Ctx &c = CTX(p);
    c.tris.assign((const flx_triangle *)tris, (const flx_triangle *)tris + ntris);
    c.indices.assign(indices, indices + nidx);
    c.nodes.assign((const flx_node *)nodes, (const flx_node *)nodes + nnodes);
    c.materials.assign((const flx_material *)materials, (const flx_material *)materials + nmat);
    c.texdesc.assign((const flx_texdesc *)texdesc, (const flx_texdesc *)texdesc + ntex);
    c.texdata.assign(texdata, texdata + texbytes);
    return 0;
}

/* reference: clcontext.cpp:467-511 (createEnvMap: RGB -> RGBA float image + 3 tables) */
int orc_upload_envmap(orc_ctx *p, const float *rgb, int w, int h, const float *prob, const int *alias, const float *pdf)
{
    Ctx &c = CTX(p);
    c.envW = w; c.envH = h;
    c.envRGBA.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        c.envRGBA[i * 4 + 0] = rgb[i * 3 + 0]; c.envRGBA[i * 4 + 1] = rgb[i * 3 + 1];
        c.envRGBA[i * 4 + 2] = rgb[i * 3 + 2]; c.envRGBA[i * 4 + 3] = 1.0f;
    }
    c.probTable.assign(prob, prob + (size_t)w * h);
    c.aliasTable.assign(alias, alias + (size_t)w * h);
    c.pdfTable.assign(pdf, pdf + (size_t)w * h);
    return 0;
}

int orc_set_params(orc_ctx *p, const void *params240)
{
    Ctx &c = CTX(p);
    memcpy(&c.params, params240, sizeof(flx_render_params));
    size_t npix = (size_t)c.params.width * c.params.height;
    if (c.pixels.size() != npix * 4) {
        c.pixels.assign(npix * 4, 0.0f); c.preview.assign(npix * 4, 0.0f);
        c.aovAlbedo.assign(npix * 4, 0.0f); c.aovNormal.assign(npix * 4, 0.0f); c.aovAlbedoOut.assign(npix * 4, 0.0f); c.aovNormalOut.assign(npix * 4, 0.0f);
    }
    return 0;
}
int orc_set_partition(orc_ctx *p, uint32_t rank, uint32_t nranks) { CTX(p).rank = rank; CTX(p).nranks = nranks; return 0; }

int orc_wf_reset(orc_ctx *p) { k_reset(CTX(p)); return 0; }
int orc_wf_raygen(orc_ctx *p) { k_raygen(CTX(p)); return 0; }
int orc_wf_extend(orc_ctx *p) { k_extend(CTX(p)); return 0; }
int orc_wf_shadow(orc_ctx *p) { k_shadow(CTX(p)); return 0; }
int orc_wf_logic(orc_ctx *p, int first) { k_logic(CTX(p), first); return 0; }
int orc_wf_materials(orc_ctx *p) { k_materials(CTX(p)); return 0; }
int orc_postprocess(orc_ctx *p) { k_postprocess(CTX(p)); return 0; }
int orc_mk_reset(orc_ctx *p) { k_mk_reset(CTX(p)); return 0; }
int orc_mk_raygen(orc_ctx *p) { k_mk_raygen(CTX(p)); return 0; }
int orc_mk_next_vertex(orc_ctx *p) { k_mk_next_vertex(CTX(p)); return 0; }
int orc_mk_sample_bsdf(orc_ctx *p) { k_mk_sample_bsdf(CTX(p)); return 0; }
int orc_mk_splat(orc_ctx *p) { k_mk_splat(CTX(p)); return 0; }
int orc_mk_splat_preview(orc_ctx *p) { k_mk_splat_preview(CTX(p)); return 0; }
int orc_mk_stats(orc_ctx *p, uint32_t *out4, int reset) { Ctx &c = CTX(p); memcpy(out4, c.mkStats, 16); if (reset) memset(c.mkStats, 0, 16); return 0; }
int orc_clear_queues(orc_ctx *p) { memset(&CTX(p).counters, 0, sizeof(flx_queue_counters)); return 0; }
int orc_get_counters(orc_ctx *p, void *out32) { memcpy(out32, &CTX(p).counters, 32); return 0; }
int orc_set_counters(orc_ctx *p, const void *in32) { memcpy(&CTX(p).counters, in32, 32); return 0; }
/* reference: clcontext.cpp:891-901 */
int orc_pixel_index_update(orc_ctx *p, uint32_t npix, uint32_t nnew)
{
    Ctx &c = CTX(p);
    c.hostPixelIdx = (c.hostPixelIdx + nnew) % npix;
    c.currPixelIdx = c.hostPixelIdx;
    return 0;
}
int orc_pixel_index_reset(orc_ctx *p) { CTX(p).hostPixelIdx = 0; CTX(p).currPixelIdx = 0; return 0; }

int orc_read_pixels(orc_ctx *p, int which, float *out)
{
    Ctx &c = CTX(p);
    const std::vector<float> *srcs[6] = {&c.pixels, &c.preview, &c.aovAlbedoOut, &c.aovNormalOut, &c.aovAlbedo, &c.aovNormal};
    if (which < 0 || which > 5) return 1;
    memcpy(out, srcs[which]->data(), srcs[which]->size() * sizeof(float));
    return 0;
}
int orc_set_option(orc_ctx *p, const char *name, int value)
{
    if (name && strcmp(name, "denoiser") == 0) { CTX(p).denoiser = value != 0; return 0; }
    return 1;
}
int orc_state_export(orc_ctx *p, float *out) { Ctx &c = CTX(p); memcpy(out, c.state.data(), c.state.size() * 4); return 0; }
int orc_state_import(orc_ctx *p, const float *in) { Ctx &c = CTX(p); memcpy(c.state.data(), in, c.state.size() * 4); return 0; }
int orc_queue_read(orc_ctx *p, int q, uint32_t *out) { Ctx &c = CTX(p); memcpy(out, c.queues[q].data(), (size_t)c.numTasks * 4); return 0; }
int orc_queue_write(orc_ctx *p, int q, const uint32_t *in, uint32_t n) { Ctx &c = CTX(p); memcpy(c.queues[q].data(), in, (size_t)n * 4); return 0; }
/* ext rays, ext inner visits, ext triangle tests, ext hits, shadow inner, shadow tri, shadow rays */
int orc_get_stats(orc_ctx *p, uint64_t *out7)
{
    Ctx &c = CTX(p);
    for (int i = 0; i < 6; i++) out7[i] = c.stat[i];
    out7[6] = c.statShadowRays;
    return 0;
}
int orc_reset_stats(orc_ctx *p) { Ctx &c = CTX(p); memset(c.stat, 0, sizeof(c.stat)); c.statShadowRays = 0; return 0; }

/* scalar probes of the arithmetic contract, for tests/test_math.py */
float orc_math(int fn, float a, float b)
{
    switch (fn) {
    case 0: return sinf_(a); case 1: return cosf_(a); case 2: return tanf_(a);
    case 3: return atan2f_(a, b); case 4: return acosf_(a); case 5: return powf_(a, b);
    case 6: return logf_(a); case 7: return expf_(a); case 8: return asinf_(a); case 9: return atanf_(a);
    case 10: return fminf_(a, b); case 11: return fmaxf_(a, b); case 12: return a / b; case 13: return sqrtf(a);
    case 14: return a * b; case 15: return a + b;
    }
    return 0.0f;
}
/* the same over arrays, results as bit patterns (signed zeros and NaN payloads included): the oracle's half of
 * tests/test_gpu_parity.py::test_arithmetic_contract_device_vs_oracle */
void orc_math_array(int fn, const float *a, const float *b, uint32_t n, uint32_t *out)
{
    for (uint32_t i = 0; i < n; i++) out[i] = f2u(orc_math(fn, a[i], b[i]));
}
uint32_t orc_hash(uint32_t s) { return hash_u32(s); }
/* What `logic`'s next-event estimation derives from an importance-sampled texel, for EVERY texel of the uploaded environment map, with the calls and in the
 * order of the kernel above (sampleEnvMapAlias' second half, normalize, evalEnvMapDir: src/wf_logic.cl:236-249, src/env_map.cl:65-92, :39-43): 8 floats per
 * texel {L.xyz, pdfW, Li.xyz (before envMapStrength), 0}.  The device keeps exactly this as a table built at upload (flx_device.h: Scene::neeRec);
 * tests/test_gpu_parity.py::test_env_sample_table_bit_identical compares the two bit for bit. */
void orc_env_sample_table(orc_ctx *p, float *out)
{
    Ctx &c = CTX(p);
    const int width = c.envW, height = c.envH;
    for (int uvInd = 0; uvInd < width * height; uvInd++) {
        float pdf_uv = c.pdfTable[uvInd];
        int uInd = uvInd % width, vInd = uvInd / width;
        float u = ((float)uInd + 0.5f) / (float)width;
        float v = ((float)vInd + 0.5f) / (float)height;
        f3 L = UVToDirection(u, v);
        float sinTh = sinf_(FLX_PI * v);
        float directPdfUV = pdf_uv * 1.0f;
        float pdfW = 0.0f;
        if (sinTh != 0.0f) pdfW = directPdfUV / (2.0f * FLX_PI * FLX_PI * sinTh);
        L = normalize(L);
        f3 Li = evalEnvMapDir(c, L);
        float *o = out + (size_t)uvInd * 8;
        o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = pdfW; o[4] = Li.x; o[5] = Li.y; o[6] = Li.z; o[7] = 0.0f;
    }
}

} /* extern "C" */
