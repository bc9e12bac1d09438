// flx_shading.h -- device helpers shared by the logic and material kernels: texture fetch,
// tangent-space normals, environment-map evaluation / importance sampling, light sampling.
// Arithmetic follows include/flx_math.h (bit-identical to the oracle).
#pragma once
#include "flx_device.h"

namespace flxd {

// nearest texel, wrap by double modulo, RGBA8 (reference: src/utils.cl:114-133)
__device__ __forceinline__ f3 read_texture(const Scene &sc, f2 uv, int idx)
{
    const flx_texdesc tex = sc.texdesc[idx];
    float ux = uv.x * (float)tex.width, uy = uv.y * (float)tex.height;
    int w = (int)tex.width, h = (int)tex.height;
    float fx = floorf(ux), fy = floorf(uy);
    int tx = (((int)fx) % w + w) % w;
    int ty = (((int)fy) % h + h) % h;
    int cx = (int)((float)tx + ux - fx);
    int cy = (int)((float)ty + uy - fy);
    cx = cx < 0 ? 0 : (cx > w - 1 ? w - 1 : cx);
    cy = cy < 0 ? 0 : (cy > h - 1 ? h - 1 : cy);
    const uint32_t texel = *reinterpret_cast<const uint32_t *>(sc.texdata + tex.offset + ((size_t)cx + (size_t)cy * tex.width) * 4);
    return mk3((float)(texel & 255u), (float)((texel >> 8) & 255u), (float)((texel >> 16) & 255u)) / 255.0f;
}

// reference: src/utils.cl:136-146
#ifdef FLX_LAB_NOTEX                 // lab build only (RESULTS INVALID): no texture fetches
__device__ __forceinline__ f3 mat_float3(const Scene &sc, f3 fallback, f2 uv, int idx) { return fallback; }
#else
__device__ __forceinline__ f3 mat_float3(const Scene &sc, f3 fallback, f2 uv, int idx) { return idx != -1 ? read_texture(sc, uv, idx) : fallback; }
#endif
__device__ __forceinline__ f3 mat_albedo(const Scene &sc, f3 fallback, f2 uv, int idx) { return pow3(mat_float3(sc, fallback, uv, idx), 2.2f); }

// reference: src/utils.cl:149-182
__device__ __forceinline__ f3 tangent_space_normal(const Scene &sc, f3 N, f2 uv, int tri, int mapN)
{
    if (mapN == -1) return N;
    f3 texNormal = mat_float3(sc, mk3(0.5f, 0.5f, 1.0f), uv, mapN);
    texNormal = 2.0f * texNormal - mk3(1.0f, 1.0f, 1.0f);
    const flx_triangle &t = sc.tris[tri];
    f3 e1 = V(t.v1.p) - V(t.v0.p), e2 = V(t.v2.p) - V(t.v0.p);
    f3 t1 = V(t.v1.t) - V(t.v0.t), t2 = V(t.v2.t) - V(t.v0.t);
    float det = t1.x * t2.y - t1.y * t2.x;
    if (det == 0.0f) return N;
    float invDet = 1.0f / det;
    f3 T = normalize(invDet * (e1 * t2.y - e2 * t1.y));
    f3 B = normalize(invDet * (e2 * t1.x - e1 * t2.x));
    f3 r;
    r.x = T.x * texNormal.x + B.x * texNormal.y + N.x * texNormal.z;
    r.y = T.y * texNormal.x + B.y * texNormal.y + N.y * texNormal.z;
    r.z = T.z * texNormal.x + B.z * texNormal.y + N.z * texNormal.z;
    return normalize(r);
}

__device__ __forceinline__ float pdf_a_to_w(float pdf, float dist, float cosine) { return pdf * (dist * dist) / absf(cosine); }
__device__ __forceinline__ float luminance(f3 v) { return 0.212671f * v.x + 0.715160f * v.y + 0.072169f * v.z; }

// ---- environment map (reference: src/env_map.cl) ----------------------------------------

__device__ __forceinline__ f2 direction_to_uv(f3 dir)
{
    if (dir.x == 0.0f && dir.y == 0.0f && dir.z == 0.0f) return mk2(0.0f, 0.0f);
    float u = 1.0f + atan2f_(dir.x, -dir.z) / FLX_PI;
    float r = clampf(dir.y / length(dir), -1.0f, 1.0f);
    float v = acosf_(r) / FLX_PI;
    return mk2(u * 0.5f, v);
}

__device__ __forceinline__ f3 uv_to_direction(float u, float v)
{
    float phi = v * FLX_PI;
    float theta = (u * 2.0f - 1.0f) * FLX_PI;
    float sinPhi, cosPhi, sinTh, cosTh;
    sincosf_(phi, &sinPhi, &cosPhi);
    sincosf_(theta, &sinTh, &cosTh);
    return mk3(sinPhi * sinTh, cosPhi, -sinPhi * cosTh);
}

// The reference reads the map through an OpenCL sampler (normalised coords, clamp-to-edge, linear).
// Hardware samplers filter with low-precision fixed-point weights, so the filter is done by hand in
// fp32 on four texel loads, per the OpenCL 1.2 s8.2 formula (reference: src/env_map.cl:10,39-43).
__device__ __forceinline__ f3 eval_env_uv(const Scene &sc, f2 uv)
{
    int w = sc.envW, h = sc.envH;
    float fu = uv.x * (float)w - 0.5f, fv = uv.y * (float)h - 0.5f;
    float flu = floorf(fu), flv = floorf(fv);
    float a = fu - flu, b = fv - flv;
    int i0 = (int)flu, j0 = (int)flv, i1 = i0 + 1, j1 = j0 + 1;
    i0 = i0 < 0 ? 0 : (i0 > w - 1 ? w - 1 : i0); i1 = i1 < 0 ? 0 : (i1 > w - 1 ? w - 1 : i1);
    j0 = j0 < 0 ? 0 : (j0 > h - 1 ? h - 1 : j0); j1 = j1 < 0 ? 0 : (j1 > h - 1 ? h - 1 : j1);
    const float4 t00 = sc.envRGBA[(size_t)j0 * w + i0], t10 = sc.envRGBA[(size_t)j0 * w + i1];
    const float4 t01 = sc.envRGBA[(size_t)j1 * w + i0], t11 = sc.envRGBA[(size_t)j1 * w + i1];
    return (1.0f - a) * (1.0f - b) * ld3(t00) + a * (1.0f - b) * ld3(t10) + (1.0f - a) * b * ld3(t01) + a * b * ld3(t11);
}
__device__ __forceinline__ f3 eval_env_dir(const Scene &sc, f3 dir) { return eval_env_uv(sc, direction_to_uv(dir)); }

// alias-method sampling of the texel distribution (reference: src/env_map.cl:65-92)
__device__ __forceinline__ void sample_env_alias(const Scene &sc, float rnd, f3 *L, float *pdfW)
{
    int width = sc.envW, height = sc.envH;
    float r = rnd * (float)width * (float)height;
    int i = (int)floorf(r);
    if (i > width * height - 1) i = width * height - 1;
    const float2 rec = sc.aliasRec[i];                              // {prob[i], alias[i]} (flx_device.h)
    int uvInd = (r - (float)i < rec.x) ? i : __float_as_int(rec.y);
    float pdf_uv = sc.pdfTable[uvInd];
    int uInd = uvInd % width, vInd = uvInd / width;
    float u = ((float)uInd + 0.5f) / (float)width;
    float v = ((float)vInd + 0.5f) / (float)height;
    *L = uv_to_direction(u, v);
    float sinTh = sinf_(FLX_PI * v);
    float directPdfUV = pdf_uv * 1.0f;
    if (sinTh != 0.0f) *pdfW = directPdfUV / (2.0f * FLX_PI * FLX_PI * sinTh);
    else *pdfW = 0.0f;
}

// the alias step alone: which texel the random number picks (the first half of sample_env_alias)
__device__ __forceinline__ int sample_env_index(const Scene &sc, float rnd)
{
    int width = sc.envW, height = sc.envH;
    float r = rnd * (float)width * (float)height;
    int i = (int)floorf(r);
    if (i > width * height - 1) i = width * height - 1;
    const float2 rec = sc.aliasRec[i];
    return (r - (float)i < rec.x) ? i : __float_as_int(rec.y);
}
// what logic's next-event estimation derives from the sampled texel uvInd (src/wf_logic.cl:236-249 through src/env_map.cl:65-92 and :39-43): the
// second half of sample_env_alias, the normalisation of the direction and the radiance lookup along it -- the operations of the inline code, in its order
struct EnvSample { f3 L; float pdfW; f3 Li; };
__device__ __forceinline__ EnvSample env_sample_compute(const Scene &sc, int uvInd)
{
    int width = sc.envW, height = sc.envH;
    EnvSample e;
    float pdf_uv = sc.pdfTable[uvInd];
    int uInd = uvInd % width, vInd = uvInd / width;
    float u = ((float)uInd + 0.5f) / (float)width;
    float v = ((float)vInd + 0.5f) / (float)height;
    f3 L = uv_to_direction(u, v);
    float sinTh = sinf_(FLX_PI * v);
    float directPdfUV = pdf_uv * 1.0f;
    if (sinTh != 0.0f) e.pdfW = directPdfUV / (2.0f * FLX_PI * FLX_PI * sinTh);
    else e.pdfW = 0.0f;
    e.L = normalize(L);
    e.Li = eval_env_dir(sc, e.L);
    return e;
}

// reference: src/env_map.cl:95-107
__device__ __forceinline__ float env_map_pdf(const Scene &sc, f3 direction)
{
    int width = sc.envW, height = sc.envH;
    f2 uv = direction_to_uv(direction);
    float sinTh = sinf_(uv.y * FLX_PI);
    if (sinTh == 0.0f) return 0.0f;
    int iu = (int)floorf(uv.x * (float)width); if (iu > width - 1) iu = width - 1;
    int iv = (int)floorf(uv.y * (float)height); if (iv > height - 1) iv = height - 1;
    return sc.pdfTable[iv * width + iu] / (FLX_2PI * FLX_PI * sinTh);
}

// ---- camera ray of a (re)generated path (reference: genRays, src/wf_raygen.cl:22-66): jittered pixel position, pinhole direction, thin-lens origin.
// Shared by k_raygen (misc.hip) and the fused logic pass's in-kernel regeneration (logic.hip).  localIdx: the rank's local pixel (cursor + queue index).
// In three parts, because only ONE of them needs the pixel: the jitter (two draws) and the thin-lens origin (two more draws) are functions of the path's seed
// alone, the direction needs the pixel = cursor + index in the raygen queue.  PREPARED REGENERATION (logic.hip / misc.hip) computes the first two in the fused
// logic pass, where the terminating lane's origin / throughput + seed records ride on full-line stores, and leaves genRays the direction and the pixel.
__device__ __forceinline__ f3 camera_lens_origin(const flx_render_params &p, uint32_t *seedp)      // draws 3 and 4: uniformSampleDisk, src/utils.cl:75-80
{
    const float sqrt_r = sqrtf(rand01(seedp));
    const float th = FLX_2PI * rand01(seedp);
    float sn, cs; sincosf_(th, &sn, &cs);
    const f2 rnd = mk2(sqrt_r * cs, sqrt_r * sn);
    return V(p.camera.pos) + p.worldRadius * p.camera.apertureSize * (V(p.camera.right) * rnd.x + V(p.camera.up) * rnd.y);
}
__device__ __forceinline__ f3 camera_direction(const Frame &fr, const flx_render_params &p, uint32_t localIdx, float jx, float jy, f3 lensOrig)
{
    const uint32_t pixelIdx = localIdx * fr.nranks + fr.rank;
    float x = (float)(pixelIdx % p.width);
    float y = (float)(pixelIdx / p.width);
    x += jx;
    y += jy;
    float NDCx = x / (float)p.width;
    float NDCy = y / (float)p.height;
    float SCRx = 2.0f * NDCx - 1.0f;
    float SCRy = 2.0f * NDCy - 1.0f;
    SCRx *= (float)p.width / (float)p.height;
    const float scale = tanf_(0.5f * p.camera.fov * FLX_PI / 180.0f);
    SCRx *= scale;
    SCRy *= scale;
    f3 rayOrig = V(p.camera.pos);
    f3 rayTarget = rayOrig + V(p.camera.right) * SCRx + V(p.camera.up) * SCRy + V(p.camera.dir);
    f3 rayDirection = normalize(rayTarget - rayOrig);
    const f3 fp = V(p.camera.pos) + rayDirection * p.camera.focalDist;
    return normalize(fp - lensOrig);
}
__device__ __forceinline__ void camera_ray(const Frame &fr, const flx_render_params &p, uint32_t localIdx, uint32_t *seedp, f3 *orig, f3 *dir)
{
    uint32_t seed = *seedp;
    const float jx = rand01(&seed);
    const float jy = rand01(&seed);
    const f3 lens = camera_lens_origin(p, &seed);
    *dir = camera_direction(fr, p, localIdx, jx, jy, lens);
    *seedp = seed; *orig = lens;
}

} // namespace flxd
