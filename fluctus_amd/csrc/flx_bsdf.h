// flx_bsdf.h -- the six BSDFs of the path (eval / pdf / sample) as device functions, shared by the wavefront material
// kernels (material.hip) and the microkernel integrator (microkernel.hip).
// Reference: src/diffuse.cl, glossy.cl, ggx.cl, ideal_reflection.cl, ideal_dielectric.cl, fresnel.cl, bxdf_partial.cl.
// Arithmetic: include/flx_math.h contract (bit-identical to the oracle).
#pragma once
#include "flx_shading.h"

namespace flxd {

struct SurfHit { f3 P, N; f2 uv; };

// ---- fresnel (src/fresnel.cl:5-20)
__device__ __forceinline__ float fresnel_dielectric(float cosThI, float etaI, float etaT)
{
    float sinThetaI = sqrtf(fmaxf_(0.0f, 1.0f - cosThI * cosThI));
    float sinThetaT = etaI / etaT * sinThetaI;
    float cosThetaT = sqrtf(fmaxf_(0.0f, 1.0f - sinThetaT * sinThetaT));
    if (sinThetaT >= 1.0f) return 1.0f;
    float parl = ((etaT * cosThI) - (etaI * cosThetaT)) / ((etaT * cosThI) + (etaI * cosThetaT));
    float perp = ((etaI * cosThI) - (etaT * cosThetaT)) / ((etaI * cosThI) + (etaT * cosThetaT));
    return 0.5f * (parl * parl + perp * perp);
}

__device__ __forceinline__ f3 reflect3(f3 dir, f3 n) { return dir - 2.0f * dot(dir, n) * n; }
__device__ __forceinline__ f3 refract3(f3 wi, f3 n, float eta)
{
    float iDotN = dot(-wi, n);
    float sin2ThetaI = fmaxf_(0.0f, 1.0f - iDotN * iDotN);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    float cosThetaT = sqrtf(fmaxf_(0.0f, 1.0f - sin2ThetaT));
    return wi * eta + n * (eta * iDotN - cosThetaT);
}

// ---- diffuse (src/diffuse.cl:9-26, src/utils.cl:83-112)
__device__ __forceinline__ f3 cos_sample_hemisphere(f3 n, uint32_t *seed, float *p)
{
    float r1 = 2.0f * FLX_PI * rand01(seed);
    float r2 = rand01(seed);
    float r2s = sqrtf(r2);
    f3 w = n, u;
    if (absf(w.x) > 0.1f) u = cross(mk3(0.0f, 1.0f, 0.0f), w);
    else u = cross(mk3(1.0f, 0.0f, 0.0f), w);
    u = normalize(u);
    f3 v = cross(w, u);
    float s, co; sincosf_(r1, &s, &co);
    u = u * (co * r2s);
    v = v * (s * r2s);
    w = w * sqrtf(1.0f - r2);
    f3 dir = u + v + w;
    *p = dot(n, dir) / FLX_PI;
    return dir;
}
__device__ __forceinline__ f3 eval_diffuse(const Scene &sc, const SurfHit &h, f3 Kd, int mapKd) { return mat_albedo(sc, Kd, h.uv, mapKd) * FLX_INV_PI; }
__device__ __forceinline__ float pdf_diffuse(const SurfHit &h, f3 dirOut) { return dot(h.N, dirOut) * FLX_INV_PI; }
__device__ __forceinline__ f3 sample_diffuse(const Scene &sc, const SurfHit &h, f3 Kd, int mapKd, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    *dirOut = cos_sample_hemisphere(h.N, seed, pdfW);
    return mat_albedo(sc, Kd, h.uv, mapKd) * FLX_INV_PI;
}

// ---- GGX (src/ggx.cl)
__device__ __forceinline__ float to_roughness(float shininess) { return sqrtf(2.0f / (2.0f + shininess)); }

__device__ __forceinline__ f3 ggx_sample_lobe(float alpha, f3 N, uint32_t *seed)
{
    f3 X, Y, Z = N;
    if (N.x != N.y || N.x != N.z) X = mk3(N.z - N.y, N.x - N.z, N.y - N.x);   // makeOrthoBasis, src/utils.cl:50-59
    else X = mk3(N.z - N.y, N.x + N.z, -N.y - N.x);
    X = normalize(X);
    Y = cross(N, X);
    float rx = rand01(seed);
    float ry = rand01(seed);
    float theta = atan2f_(alpha * sqrtf(rx), sqrtf(1.0f - rx));
    float phi = FLX_2PI * ry;
    float sinTheta, cosTheta, sinPhi, cosPhi;
    sincosf_(theta, &sinTheta, &cosTheta);
    sincosf_(phi, &sinPhi, &cosPhi);
    return normalize(X * sinTheta * cosPhi + Y * sinTheta * sinPhi + Z * cosTheta);
}
__device__ __forceinline__ float ggx_g1(float alpha, f3 v, f3 n, f3 m)
{
    float mDotV = dot(m, v), nDotV = dot(n, v);
    if (nDotV * mDotV <= 0.0f) return 0.0f;
    float cosThSq = nDotV * nDotV;
    float tanSq = (cosThSq > 0.0f) ? ((1.0f - cosThSq) / cosThSq) : 0.0f;
    return 2.0f / (1.0f + sqrtf(1.0f + alpha * alpha * tanSq));
}
__device__ __forceinline__ float ggx_g(float alpha, f3 dirIn, f3 dirOut, f3 n, f3 m) { return ggx_g1(alpha, dirIn, n, m) * ggx_g1(alpha, dirOut, n, m); }
__device__ __forceinline__ float ggx_d(float alpha, f3 n, f3 m)
{
    float nDotM = dot(n, m);
    if (nDotM <= 0.0f) return 0.0f;
    float nDotMSq = nDotM * nDotM;
    float tanSq = nDotM != 0.0f ? ((1.0f - nDotMSq) / nDotMSq) : 0.0f;
    float aSq = alpha * alpha;
    float denom = FLX_PI * nDotMSq * nDotMSq * (aSq + tanSq) * (aSq + tanSq);
    return denom > 0.0f ? (aSq / denom) : 0.0f;
}
__device__ __forceinline__ float ggx_pdf_reflect(float alpha, f3 dirOut, f3 N, f3 H)
{
    float nDotH = absf(dot(N, H));
    float oDotH = absf(dot(dirOut, H));
    float jInv = 4.0f * oDotH;
    return jInv == 0.0f ? 0.0f : ggx_d(alpha, N, H) * nDotH / jInv;
}
__device__ __forceinline__ float ggx_pdf_refract(float alpha, float etaI, float etaO, f3 dirIn, f3 dirOut, f3 N, f3 H)
{
    float nDotH = absf(dot(N, H));
    float iDotH = absf(dot(dirIn, H));
    float oDotH = absf(dot(dirOut, H));
    float sqrtJInv = etaI * iDotH + etaO * oDotH;
    return sqrtJInv == 0.0f ? 0.0f : ggx_d(alpha, N, H) * nDotH * oDotH * etaO * etaO / (sqrtJInv * sqrtJInv);
}

struct Mat {    // the fields of the 80-byte Material the BSDFs use
    f3 Kd, Ks; float Ns, Ni; int mapKd, mapKs, type;
};

// src/ggx.cl:89-113
__device__ __forceinline__ f3 sample_ggx_reflect(const Scene &sc, const SurfHit &h, f3 Ks, int mapKs, float Ns, float Ni, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    dirIn = dirIn * -1.0f;
    float alpha = to_roughness(Ns);
    f3 H = ggx_sample_lobe(alpha, h.N, seed);
    *dirOut = reflect3(-dirIn, H);
    *pdfW = ggx_pdf_reflect(alpha, *dirOut, h.N, H);
    float iDotN = dot(dirIn, h.N);
    float oDotN = dot(*dirOut, h.N);
    float Fr = (Ni > 1.0f) ? fresnel_dielectric(iDotN, 1.0f, Ni) : 1.0f;
    f3 ks = mat_float3(sc, Ks, h.uv, mapKs);
    float D = ggx_d(alpha, h.N, H);
    float G = ggx_g(alpha, dirIn, *dirOut, h.N, H);
    float den = 4.0f * iDotN * oDotN;
    return (den != 0.0f) ? (ks * Fr * G * D / den) : mk3(0.0f);
}
// src/ggx.cl:115-136
__device__ __forceinline__ f3 eval_ggx_reflect(const Scene &sc, const SurfHit &h, f3 Ks, int mapKs, float Ns, float Ni, f3 dirIn, f3 dirOut)
{
    dirIn = dirIn * -1.0f;
    float alpha = to_roughness(Ns);
    f3 H = normalize(dirIn + dirOut);
    float iDotN = dot(dirIn, h.N);
    float oDotN = dot(dirOut, h.N);
    float Fr = (Ni > 1.0f) ? fresnel_dielectric(iDotN, 1.0f, Ni) : 1.0f;
    f3 ks = mat_float3(sc, Ks, h.uv, mapKs);
    float D = ggx_d(alpha, h.N, H);
    float G = ggx_g(alpha, dirIn, dirOut, h.N, H);
    float den = 4.0f * iDotN * oDotN;
    return (den != 0.0f) ? (ks * Fr * G * D / den) : mk3(0.0f);
}
// src/ggx.cl:138-144
__device__ __forceinline__ float pdf_ggx_reflect(const SurfHit &h, float Ns, f3 dirIn, f3 dirOut)
{
    dirIn = dirIn * -1.0f;
    float alpha = to_roughness(Ns);
    f3 H = normalize(dirIn + dirOut);
    return ggx_pdf_reflect(alpha, dirOut, h.N, H);
}

// src/ggx.cl:156-221
__device__ __forceinline__ f3 sample_ggx_refract(const Scene &sc, const SurfHit &h, const Mat &m, bool backface, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    dirIn = dirIn * -1.0f;
    float raylen = length(dirIn);
    float alpha = to_roughness(m.Ns);
    float etaI = 1.0f, etaO = m.Ni;
    if (backface) { float t = etaI; etaI = etaO; etaO = t; }
    float iDotN = dot(normalize(dirIn), h.N);
    f3 H = ggx_sample_lobe(alpha, h.N, seed);
    float Fr = fresnel_dielectric(iDotN, etaI, etaO);
    if (rand01(seed) < Fr) {
        *dirOut = raylen * reflect3(normalize(-dirIn), H);
        *pdfW = ggx_pdf_reflect(alpha, *dirOut, h.N, H);
        float oDotN = dot(*dirOut, h.N);
        float D = ggx_d(alpha, h.N, H);
        float G = ggx_g(alpha, dirIn, *dirOut, h.N, H);
        float den = 4.0f * iDotN * oDotN;
        return (den != 0.0f) ? mk3(Fr * G * D / den) : mk3(0.0f);
    } else {
        float eta = etaI / etaO;
        *dirOut = raylen * refract3(normalize(-dirIn), h.N, eta);
        H = normalize(-(dirIn * etaI + *dirOut * etaO));
        f3 Nn = backface ? -h.N : h.N;
        *pdfW = ggx_pdf_refract(alpha, etaI, etaO, dirIn, *dirOut, Nn, H);
        f3 bsdf = mk3(eta * eta);
        f3 ks = mat_float3(sc, m.Ks, h.uv, m.mapKs);
        bsdf = bsdf * ks;
        float iDotH = absf(dot(normalize(dirIn), H));
        float oDotH = absf(dot(*dirOut, H));
        float oDotN = dot(*dirOut, h.N);
        float focusTermDenom = iDotN * oDotN * (etaI * iDotH + etaO * oDotH) * (etaI * iDotH + etaO * oDotH);
        if (focusTermDenom == 0.0f) return mk3(0.0f);
        float focusTerm = etaO * etaO * iDotH * oDotH / focusTermDenom;
        float D = ggx_d(alpha, Nn, H);
        float G = ggx_g(alpha, dirIn, *dirOut, Nn, H);
        return (1.0f - Fr) * bsdf * D * G * focusTerm;
    }
}
// src/ggx.cl:223-271
__device__ __forceinline__ f3 eval_ggx_refract(const Scene &sc, const SurfHit &h, const Mat &m, bool backface, f3 dirIn, f3 dirOut)
{
    dirIn = dirIn * -1.0f;
    float alpha = to_roughness(m.Ns);
    float etaI = 1.0f, etaO = m.Ni;
    if (backface) { float t = etaI; etaI = etaO; etaO = t; }
    float iDotN = dot(normalize(dirIn), h.N);
    float oDotN = dot(normalize(dirOut), h.N);
    float Fr = fresnel_dielectric(iDotN, etaI, etaO);
    if (!backface) {
        f3 H = normalize(dirIn + dirOut);
        float D = ggx_d(alpha, h.N, H);
        float G = ggx_g(alpha, dirIn, dirOut, h.N, H);
        float den = 4.0f * iDotN * oDotN;
        return (den != 0.0f) ? mk3(Fr * G * D / den) : mk3(0.0f);
    } else {
        f3 H = normalize(-(dirIn * etaI + dirOut * etaO));
        float eta = etaI / etaO;
        f3 bsdf = mk3(eta * eta);
        f3 ks = mat_float3(sc, m.Ks, h.uv, m.mapKs);
        bsdf = bsdf * ks;
        float iDotH = absf(dot(normalize(dirIn), H));
        float oDotH = absf(dot(normalize(dirOut), H));
        float focusTermDenom = iDotN * oDotN * (etaI * iDotH + etaO * oDotH) * (etaI * iDotH + etaO * oDotH);
        if (focusTermDenom == 0.0f) return mk3(0.0f);
        float focusTerm = etaO * etaO * iDotH * oDotH / focusTermDenom;
        float D = ggx_d(alpha, -h.N, H);
        float G = ggx_g(alpha, dirIn, dirOut, -h.N, H);
        return (1.0f - Fr) * bsdf * D * G * focusTerm;
    }
}
// src/ggx.cl:273-292
__device__ __forceinline__ float pdf_ggx_refract(const SurfHit &h, const Mat &m, bool backface, f3 dirIn, f3 dirOut)
{
    dirIn = dirIn * -1.0f;
    float alpha = to_roughness(m.Ns);
    float etaI = 1.0f, etaO = m.Ni;
    if (!backface) {
        f3 H = normalize(dirIn + dirOut);
        return ggx_pdf_reflect(alpha, dirOut, h.N, H);
    } else {
        float t = etaI; etaI = etaO; etaO = t;
        f3 H = normalize(-(dirIn * etaI + dirOut * etaO));
        return ggx_pdf_refract(alpha, etaI, etaO, dirIn, dirOut, -h.N, H);
    }
}

// ---- glossy = fresnel-blended diffuse base + GGX coat (src/glossy.cl)
__device__ __forceinline__ f3 eta_to_ks(float eta) { float r = (eta > 0.0f) ? ((eta - 1.0f) / (eta + 1.0f)) : 0.0f; return mk3(r * r); }
__device__ __forceinline__ float ks_to_eta(f3 Ks)
{
    float k = clampf((Ks.x + Ks.y + Ks.z) / 3.0f, 0.0f, 0.99f);
    return (sqrtf(k) + 1.0f) / (1.0f - sqrtf(k));
}
// src/glossy.cl:24-64; an early-out leaves pdfW at its initial 0 (undefined in the reference)
__device__ __forceinline__ f3 sample_glossy(const Scene &sc, const SurfHit &h, const Mat &m, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    f3 Ks = mat_float3(sc, m.Ks, h.uv, m.mapKs);
    float Ni = (m.Ni > 0.0f) ? m.Ni : ks_to_eta(Ks);
    if (is_zero(Ks)) Ks = eta_to_ks(Ni);
    float cosTh = dot(normalize(-dirIn), h.N);
    float Fr = fresnel_dielectric(cosTh, 1.0f, Ni);
    float basePdf, coatingPdf;
    f3 baseBrdf, coatingBrdf;
    if (rand01(seed) < Fr) {
        coatingBrdf = sample_ggx_reflect(sc, h, Ks, m.mapKs, m.Ns, Ni, dirIn, dirOut, &coatingPdf, seed);
        baseBrdf = eval_diffuse(sc, h, m.Kd, m.mapKd);
        basePdf = pdf_diffuse(h, *dirOut);
    } else {
        baseBrdf = sample_diffuse(sc, h, m.Kd, m.mapKd, dirOut, &basePdf, seed);
        coatingBrdf = eval_ggx_reflect(sc, h, Ks, m.mapKs, m.Ns, Ni, dirIn, *dirOut);
        coatingPdf = pdf_ggx_reflect(h, m.Ns, dirIn, *dirOut);
    }
    if (dot(h.N, *dirOut) < 1e-5f) return mk3(0.0f);
    *pdfW = (1.0f - Fr) * basePdf + Fr * coatingPdf;
    return baseBrdf * (1.0f - Fr) + coatingBrdf;
}
// src/glossy.cl:66-85
__device__ __forceinline__ f3 eval_glossy(const Scene &sc, const SurfHit &h, const Mat &m, f3 dirIn, f3 dirOut)
{
    f3 Ks = mat_float3(sc, m.Ks, h.uv, m.mapKs);
    float Ni = (m.Ni > 0.0f) ? m.Ni : ks_to_eta(Ks);
    if (length(Ks) == 0.0f) Ks = eta_to_ks(Ni);
    f3 baseBrdf = eval_diffuse(sc, h, m.Kd, m.mapKd);
    f3 coatingBrdf = eval_ggx_reflect(sc, h, Ks, m.mapKs, m.Ns, Ni, dirIn, dirOut);
    float cosTh = dot(normalize(-dirIn), h.N);
    float Fr = fresnel_dielectric(cosTh, 1.0f, Ni);
    return baseBrdf * (1.0f - Fr) + coatingBrdf;
}
// src/glossy.cl:87-101
__device__ __forceinline__ float pdf_glossy(const Scene &sc, const SurfHit &h, const Mat &m, f3 dirIn, f3 dirOut)
{
    f3 Ks = mat_float3(sc, m.Ks, h.uv, m.mapKs);
    float Ni = (m.Ni > 0.0f) ? m.Ni : ks_to_eta(Ks);
    float basePdf = pdf_diffuse(h, dirOut);
    float coatingPdf = pdf_ggx_reflect(h, m.Ns, dirIn, dirOut);
    float cosTh = dot(normalize(-dirIn), h.N);
    float Fr = fresnel_dielectric(cosTh, 1.0f, Ni);
    return (1.0f - Fr) * basePdf + Fr * coatingPdf;
}

// ---- delta BSDFs (src/ideal_reflection.cl:9-22, src/ideal_dielectric.cl:10-45)
__device__ __forceinline__ f3 sample_ideal_reflection(const Scene &sc, const SurfHit &h, const Mat &m, f3 dirIn, f3 *dirOut, float *pdfW)
{
    float len = length(dirIn);
    *dirOut = len * reflect3(normalize(dirIn), h.N);
    *pdfW = 1.0f;
    f3 ks = mat_float3(sc, m.Ks, h.uv, m.mapKs);
    float cosO = dot(normalize(*dirOut), h.N);
    return (cosO != 0.0f) ? ks / cosO : mk3(0.0f);
}
__device__ __forceinline__ f3 sample_ideal_dielectric(const Scene &sc, const SurfHit &h, const Mat &m, bool backface, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    float raylen = length(dirIn);
    f3 bsdf = mk3(1.0f);
    float cosI = dot(normalize(-dirIn), h.N);
    float n1 = 1.0f, n2 = m.Ni;
    if (backface) { float t = n1; n1 = n2; n2 = t; }
    float eta = n1 / n2;
    float fr = fresnel_dielectric(cosI, n1, n2);
    if (rand01(seed) < fr) {
        *dirOut = raylen * reflect3(normalize(dirIn), h.N);
    } else {
        *dirOut = raylen * refract3(normalize(dirIn), h.N, eta);
        bsdf = bsdf * (eta * eta);
        f3 ks = mat_float3(sc, m.Ks, h.uv, m.mapKs);
        bsdf = bsdf * ks;
    }
    *pdfW = 1.0f;
    float cosO = dot(normalize(*dirOut), h.N);
    return bsdf / cosO;
}

__device__ __forceinline__ Mat load_mat(const flx_material &gm)
{
    Mat m; m.Kd = V(gm.Kd); m.Ks = V(gm.Ks); m.Ns = gm.Ns; m.Ni = gm.Ni; m.mapKd = gm.map_Kd; m.mapKs = gm.map_Ks; m.type = gm.type;
    return m;
}

// bxdfEval / bxdfPdf / bxdfSample with every BSDF compiled in (reference: src/bxdf_partial.cl:19-153)
__device__ __forceinline__ f3 bxdf_eval(const Scene &sc, const SurfHit &h, const Mat &m, bool backface, f3 dirIn, f3 dirOut)
{
    switch (m.type) {
    case FLX_BXDF_DIFFUSE: return eval_diffuse(sc, h, m.Kd, m.mapKd);
    case FLX_BXDF_GLOSSY: return eval_glossy(sc, h, m, dirIn, dirOut);
    case FLX_BXDF_GGX_ROUGH_REFLECTION: return eval_ggx_reflect(sc, h, m.Ks, m.mapKs, m.Ns, m.Ni, dirIn, dirOut);
    case FLX_BXDF_GGX_ROUGH_DIELECTRIC: return eval_ggx_refract(sc, h, m, backface, dirIn, dirOut);
    }
    return mk3(0.0f);
}
__device__ __forceinline__ float bxdf_pdf(const Scene &sc, const SurfHit &h, const Mat &m, bool backface, f3 dirIn, f3 dirOut)
{
    switch (m.type) {
    case FLX_BXDF_DIFFUSE: return pdf_diffuse(h, dirOut);
    case FLX_BXDF_GLOSSY: return pdf_glossy(sc, h, m, dirIn, dirOut);
    case FLX_BXDF_GGX_ROUGH_REFLECTION: return pdf_ggx_reflect(h, m.Ns, dirIn, dirOut);
    case FLX_BXDF_GGX_ROUGH_DIELECTRIC: return pdf_ggx_refract(h, m, backface, dirIn, dirOut);
    }
    return 0.0f;
}
__device__ __forceinline__ f3 bxdf_sample(const Scene &sc, const SurfHit &h, const Mat &m, bool backface, f3 dirIn, f3 *dirOut, float *pdfW, uint32_t *seed)
{
    switch (m.type) {
    case FLX_BXDF_DIFFUSE: return sample_diffuse(sc, h, m.Kd, m.mapKd, dirOut, pdfW, seed);
    case FLX_BXDF_GLOSSY: return sample_glossy(sc, h, m, dirIn, dirOut, pdfW, seed);
    case FLX_BXDF_GGX_ROUGH_REFLECTION: return sample_ggx_reflect(sc, h, m.Ks, m.mapKs, m.Ns, m.Ni, dirIn, dirOut, pdfW, seed);
    case FLX_BXDF_IDEAL_REFLECTION: return sample_ideal_reflection(sc, h, m, dirIn, dirOut, pdfW);
    case FLX_BXDF_GGX_ROUGH_DIELECTRIC: return sample_ggx_refract(sc, h, m, backface, dirIn, dirOut, pdfW, seed);
    case FLX_BXDF_IDEAL_DIELECTRIC: return sample_ideal_dielectric(sc, h, m, backface, dirIn, dirOut, pdfW, seed);
    }
    return mk3(0.0f);
}

// ---- one path's material step (src/wf_mat_diffuse.cl:30-62 and twins): f and pdf toward the stored light direction (consumed by
// `logic` next iteration), the continuation sample, the throughput update and the new ray.  Shared by the per-queue material kernels
// (material.hip) and the fused logic+material kernel (logic.hip), so both run the same arithmetic in the same order.
// USE = the BSDF types the caller can meet (the reference gets the same effect from -DBXDF_USE_* at OpenCL build time).
enum { USE_DIFFUSE = 1, USE_GLOSSY = 2, USE_GGX_REFL = 4, USE_GGX_REFR = 8, USE_DELTA = 16, USE_ALL = 31 };

struct MatStep {
    f3 bsdfNEE; float bsdfPdfW;      // -> lastBsdf, lastPdfImplicit
    f3 newT;                         // -> T
    f3 orig; float pdfW;             // -> ray origin, lastPdfW
    f3 newDir;                       // -> ray direction
    uint32_t singular;               // -> lastSpecular
};

template <int USE>
__device__ __forceinline__ MatStep material_step(const Scene &sc, const SurfHit &h, const flx_material &gm, bool backface, f3 dirIn, f3 L, f3 oldT, uint32_t *seed)
{
    Mat m; m.Kd = V(gm.Kd); m.Ks = V(gm.Ks); m.Ns = gm.Ns; m.Ni = gm.Ni; m.mapKd = gm.map_Kd; m.mapKs = gm.map_Ks; m.type = gm.type;
    MatStep o;
    // f and pdf toward the stored light direction (src/wf_mat_diffuse.cl:34-37)
    f3 bsdfNEE = mk3(0.0f); float bsdfPdfW = 0.0f;
    // continuation sample (:40-42)
    float pdfW = 0.0f; f3 newDir = mk3(0.0f); f3 bsdf = mk3(0.0f);
    if ((USE & USE_DIFFUSE) && m.type == FLX_BXDF_DIFFUSE) {
        bsdfNEE = eval_diffuse(sc, h, m.Kd, m.mapKd); bsdfPdfW = pdf_diffuse(h, L);
        bsdf = sample_diffuse(sc, h, m.Kd, m.mapKd, &newDir, &pdfW, seed);
    } else if ((USE & USE_GLOSSY) && m.type == FLX_BXDF_GLOSSY) {
        bsdfNEE = eval_glossy(sc, h, m, dirIn, L); bsdfPdfW = pdf_glossy(sc, h, m, dirIn, L);
        bsdf = sample_glossy(sc, h, m, dirIn, &newDir, &pdfW, seed);
    } else if ((USE & USE_GGX_REFL) && m.type == FLX_BXDF_GGX_ROUGH_REFLECTION) {
        bsdfNEE = eval_ggx_reflect(sc, h, m.Ks, m.mapKs, m.Ns, m.Ni, dirIn, L); bsdfPdfW = pdf_ggx_reflect(h, m.Ns, dirIn, L);
        bsdf = sample_ggx_reflect(sc, h, m.Ks, m.mapKs, m.Ns, m.Ni, dirIn, &newDir, &pdfW, seed);
    } else if ((USE & USE_GGX_REFR) && m.type == FLX_BXDF_GGX_ROUGH_DIELECTRIC) {
        bsdfNEE = eval_ggx_refract(sc, h, m, backface, dirIn, L); bsdfPdfW = pdf_ggx_refract(h, m, backface, dirIn, L);
        bsdf = sample_ggx_refract(sc, h, m, backface, dirIn, &newDir, &pdfW, seed);
    } else if ((USE & USE_DELTA) && m.type == FLX_BXDF_IDEAL_REFLECTION) {
        bsdf = sample_ideal_reflection(sc, h, m, dirIn, &newDir, &pdfW);
    } else if ((USE & USE_DELTA) && m.type == FLX_BXDF_IDEAL_DIELECTRIC) {
        bsdf = sample_ideal_dielectric(sc, h, m, backface, dirIn, &newDir, &pdfW, seed);
    }
    o.bsdfNEE = bsdfNEE;
    o.bsdfPdfW = fmaxf_(0.0f, bsdfPdfW);
    const float costh = dot(h.N, normalize(newDir));
    if (pdfW == 0.0f || is_zero(bsdf)) o.newT = mk3(0.0f);
    else o.newT = oldT * bsdf * costh / pdfW;
    o.orig = h.P + 1e-4f * newDir;                              // :53
    o.pdfW = pdfW;
    o.newDir = newDir;
    o.singular = FLX_BXDF_IS_SINGULAR(m.type) ? 1u : 0u;
    return o;
}

} // namespace flxd
