// microkernel.hip -- the reference's second integrator ("LuxRender-style microkernels": one path per pixel, a per-path
// phase state machine instead of queues), SURVEY 8(f) N3.  It is the path Tracer::renderSingle(spp) uses because it
// guarantees exactly one sample per pixel per pass (reference: src/tracer.cpp:95-169).
//
// Replaces reference kernels reset (src/mk_reset.cl:3-43), genCameraRays (src/mk_raygen.cl:4-63), nextVertex
// (src/mk_next_vertex.cl:7-123), sampleBsdf (src/mk_sample_bsdf.cl:8-197), splat (src/mk_splat.cl:4-41) and splatPreview
// (src/mk_splat_preview.cl:3-25).  Same packed path state, traversal (flx_trace.h) and BSDFs (flx_bsdf.h) as the
// wavefront path; gid = pixel = path, so the framebuffer is written without atomics.
#include "flx_trace.h"
#include "flx_bsdf.h"

namespace flxd {

#define MK_BLOCK TRACE_BLOCK      // the traversal stack's LDS layout is [level][TRACE_BLOCK lanes]
enum { MK_RT_NEXT_VERTEX = 0, MK_SAMPLE_BSDF = 1, MK_SPLAT_SAMPLE = 4, MK_GENERATE_CAMERA_RAY = 5 };

__device__ __forceinline__ uint32_t mk_limit(const State &st, const flx_render_params &p)
{
    uint32_t npix = p.width * p.height;
    return npix < st.numTasks ? npix : st.numTasks;
}

// one atomic per wave and counter (the reference does one atomic_inc per work-item, NVIDIA: per warp)
__device__ __forceinline__ void wave_count(uint32_t *counter, bool pred)
{
    const uint64_t m = __ballot(pred);
    if (m != 0ull && lane_id() == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(counter, (uint32_t)__popcll(m));
}

__global__ __launch_bounds__(256) void k_mk_reset(State st, Frame fr, flx_render_params p)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= mk_limit(st, p)) return;
    reinterpret_cast<float4 *>(fr.pixels)[gid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (fr.aovNormal) {                                                                           // src/mk_reset.cl:24-25
        reinterpret_cast<float4 *>(fr.aovNormal)[gid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        reinterpret_cast<float4 *>(fr.aovAlbedo)[gid] = make_float4(0.1f, 0.1f, 0.1f, 0.0f);
    }
    st.phase[gid] = MK_GENERATE_CAMERA_RAY;
    float4 ei = rd4(st.at(S_EI, gid)); wr4(st.at(S_EI, gid), mk4u(mk3(0.0f), __float_as_uint(ei.w) & ~FLX_FRESH));   // (drops a wavefront-path flag, flx_device.h)
    wr4(st.at(S_THR, gid), mk4u(mk3(1.0f), gid));                                           // T = 1, seed = gid
    float4 d = rd4(st.at(S_DIR, gid)); d.w = __uint_as_float(0u); wr4(st.at(S_DIR, gid), d);      // pathLen
    float4 lt = rd4(st.at(S_LT, gid)); lt.w = __uint_as_float(1u); wr4(st.at(S_LT, gid), lt);     // lastSpecular
    float4 o = rd4(st.at(S_ORIG, gid)); o.w = 1.0f; wr4(st.at(S_ORIG, gid), o);                   // lastPdfW
    st.firstDiffuse[gid] = 0u;
}

__global__ __launch_bounds__(256) void k_mk_raygen(State st, flx_render_params p)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= mk_limit(st, p) || st.phase[gid] != MK_GENERATE_CAMERA_RAY) return;
    const float4 thr = rd4(st.at(S_THR, gid));
    uint32_t seed = __float_as_uint(thr.w);
    float x = (float)(gid % p.width), y = (float)(gid / p.width);
    x += rand01(&seed);
    y += rand01(&seed);
    float NDCx = x / (float)p.width, NDCy = y / (float)p.height;
    float SCRx = 2.0f * NDCx - 1.0f, SCRy = 2.0f * NDCy - 1.0f;
    SCRx *= (float)p.width / (float)p.height;
    const float scale = tanf_(0.5f * p.camera.fov * FLX_PI / 180.0f);
    SCRx *= scale; SCRy *= scale;
    f3 rayOrig = V(p.camera.pos);
    f3 rayTarget = rayOrig + V(p.camera.right) * SCRx + V(p.camera.up) * SCRy + V(p.camera.dir);
    f3 rayDirection = normalize(rayTarget - rayOrig);
    const f3 fp = V(p.camera.pos) + rayDirection * p.camera.focalDist;
    const float sqrt_r = sqrtf(rand01(&seed));
    const float th = FLX_2PI * rand01(&seed);
    float sn, cs; sincosf_(th, &sn, &cs);
    const f2 rnd = mk2(sqrt_r * cs, sqrt_r * sn);
    rayOrig = rayOrig + p.worldRadius * p.camera.apertureSize * (V(p.camera.right) * rnd.x + V(p.camera.up) * rnd.y);
    rayDirection = normalize(fp - rayOrig);
    const float4 o = rd4(st.at(S_ORIG, gid)), d = rd4(st.at(S_DIR, gid));
    wr4(st.at(S_ORIG, gid), mk4(rayOrig, o.w));
    wr4(st.at(S_DIR, gid), mk4(rayDirection, d.w));
    wr4(st.at(S_THR, gid), mk4u(ld3(thr), seed));
    st.phase[gid] = MK_RT_NEXT_VERTEX;
}

__global__ __launch_bounds__(MK_BLOCK) void k_mk_next_vertex(State st, Scene sc, Frame fr, flx_render_params p, uint32_t *spill, uint32_t totalThreads, uint32_t *stats)
{
    __shared__ uint32_t s_stack[LDS_LEVELS * MK_BLOCK];
    const uint32_t gid = blockIdx.x * MK_BLOCK + threadIdx.x;
    const bool active = gid < mk_limit(st, p) && st.phase[gid] == MK_RT_NEXT_VERTEX;
    bool primary = false;
    if (active) {
        const float4 o4 = rd4(st.at(S_ORIG, gid)), d4 = rd4(st.at(S_DIR, gid));
        const f3 orig = ld3(o4), dir = ld3(d4);
        Stack stk; stk.lds = s_stack + threadIdx.x; stk.stride = totalThreads; stk.spill = spill + gid;
        float t = FLX_FLT_MAX, u = 0.0f, v = 0.0f; int tri = -1; uint32_t a = 0, b = 0;
        traverse<false, false>(sc, stk, orig, dir, t, u, v, tri, a, b);
        f3 P = mk3(0.0f), N = mk3(0.0f); float tu = 0.0f, tv = 0.0f; int matId = -1; uint32_t flags = 0;
        if (tri >= 0) {
            const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri);
            const float4 sa = sp[0], sb = sp[1], scn = sp[2], sd = sp[3];
            P = orig + t * dir;
            N = normalize(bary(u, v, ld3(sa), ld3(sb), ld3(scn)));
            const f3 uv = bary(u, v, mk3(sa.w, sb.w, 0.0f), mk3(scn.w, sd.x, 0.0f), mk3(sd.y, sd.z, 0.0f));
            tu = uv.x; tv = uv.y; matId = __float_as_int(sd.w);
        }
        if (p.sampleImpl && p.useAreaLight && light_quad(p.areaLight, orig, dir, &t)) { flags = 1u; P = orig + t * dir; N = V(p.areaLight.N); tri = 0; matId = 0; }
        const uint32_t keep = __float_as_uint(reinterpret_cast<const float *>(st.at(S_HITN, gid))[3]) & 2u;
        wr4(st.at(S_HITP, gid), mk4(P, t));
        wr4(st.at(S_HITN, gid), mk4u(N, flags | keep));
        wr4(st.at(S_HITUV, gid), make_float4(tu, tv, __int_as_float(tri), __int_as_float(matId)));
        uint32_t len = __float_as_uint(d4.w);
        primary = len == 0u;
        len += 1u;
        wr4(st.at(S_DIR, gid), mk4u(dir, len));
        if (fr.aovNormal && len == 1u) {                             // first-hit normal, camera space (src/mk_next_vertex.cl:59-69); thread = pixel
            float4 *px = reinterpret_cast<float4 *>(fr.aovNormal) + gid;
            const float4 acc = *px; const f3 n = camera_space_normal(p, N);
            *px = make_float4(acc.x + n.x, acc.y + n.y, acc.z + n.z, acc.w + 1.0f);
        }
        uint32_t phase = MK_SAMPLE_BSDF;
        if (tri < 0) {                                               // miss: environment (src/mk_next_vertex.cl:72-93)
            f3 bg = mk3(0.0f);
            if (p.useEnvMap && (len == 1u || p.sampleImpl)) bg = eval_env_dir(sc, dir) * p.envMapStrength;
            float weight = 1.0f;
            const bool lastSpecular = __float_as_uint(rd4(st.at(S_LT, gid)).w) != 0u;
            if (p.sampleImpl && p.sampleExpl && p.useEnvMap && len > 1u && !lastSpecular) {
                const float lightPickProb = 1.0f;
                const float directPdfW = env_map_pdf(sc, dir);
                const float actualPdfW = o4.w;
                weight = (actualPdfW * lightPickProb) / (actualPdfW * lightPickProb + directPdfW);
            }
            const f3 T = ld3(rd4(st.at(S_THR, gid)));
            const float4 ei = rd4(st.at(S_EI, gid));
            wr4(st.at(S_EI, gid), mk4(ld3(ei) + weight * T * bg, ei.w));
            phase = MK_SPLAT_SAMPLE;
        } else if (flags & 1u) {                                     // implicit area-light hit (:94-113)
            float misWeight = 1.0f;
            const bool lastSpecular = __float_as_uint(rd4(st.at(S_LT, gid)).w) != 0u;
            if (p.sampleExpl && len > 1u && !lastSpecular) {
                const float directPdfA = 1.0f / (4.0f * p.areaLight.size.x * p.areaLight.size.y);
                const float directPdfW = pdf_a_to_w(directPdfA, length(P - orig), dot(normalize(-dir), N));
                const float lightPickProb = 1.0f;
                const float lastPdfW = o4.w;
                misWeight = lastPdfW / (lastPdfW + directPdfW * lightPickProb);
            }
            const f3 T = ld3(rd4(st.at(S_THR, gid)));
            const float4 ei = rd4(st.at(S_EI, gid));
            wr4(st.at(S_EI, gid), mk4(ld3(ei) + T * misWeight * V(p.areaLight.E), ei.w));
            phase = MK_SPLAT_SAMPLE;
        }
        st.phase[gid] = phase;
    }
    wave_count(&stats[0], active && primary);
    wave_count(&stats[1], active && !primary);
}

__global__ __launch_bounds__(MK_BLOCK) void k_mk_sample_bsdf(State st, Scene sc, Frame fr, flx_render_params p, uint32_t *spill, uint32_t totalThreads, uint32_t *stats)
{
    __shared__ uint32_t s_stack[LDS_LEVELS * MK_BLOCK];
    const uint32_t gid = blockIdx.x * MK_BLOCK + threadIdx.x;
    const bool active = gid < mk_limit(st, p) && st.phase[gid] == MK_SAMPLE_BSDF;
    uint32_t nShadow = 0;
    if (active) {
        const float4 thr = rd4(st.at(S_THR, gid));
        uint32_t seed = __float_as_uint(thr.w);
        const float4 d4 = rd4(st.at(S_DIR, gid)), hp = rd4(st.at(S_HITP, gid)), hn = rd4(st.at(S_HITN, gid)), huv = rd4(st.at(S_HITUV, gid));
        const f3 rayDir = ld3(d4);
        const int hitI = __float_as_int(huv.z);
        const flx_material &gm = sc.materials[__float_as_int(huv.w)];
        const Mat m = load_mat(gm);
        SurfHit h; h.P = ld3(hp); h.uv = mk2(huv.x, huv.y);
        h.N = tangent_space_normal(sc, ld3(hn), h.uv, hitI, gm.map_N);
        const bool backface = dot(h.N, rayDir) > 0.0f;
        if (backface) h.N = h.N * -1.0f;
        f3 orig = h.P - 1e-3f * rayDir;
        if (fr.aovAlbedo && !FLX_BXDF_IS_SINGULAR(m.type) && !st.firstDiffuse[gid]) {            // src/mk_sample_bsdf.cl:56-66
            st.firstDiffuse[gid] = 1u;
            float4 *px = reinterpret_cast<float4 *>(fr.aovAlbedo) + gid;
            const float4 acc = *px; const f3 al = mat_float3(sc, V(gm.Kd), h.uv, gm.map_Kd);
            *px = make_float4(acc.x + al.x, acc.y + al.y, acc.z + al.z, acc.w + 1.0f);
        }
        f3 Ei = ld3(rd4(st.at(S_EI, gid)));
        const float eiw = rd4(st.at(S_EI, gid)).w;
        const f3 T = ld3(thr);
        Stack stk; stk.lds = s_stack + threadIdx.x; stk.stride = totalThreads; stk.spill = spill + gid;
        if (p.sampleExpl && !FLX_BXDF_IS_SINGULAR(m.type)) {         // next event estimation, both lights (src/mk_sample_bsdf.cl:62-141)
            const float lightPickProb = 1.0f;
            if (p.useEnvMap) {
                f3 L; float directPdfW = 0.0f;
                sample_env_alias(sc, rand01(&seed), &L, &directPdfW);
                const float lenL = 2.0f * p.worldRadius;
                L = normalize(L);
                bool occluded = false;
                if (p.useAreaLight) { float tl = lenL; occluded = light_quad(p.areaLight, orig, L, &tl); }
                if (!occluded) { float t = lenL, u, v; int tri; uint32_t a = 0, b = 0; occluded = traverse<true, false>(sc, stk, orig, L, t, u, v, tri, a, b); }
                nShadow++;
                if (!occluded && directPdfW != 0.0f) {
                    const f3 brdf = bxdf_eval(sc, h, m, backface, rayDir, L);
                    const float cosTh = fmaxf_(0.0f, dot(L, h.N));
                    const float bsdfPdfW = fmaxf_(0.0f, bxdf_pdf(sc, h, m, backface, rayDir, L));
                    float weight = 1.0f;
                    if (p.sampleImpl) weight = (directPdfW * lightPickProb) / (directPdfW * lightPickProb + bsdfPdfW);
                    const f3 envMapLi = eval_env_dir(sc, L) * p.envMapStrength;
                    Ei = Ei + brdf * T * envMapLi * weight * cosTh / (lightPickProb * directPdfW);
                }
            }
            if (p.useAreaLight) {
                const float directPdfA = 1.0f / (4.0f * p.areaLight.size.x * p.areaLight.size.y);
                f3 posL = V(p.areaLight.pos);
                const float r1 = 2.0f * rand01(&seed) - 1.0f;
                const float r2 = 2.0f * rand01(&seed) - 1.0f;
                posL = posL + r1 * p.areaLight.size.x * V(p.areaLight.right);
                posL = posL + r2 * p.areaLight.size.y * V(p.areaLight.up);
                f3 L = posL - orig;
                const float lenL = length(L);
                L = normalize(L);
                float t = lenL, u, v; int tri; uint32_t a = 0, b = 0;
                const bool occluded = traverse<true, false>(sc, stk, orig, L, t, u, v, tri, a, b);
                nShadow++;
                const float cosLight = fmaxf_(dot(V(p.areaLight.N), -L), 0.0f);
                if (!occluded && cosLight > 0.0f) {
                    const f3 brdf = bxdf_eval(sc, h, m, backface, rayDir, L);
                    const float cosTh = fmaxf_(0.0f, dot(L, h.N));
                    const float directPdfW = pdf_a_to_w(directPdfA, lenL, cosLight);
                    const float bsdfPdfW = fmaxf_(0.0f, bxdf_pdf(sc, h, m, backface, rayDir, L));
                    float weight = 1.0f;
                    if (p.sampleImpl) weight = (directPdfW * lightPickProb) / (directPdfW * lightPickProb + bsdfPdfW);
                    Ei = Ei + brdf * T * V(p.areaLight.E) * weight * cosTh / (lightPickProb * directPdfW);
                }
            }
        }
        float contProb = 1.0f;
        const uint32_t len = __float_as_uint(d4.w);
        bool terminate = (len - 1u >= p.maxBounces);
        if (terminate && p.useRoulette) {
            contProb = clampf(luminance(T), 0.01f, 0.5f);
            terminate = (rand01(&seed) > contProb);
        }
        float pdfW = 0.0f; f3 newDir = mk3(0.0f);
        const f3 bsdf = bxdf_sample(sc, h, m, backface, rayDir, &newDir, &pdfW, &seed);
        const float costh = dot(h.N, normalize(newDir));
        pdfW *= contProb;
        if (pdfW == 0.0f || is_zero(bsdf)) terminate = true;
        const f3 newT = T * bsdf * costh / pdfW;
        orig = h.P + 1e-4f * newDir;
        wr4(st.at(S_EI, gid), mk4(Ei, eiw));
        wr4(st.at(S_THR, gid), mk4u(newT, seed));
        wr4(st.at(S_ORIG, gid), mk4(orig, pdfW));
        wr4(st.at(S_DIR, gid), mk4(newDir, d4.w));
        float4 lt = rd4(st.at(S_LT, gid)); lt.w = __uint_as_float(FLX_BXDF_IS_SINGULAR(m.type) ? 1u : 0u); wr4(st.at(S_LT, gid), lt);
        st.phase[gid] = terminate ? MK_SPLAT_SAMPLE : MK_RT_NEXT_VERTEX;
    }
    for (int o = 32; o > 0; o >>= 1) nShadow += __shfl_xor(nShadow, o, 64);
    if (lane_id() == 0u && nShadow) atomicAdd(&stats[2], nShadow);
}

__global__ __launch_bounds__(256) void k_mk_splat(State st, Frame fr, flx_render_params p, uint32_t *stats, int preview)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const bool inRange = gid < mk_limit(st, p);
    const bool active = inRange && (preview || st.phase[gid] == MK_SPLAT_SAMPLE);
    if (active) {
        const float4 ei = rd4(st.at(S_EI, gid));
        float4 *px = reinterpret_cast<float4 *>(fr.pixels) + gid;
        if (preview) *px = make_float4(ei.x, ei.y, ei.z, 0.0f);     // alpha 0 => overwritten by the next real sample
        else {
            float4 col = make_float4(ei.x, ei.y, ei.z, 1.0f);
            const float4 prev = *px;
            if (prev.w > 0.0f) { col.x += prev.x; col.y += prev.y; col.z += prev.z; col.w += prev.w; }
            *px = col;
        }
        wr4(st.at(S_EI, gid), make_float4(0.0f, 0.0f, 0.0f, ei.w));
        const float4 thr = rd4(st.at(S_THR, gid)); wr4(st.at(S_THR, gid), mk4u(mk3(1.0f), __float_as_uint(thr.w)));
        float4 d = rd4(st.at(S_DIR, gid)); d.w = __uint_as_float(0u); wr4(st.at(S_DIR, gid), d);
        if (!preview) st.firstDiffuse[gid] = 0u;
        st.phase[gid] = MK_GENERATE_CAMERA_RAY;
    }
    if (!preview) wave_count(&stats[3], active);
}

static uint32_t mkThreads(const State &st, const flx_render_params &p) { uint32_t n = p.width * p.height; return n < st.numTasks ? n : st.numTasks; }

void launch_mk_reset(hipStream_t s, const State &st, const Frame &fr, const flx_render_params &p)
{ hipLaunchKernelGGL(k_mk_reset, dim3((mkThreads(st, p) + 255) / 256), dim3(256), 0, s, st, fr, p); }
void launch_mk_raygen(hipStream_t s, const State &st, const flx_render_params &p)
{ hipLaunchKernelGGL(k_mk_raygen, dim3((mkThreads(st, p) + 255) / 256), dim3(256), 0, s, st, p); }
void launch_mk_next_vertex(hipStream_t s, const State &st, const Scene &sc, const Frame &fr, const flx_render_params &p, uint32_t *spill, uint32_t *stats)
{
    uint32_t blocks = (mkThreads(st, p) + MK_BLOCK - 1) / MK_BLOCK;
    hipLaunchKernelGGL(k_mk_next_vertex, dim3(blocks), dim3(MK_BLOCK), 0, s, st, sc, fr, p, spill, blocks * MK_BLOCK, stats);
}
void launch_mk_sample_bsdf(hipStream_t s, const State &st, const Scene &sc, const Frame &fr, const flx_render_params &p, uint32_t *spill, uint32_t *stats)
{
    uint32_t blocks = (mkThreads(st, p) + MK_BLOCK - 1) / MK_BLOCK;
    hipLaunchKernelGGL(k_mk_sample_bsdf, dim3(blocks), dim3(MK_BLOCK), 0, s, st, sc, fr, p, spill, blocks * MK_BLOCK, stats);
}
void launch_mk_splat(hipStream_t s, const State &st, const Frame &fr, const flx_render_params &p, uint32_t *stats, int preview)
{ hipLaunchKernelGGL(k_mk_splat, dim3((mkThreads(st, p) + 255) / 256), dim3(256), 0, s, st, fr, p, stats, preview); }

} // namespace flxd
