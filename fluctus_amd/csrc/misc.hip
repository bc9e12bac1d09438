// misc.hip -- reset, camera-ray generation, post-process and the state import/export test hooks.
//
// Replaces reference kernels reset (src/wf_reset.cl:5-66), genRays (src/wf_raygen.cl:4-97) and
// process (src/mk_postprocess.cl:7-55, src/tonemap.cl:3-26).
#include "flx_shading.h"

namespace flxd {

#ifndef MISC_BLOCK
#define MISC_BLOCK 256
#endif

// Everything a (re)generated path starts from (src/wf_reset.cl:30-60 == src/wf_raygen.cl:77-96)
__device__ __forceinline__ void init_path_state(const State &st, uint32_t gid, float shadowLen)
{
    wr4(st.at(S_LBSDF, gid), make_float4(0.0f, 0.0f, 0.0f, 0.0f));       // lastBsdf, lastPdfImplicit
    wr4(st.at(S_LEMIT, gid), make_float4(0.0f, 0.0f, 0.0f, 0.0f));       // lastEmission, lastCosTh
    wr4(st.at(S_HITP, gid), make_float4(0.0f, 0.0f, 0.0f, FLX_FLT_MAX)); // EMPTY_HIT
    wr4(st.at(S_HITN, gid), make_float4(0.0f, 0.0f, 0.0f, 0.0f));       // N, areaLightHit = backfaceHit = 0
    wr4(st.at(S_HITUV, gid), make_float4(0.0f, 0.0f, __int_as_float(-1), __int_as_float(-1)));
    st.pickProb[gid] = 1.0f;
    st.blocked[gid] = 1u;
    st.firstDiffuse[gid] = 0u;
    // records with members reset() leaves alone are read-modify-written
    float4 sho = rd4(st.at(S_SHO, gid)); sho.w = shadowLen; wr4(st.at(S_SHO, gid), sho);
    float4 shd = rd4(st.at(S_SHD, gid)); shd.w = 0.0f; wr4(st.at(S_SHD, gid), shd);            // lastPdfDirect
    float4 lt = rd4(st.at(S_LT, gid)); lt.w = __uint_as_float(1u); wr4(st.at(S_LT, gid), lt);  // lastSpecular
}

__global__ __launch_bounds__(MISC_BLOCK) void k_reset(State st, Queues qs, Frame fr, flx_render_params p, uint32_t n)
{
    const uint32_t gid = blockIdx.x * MISC_BLOCK + threadIdx.x;
    if (gid >= n) return;
    if (gid < fr.localPixels) {
        reinterpret_cast<float4 *>(fr.pixels)[gid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (fr.aovNormal) {                                                                      // src/wf_reset.cl:22-24
            reinterpret_cast<float4 *>(fr.aovNormal)[gid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            reinterpret_cast<float4 *>(fr.aovAlbedo)[gid] = make_float4(0.1f, 0.1f, 0.1f, 0.0f);   // default for direct emission
        }
    }
    if (gid >= st.numTasks) return;
    init_path_state(st, gid, 2.0f * p.worldRadius);
    wr4(st.at(S_EI, gid), make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u)));     // Ei, pixelIndex
    wr4(st.at(S_THR, gid), mk4u(mk3(1.0f), gid));                                   // T, seed = gid
    float4 o = rd4(st.at(S_ORIG, gid)); o.w = 1.0f; wr4(st.at(S_ORIG, gid), o);          // lastPdfW
    float4 d = rd4(st.at(S_DIR, gid)); d.w = __uint_as_float(0u); wr4(st.at(S_DIR, gid), d);   // pathLen
    qs.q[FLX_Q_RAYGEN][gid] = gid;
    if (gid == 0) qs.counters[FLX_Q_RAYGEN] = st.numTasks;
}

// appendExt 0: the extension-queue entries of the regenerated paths were written already -- by the fused scatter, merged with the continuing
// paths into ONE list in path-id order (logic.hip: k_queue_scatter, ext_order 2) -- and only the paths are regenerated here
// prepared 1 (PREPARED REGENERATION, logic.hip): the fused RAW pass of this chain has stored everything of the regenerated paths that does not depend on the
// pixel -- origin, throughput + the seed genRays leaves, shadowRayBlocked, lastLightPickProb -- and left {jitter x, jitter y, seed after the jitter} in the
// direction record: only the direction and the pixel word are stored here (two scattered 16-byte stores per path instead of four + three scalars; this kernel
// is bound by them).  The lens origin is recomputed from the seed (same function, same bits) rather than gathered.
__global__ __launch_bounds__(MISC_BLOCK) void k_raygen(State st, Queues qs, Frame fr, flx_render_params p, uint32_t appendExt, uint32_t prepared)
{
    const uint32_t qlen = qs.counters[FLX_Q_RAYGEN];
    // capped grid striding over the queue (its length is only known here; see MAT_GRID in material.hip)
    for (uint32_t gd = blockIdx.x * MISC_BLOCK + threadIdx.x; gd < qlen; gd += gridDim.x * MISC_BLOCK) {
        const uint32_t gid = qs.q[FLX_Q_RAYGEN][gd];
        // pixel cursor over the rank's local pixels; local p <-> global p*nranks + rank
        // (1 rank: the reference's (cur + gid_direct) % numPixels, src/wf_raygen.cl:25)
        const uint32_t localIdx = (*fr.currPixelIdx + gd) % fr.localPixels;
        if (prepared) {
            const float4 j4 = rd4t(st.at(S_DIR, gid));
            uint32_t sd = __float_as_uint(j4.z);
            const f3 lens = camera_lens_origin(p, &sd);
            wr4(st.at(S_DIR, gid), mk4u(camera_direction(fr, p, localIdx, j4.x, j4.y, lens), FLX_FRESH | 0u));
            wr4(st.at(S_EI, gid), mk4u(mk3(0.0f), FLX_FRESH | localIdx));
            if (st.firstDiffuse[gid]) st.firstDiffuse[gid] = 0u;     // (a read instead of a third scattered store: the flag is set only while the denoiser features accumulate)
            if (appendExt) qs.q[FLX_Q_EXTENSION][ext_len(qs) + gd] = gid;
            continue;
        }
        uint32_t seed = __float_as_uint(rd4t(st.at(S_THR, gid)).w);
        f3 rayOrig, rayDirection;
        camera_ray(fr, p, localIdx, &seed, &rayOrig, &rayDirection);

        // (temporal instead of non-temporal stores here: within the run-to-run spread, profiles/r03_raygen_temporal_ab.txt)
        wr4(st.at(S_ORIG, gid), mk4(rayOrig, 1.0f));                  // lastPdfW = 1
        wr4(st.at(S_DIR, gid), mk4u(rayDirection, FLX_FRESH | 0u));   // pathLen = 0; the rest of init_path_state's resets are
        wr4(st.at(S_EI, gid), mk4u(mk3(0.0f), FLX_FRESH | localIdx)); // implied by the two flags (flx_device.h), not stored
        wr4(st.at(S_THR, gid), mk4u(mk3(1.0f), seed));
        st.pickProb[gid] = 1.0f;                                       // read by the MIS weights before any NEE may have written it
        st.blocked[gid] = 1u;
        st.firstDiffuse[gid] = 0u;
        if (appendExt) qs.q[FLX_Q_EXTENSION][ext_len(qs) + gd] = gid;              // extBase + index (see flx_device.h)
    }
}

__global__ void k_bump_extension(uint32_t *counters, uint32_t srcMask)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t add = 0;
        for (int q = 0; q < FLX_NUM_QUEUES; q++) if (srcMask & (1u << q)) add += counters[q];
        counters[FLX_Q_EXTENSION] += add;
    }
}
// multi-GPU gather, root side: rank r's tile holds its local pixels p = 0, 1, ... = global pixels p * nranks + r (flx_set_partition);
// stage = nranks tiles of maxlp float4 each
__global__ __launch_bounds__(MISC_BLOCK) void k_deinterleave(const float4 *stage, float4 *full, uint32_t npix, uint32_t nranks, uint32_t maxlp)
{
    const uint32_t g = blockIdx.x * MISC_BLOCK + threadIdx.x;
    if (g >= npix) return;
    full[g] = stage[(size_t)(g % nranks) * maxlp + g / nranks];
}
void launch_deinterleave(hipStream_t s, const float *stage, float *full, uint32_t npix, uint32_t nranks, uint32_t maxlp)
{
    hipLaunchKernelGGL(k_deinterleave, dim3((npix + MISC_BLOCK - 1) / MISC_BLOCK), dim3(MISC_BLOCK), 0, s,
                       reinterpret_cast<const float4 *>(stage), reinterpret_cast<float4 *>(full), npix, nranks, maxlp);
}

void launch_bump_extension(hipStream_t s, uint32_t *counters, uint32_t srcMask)
{
    hipLaunchKernelGGL(k_bump_extension, dim3(1), dim3(64), 0, s, counters, srcMask);
}

__device__ __forceinline__ f3 uc2_func(f3 x)
{
    const float A = 0.22f, B = 0.30f, C = 0.10f, D = 0.20f, E = 0.01f, Fq = 0.30f;
    return ((x * (A * x + mk3(C * B)) + mk3(D * E)) / (x * (A * x + mk3(B)) + mk3(D * Fq))) - mk3(E / Fq);
}

__global__ __launch_bounds__(MISC_BLOCK) void k_postprocess(Frame fr, flx_render_params p)
{
    const uint32_t gid = blockIdx.x * MISC_BLOCK + threadIdx.x;
    if (gid >= fr.localPixels) return;
    const float4 in = reinterpret_cast<const float4 *>(fr.pixels)[gid];
    f3 col = ld3(in); float w = in.w;
    if (w > 0.0f) { col = col / w; w = w / w; }
    col = col * p.exposure;
    if (p.tmOperator == 1u) col = col / (mk3(1.0f) + col);
    if (p.tmOperator == 2u) col = uc2_func(2.0f * col) / uc2_func(mk3(11.2f));
    col = pow3(col, 1.0f / 2.2f);
    reinterpret_cast<float4 *>(fr.preview)[gid] = mk4(col, w);
    if (fr.aovNormal) {                                                                          // src/mk_postprocess.cl:49-54
        const float4 n = reinterpret_cast<const float4 *>(fr.aovNormal)[gid], a = reinterpret_cast<const float4 *>(fr.aovAlbedo)[gid];
        reinterpret_cast<float4 *>(fr.aovNormalOut)[gid] = n.w > 1.0f ? make_float4(n.x / n.w, n.y / n.w, n.z / n.w, n.w / n.w) : n;
        reinterpret_cast<float4 *>(fr.aovAlbedoOut)[gid] = a.w > 1.0f ? make_float4(a.x / a.w, a.y / a.w, a.z / a.w, a.w / a.w) : a;
    }
}

// ---- test hooks: internal packed layout <-> reference 64-column SoA (src/geom.h:199-236)
__global__ __launch_bounds__(MISC_BLOCK) void k_state_export(State st, float *out, float shadowLenReset)
{
    const uint32_t gid = blockIdx.x * MISC_BLOCK + threadIdx.x;
    const uint32_t N = st.numTasks;
    if (gid >= N) return;
    auto W = [&](int col, float v) { out[(size_t)col * N + gid] = v; };
    auto W3 = [&](int col, float4 v) { W(col, v.x); W(col + 1, v.y); W(col + 2, v.z); W(col + 3, 0.0f); };
    float4 r;
    // the flags of a regenerated path (flx_device.h): members they cover are exported with genRays' reset values (src/wf_raygen.cl:77-96)
    r = rd4(st.at(S_ORIG, gid)); W3(FLX_COL_ORIG, r); W(FLX_COL_LAST_PDF_W, r.w);
    r = rd4(st.at(S_DIR, gid)); W3(FLX_COL_DIR, r);
    const uint32_t lenBits = __float_as_uint(r.w);
    const bool noMaterial = (lenBits & FLX_FRESH) != 0u, notExtended = noMaterial && (lenBits & ~FLX_FRESH) == 0u;
    W(FLX_COL_PATH_LEN, __uint_as_float(lenBits & ~FLX_FRESH));
    r = rd4(st.at(S_EI, gid)); W3(FLX_COL_EI, r);
    const bool noNee = (__float_as_uint(r.w) & FLX_FRESH) != 0u;
    W(FLX_COL_PIXEL_INDEX, __uint_as_float(__float_as_uint(r.w) & ~FLX_FRESH));
    r = rd4(st.at(S_SHO, gid)); W3(FLX_COL_SHADOW_ORIG, r); W(FLX_COL_SHADOW_LEN, noNee ? shadowLenReset : r.w);
    r = rd4(st.at(S_SHD, gid)); W3(FLX_COL_SHADOW_DIR, r); W(FLX_COL_LAST_PDF_DIRECT, noNee ? 0.0f : r.w);
    r = rd4(st.at(S_THR, gid)); W3(FLX_COL_T, r); W(FLX_COL_SEED, r.w);
    r = rd4(st.at(S_LBSDF, gid)); if (noMaterial) r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    W3(FLX_COL_LAST_BSDF, r); W(FLX_COL_LAST_PDF_IMPLICIT, r.w);
    r = rd4(st.at(S_LEMIT, gid)); if (noNee) r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    W3(FLX_COL_LAST_EMISSION, r); W(FLX_COL_LAST_COS_TH, r.w);
    r = rd4(st.at(S_LT, gid)); W3(FLX_COL_LAST_T, r); W(FLX_COL_LAST_SPECULAR, noMaterial ? __uint_as_float(1u) : r.w);
    r = rd4(st.at(S_HITP, gid)); if (notExtended) r = make_float4(0.0f, 0.0f, 0.0f, FLX_FLT_MAX);
    W3(FLX_COL_P, r); W(FLX_COL_HIT_T, r.w);
    r = rd4(st.at(S_HITN, gid)); if (notExtended) r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    W3(FLX_COL_N, r);
    // backfaceHit: logic's, reset by genRays; a regenerated path keeps it clear until logic shades its first hit (commit_hit drops the
    // stale bit of the slot's previous path when it extends a fresh ray)
    const uint32_t fl = __float_as_uint(r.w);
    W(FLX_COL_AREA_LIGHT_HIT, __uint_as_float(fl & 1u)); W(FLX_COL_BACKFACE, __uint_as_float((fl >> 1) & 1u));
    r = rd4(st.at(S_HITUV, gid)); if (notExtended) r = make_float4(0.0f, 0.0f, __int_as_float(-1), __int_as_float(-1));
    W(FLX_COL_UV, r.x); W(FLX_COL_UV + 1, r.y); W(FLX_COL_HIT_I, r.z); W(FLX_COL_MAT_ID, r.w);
    W(FLX_COL_PHASE, __uint_as_float(st.phase[gid]));
    W(FLX_COL_SHADOW_BLOCKED, __uint_as_float(st.blocked[gid]));
    W(FLX_COL_LAST_PICK_PROB, st.pickProb[gid]);
    W(FLX_COL_FIRST_DIFFUSE, __uint_as_float(st.firstDiffuse[gid]));
}

__global__ __launch_bounds__(MISC_BLOCK) void k_state_import(State st, const float *in)
{
    const uint32_t gid = blockIdx.x * MISC_BLOCK + threadIdx.x;
    const uint32_t N = st.numTasks;
    if (gid >= N) return;
    auto R = [&](int col) { return in[(size_t)col * N + gid]; };
    auto R4 = [&](int col, int wcol) { return make_float4(R(col), R(col + 1), R(col + 2), R(wcol)); };
    wr4(st.at(S_ORIG, gid), R4(FLX_COL_ORIG, FLX_COL_LAST_PDF_W));
    wr4(st.at(S_DIR, gid), R4(FLX_COL_DIR, FLX_COL_PATH_LEN));
    wr4(st.at(S_SHO, gid), R4(FLX_COL_SHADOW_ORIG, FLX_COL_SHADOW_LEN));
    wr4(st.at(S_SHD, gid), R4(FLX_COL_SHADOW_DIR, FLX_COL_LAST_PDF_DIRECT));
    wr4(st.at(S_THR, gid), R4(FLX_COL_T, FLX_COL_SEED));
    wr4(st.at(S_EI, gid), R4(FLX_COL_EI, FLX_COL_PIXEL_INDEX));
    wr4(st.at(S_LBSDF, gid), R4(FLX_COL_LAST_BSDF, FLX_COL_LAST_PDF_IMPLICIT));
    wr4(st.at(S_LEMIT, gid), R4(FLX_COL_LAST_EMISSION, FLX_COL_LAST_COS_TH));
    wr4(st.at(S_LT, gid), R4(FLX_COL_LAST_T, FLX_COL_LAST_SPECULAR));
    wr4(st.at(S_HITP, gid), R4(FLX_COL_P, FLX_COL_HIT_T));
    const uint32_t fl = (__float_as_uint(R(FLX_COL_AREA_LIGHT_HIT)) ? 1u : 0u) | (__float_as_uint(R(FLX_COL_BACKFACE)) ? 2u : 0u);
    wr4(st.at(S_HITN, gid), make_float4(R(FLX_COL_N), R(FLX_COL_N + 1), R(FLX_COL_N + 2), __uint_as_float(fl)));
    wr4(st.at(S_HITUV, gid), make_float4(R(FLX_COL_UV), R(FLX_COL_UV + 1), R(FLX_COL_HIT_I), R(FLX_COL_MAT_ID)));
    st.blocked[gid] = __float_as_uint(R(FLX_COL_SHADOW_BLOCKED));
    st.pickProb[gid] = R(FLX_COL_LAST_PICK_PROB);
    st.firstDiffuse[gid] = __float_as_uint(R(FLX_COL_FIRST_DIFFUSE));
    st.phase[gid] = __float_as_uint(R(FLX_COL_PHASE));
}

__global__ void k_end_iteration(uint32_t *counters, unsigned long long *totals, uint32_t *cursor, uint32_t localPixels, uint32_t extPend, uint32_t *blockCursors)
{
    const uint32_t i = threadIdx.x;
    if (i < 8u) {
        uint32_t v = counters[i];
        if (i == FLX_Q_EXTENSION) for (int q = 0; q < FLX_NUM_QUEUES; q++) if (extPend & (1u << q)) v += counters[q];     // lazy extension counter
        totals[i] += v;
        if (i == FLX_Q_RAYGEN) *cursor = (uint32_t)(((unsigned long long)*cursor + v) % localPixels);
        counters[i] = 0u;
    }
    if (i >= 8u && i < 8u + FLX_NUM_BLOCK_CURSORS) blockCursors[(i - 8u) * FLX_CURSOR_STRIDE] = 0u;     // block cursors of the persistent traversal kernels (flx_device.h)
}
void launch_end_iteration(hipStream_t s, uint32_t *counters, unsigned long long *totals, uint32_t *cursor, uint32_t localPixels, uint32_t extPend, uint32_t *blockCursors)
{
    hipLaunchKernelGGL(k_end_iteration, dim3(1), dim3(64), 0, s, counters, totals, cursor, localPixels, extPend, blockCursors);
}

void launch_reset(hipStream_t s, const State &st, const Queues &qs, const Frame &fr, const flx_render_params &p)
{
    uint32_t n = st.numTasks > fr.localPixels ? st.numTasks : fr.localPixels;      // src/clcontext.cpp:767
    hipLaunchKernelGGL(k_reset, dim3((n + MISC_BLOCK - 1) / MISC_BLOCK), dim3(MISC_BLOCK), 0, s, st, qs, fr, p, n);
}
void launch_raygen(hipStream_t s, const State &st, const Queues &qs, const Frame &fr, const flx_render_params &p, int appendExt, int prepared)
{
    uint32_t blocks = (st.numTasks + MISC_BLOCK - 1) / MISC_BLOCK;
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(k_raygen, dim3(blocks), dim3(MISC_BLOCK), 0, s, st, qs, fr, p, (uint32_t)appendExt, (uint32_t)prepared);
}
// the per-texel table of next-event estimation (flx_device.h: Scene::neeRec)
__global__ __launch_bounds__(MISC_BLOCK) void k_env_nee_table(Scene sc, float4 *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * MISC_BLOCK + threadIdx.x;
    if (i >= n) return;
    const EnvSample e = env_sample_compute(sc, (int)i);
    out[2 * (size_t)i] = mk4(e.L, e.pdfW);
    out[2 * (size_t)i + 1] = mk4(e.Li, 0.0f);
}
void launch_env_nee_table(hipStream_t s, const Scene &sc, float4 *out, uint32_t n)
{
    hipLaunchKernelGGL(k_env_nee_table, dim3((n + MISC_BLOCK - 1) / MISC_BLOCK), dim3(MISC_BLOCK), 0, s, sc, out, n);
}
void launch_postprocess(hipStream_t s, const Frame &fr, const flx_render_params &p)
{
    hipLaunchKernelGGL(k_postprocess, dim3((fr.localPixels + MISC_BLOCK - 1) / MISC_BLOCK), dim3(MISC_BLOCK), 0, s, fr, p);
}
// test hook: the arithmetic contract (include/flx_math.h) evaluated on the device, results as bit patterns -- the oracle's orc_math_array
// is the other half (tests/test_gpu_parity.py::test_arithmetic_contract_device_vs_oracle)
__global__ __launch_bounds__(MISC_BLOCK) void k_math_probe(int fn, const float *a, const float *b, uint32_t n, uint32_t *out)
{
    const uint32_t i = blockIdx.x * MISC_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    float r = 0.0f;
    switch (fn) {
    case 0: r = sinf_(x); break; case 1: r = cosf_(x); break; case 2: r = tanf_(x); break; case 3: r = atan2f_(x, y); break;
    case 4: r = acosf_(x); break; case 5: r = powf_(x, y); break; case 6: r = logf_(x); break; case 7: r = expf_(x); break;
    case 8: r = asinf_(x); break; case 9: r = atanf_(x); break; case 10: r = fminf_(x, y); break; case 11: r = fmaxf_(x, y); break;
    case 12: r = x / y; break; case 13: r = sqrtf(x); break; case 14: r = x * y; break; case 15: r = x + y; break;
    }
    out[i] = __float_as_uint(r);
}
void launch_math_probe(hipStream_t s, int fn, const float *a, const float *b, uint32_t n, uint32_t *out)
{
    hipLaunchKernelGGL(k_math_probe, dim3((n + MISC_BLOCK - 1) / MISC_BLOCK), dim3(MISC_BLOCK), 0, s, fn, a, b, n, out);
}
void launch_state_export(hipStream_t s, const State &st, float *out, float shadowLenReset)
{
    hipLaunchKernelGGL(k_state_export, dim3((st.numTasks + MISC_BLOCK - 1) / MISC_BLOCK), dim3(MISC_BLOCK), 0, s, st, out, shadowLenReset);
}
void launch_state_import(hipStream_t s, const State &st, const float *in)
{
    hipLaunchKernelGGL(k_state_import, dim3((st.numTasks + MISC_BLOCK - 1) / MISC_BLOCK), dim3(MISC_BLOCK), 0, s, st, in);
}

} // namespace flxd
