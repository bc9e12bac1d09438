// material.hip -- per-material BSDF evaluation / sampling kernels of the wavefront loop.
//
// Replaces reference kernels wavefrontDiffuse / Glossy / GGXReflection / GGXRefraction / Delta /
// AllMaterials (src/wf_mat_diffuse.cl:7-67 and twins) with bxdfSample/Eval/Pdf
// (src/bxdf_partial.cl:19-153) and the BSDFs of src/diffuse.cl, glossy.cl, ggx.cl,
// ideal_reflection.cl, ideal_dielectric.cl, fresnel.cl.
//
// One kernel template, specialised at compile time by the set of BSDF types its queue can contain
// (the reference gets the same effect from -DBXDF_USE_* at OpenCL build time).  For every queued
// path: evaluate f and pdf toward the stored light direction (consumed by `logic` next iteration),
// sample the continuation direction, update throughput, write the new ray and append the path to
// the extension queue at a COMPUTED slot (extension base + lengths of the source queues appended earlier
// + own index; no atomic -- see flx_device.h).
#include "flx_bsdf.h"

namespace flxd {

#ifndef MAT_BLOCK
#define MAT_BLOCK 64            // one wave per block: interleaves best with the one-wave blocks of the concurrent shadow traversal (+1 %)
#endif

// earlierMask: the material queues appended to the extension queue before this one; FLX_NO_APPEND: the extension-queue entries of
// this iteration's continuing paths exist already (written by the fused logic pass's scatter, logic.hip)
#define FLX_NO_APPEND 0x80000000u
template <int USE>
__device__ __forceinline__ void material_body(const State &st, const Queues &qs, const Scene &sc, int queueId, uint32_t idx, uint32_t earlierMask)
{
    const uint32_t qlen = qs.counters[queueId];
    const bool active = idx < qlen;
    uint32_t gid = 0;
    if (active) {
        gid = qs.q[queueId][idx];
        const float4 thr = rd4t(st.at(S_THR, gid));                 // (temporal loads: `logic` has just touched these lines)
        const float4 hp = rd4t(st.at(S_HITP, gid));
        const float4 hn = rd4t(st.at(S_HITN, gid));
        const float4 huv = rd4t(st.at(S_HITUV, gid));
        const float4 d4 = rd4t(st.at(S_DIR, gid));
        const float4 sd = rd4t(st.at(S_SHD, gid));
        uint32_t seed = __float_as_uint(thr.w);
        SurfHit h; h.P = ld3(hp); h.N = ld3(hn); h.uv = mk2(huv.x, huv.y);
        const bool backface = (__float_as_uint(hn.w) & 2u) != 0u;
        const f3 oldT = ld3(thr);
        const MatStep o = material_step<USE>(sc, h, sc.materials[__float_as_int(huv.w)], backface, ld3(d4), ld3(sd), oldT, &seed);

        wr4(st.at(S_LBSDF, gid), mk4(o.bsdfNEE, o.bsdfPdfW));
        wr4(st.at(S_LT, gid), mk4u(oldT, o.singular));
        wr4(st.at(S_THR, gid), mk4u(o.newT, seed));
        wr4(st.at(S_ORIG, gid), mk4(o.orig, o.pdfW));
        wr4(st.at(S_DIR, gid), mk4u(o.newDir, __float_as_uint(d4.w) & ~FLX_FRESH));   // pathLen; "no material kernel since regeneration" ends here
    }
    if (active && !(earlierMask & FLX_NO_APPEND)) {
        // slot = extBase + lengths of the material queues appended before this one + own index (flx_device.h)
        uint32_t base = ext_len(qs);
        for (int q = FLX_Q_DIFFUSE; q < FLX_NUM_QUEUES; q++) if (earlierMask & (1u << q)) base += qs.counters[q];
        qs.q[FLX_Q_EXTENSION][base + idx] = gid;
    }
}

// Grids are capped (MAT_GRID blocks) and stride over the queue: a queue's length is only known on the device, and a grid sized for
// the worst case (numTasks / 64 blocks) that mostly exits at once still has to be dispatched block by block -- behind the waves of the
// concurrent shadow traversal that took 0.3 ms per iteration for 0.05 ms of work.
#define MAT_GRID 4096u
template <int USE>
__global__ __launch_bounds__(MAT_BLOCK) void k_material(State st, Queues qs, Scene sc, int queueId, uint32_t earlierMask)
{
    const uint32_t nb = (qs.counters[queueId] + MAT_BLOCK - 1) / MAT_BLOCK;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x)
        material_body<USE>(st, qs, sc, queueId, b * MAT_BLOCK + threadIdx.x, earlierMask);
}

// The four small queues (glossy, GGX reflection, GGX refraction, delta) in ONE launch: each block serves one queue
// (block ranges follow the queue lengths), so waves stay BSDF-uniform like in the per-queue kernels, but three
// near-empty 4096-block launches per iteration disappear.
// doneMask: queues the fused logic pass has already served (logic.hip); they only count as "appended earlier".
__global__ __launch_bounds__(MAT_BLOCK) void k_material_rest(State st, Queues qs, Scene sc, uint32_t doneMask)
{
    const uint32_t noAppend = doneMask & FLX_NO_APPEND;
    uint32_t nbq[4], total = 0;
    for (int q = FLX_Q_GLOSSY; q <= FLX_Q_DELTA; q++) {
        nbq[q - FLX_Q_GLOSSY] = (doneMask & (1u << q)) ? 0u : (qs.counters[q] + MAT_BLOCK - 1) / MAT_BLOCK;
        total += nbq[q - FLX_Q_GLOSSY];
    }
    for (uint32_t vb = blockIdx.x; vb < total; vb += gridDim.x) {
        uint32_t b = vb;
        uint32_t earlier = (1u << FLX_Q_DIFFUSE) | noAppend;
        for (int q = FLX_Q_GLOSSY; q <= FLX_Q_DELTA; q++) {
            const uint32_t nb = nbq[q - FLX_Q_GLOSSY];
            if (b < nb) { material_body<USE_GLOSSY | USE_GGX_REFL | USE_GGX_REFR | USE_DELTA>(st, qs, sc, q, b * MAT_BLOCK + threadIdx.x, earlier); break; }
            b -= nb;
            earlier |= 1u << q;
        }
    }
}

void launch_bump_extension(hipStream_t s, uint32_t *counters, uint32_t srcMask);

static void launch_one(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, int queueId, int use, uint32_t earlierMask)
{
    uint32_t blocks = (st.numTasks + MAT_BLOCK - 1) / MAT_BLOCK;
    if (blocks > MAT_GRID) blocks = MAT_GRID;
    switch (use) {
    case USE_DIFFUSE: hipLaunchKernelGGL(k_material<USE_DIFFUSE>, dim3(blocks), dim3(MAT_BLOCK), 0, s, st, qs, sc, queueId, earlierMask); break;
    case USE_GLOSSY: hipLaunchKernelGGL(k_material<USE_GLOSSY>, dim3(blocks), dim3(MAT_BLOCK), 0, s, st, qs, sc, queueId, earlierMask); break;
    case USE_GGX_REFL: hipLaunchKernelGGL(k_material<USE_GGX_REFL>, dim3(blocks), dim3(MAT_BLOCK), 0, s, st, qs, sc, queueId, earlierMask); break;
    case USE_GGX_REFR: hipLaunchKernelGGL(k_material<USE_GGX_REFR>, dim3(blocks), dim3(MAT_BLOCK), 0, s, st, qs, sc, queueId, earlierMask); break;
    case USE_DELTA: hipLaunchKernelGGL(k_material<USE_DELTA>, dim3(blocks), dim3(MAT_BLOCK), 0, s, st, qs, sc, queueId, earlierMask); break;
    default: hipLaunchKernelGGL(k_material<USE_ALL>, dim3(blocks), dim3(MAT_BLOCK), 0, s, st, qs, sc, queueId, earlierMask); break;
    }
}

// reference: CLContext::enqueueWfMaterialKernels (src/clcontext.cpp:796-813)
void launch_materials(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, uint32_t separateQueues)
{
    if (separateQueues) {
        const uint32_t D = 1u << FLX_Q_DIFFUSE, G = 1u << FLX_Q_GLOSSY, RL = 1u << FLX_Q_GGX_REFL, RR = 1u << FLX_Q_GGX_REFR, DL = 1u << FLX_Q_DELTA;
        launch_one(s, st, qs, sc, FLX_Q_DIFFUSE, USE_DIFFUSE, 0u);
        {   // glossy + ggxRefl + ggxRefr + delta (at most numTasks paths in total -> blocks + 4 partial blocks)
            uint32_t blocks = (st.numTasks + MAT_BLOCK - 1) / MAT_BLOCK + 4;
            if (blocks > MAT_GRID) blocks = MAT_GRID;
            hipLaunchKernelGGL(k_material_rest, dim3(blocks), dim3(MAT_BLOCK), 0, s, st, qs, sc, 0u);
        }
        (void)G; (void)RL; (void)RR; (void)DL; (void)D;            // the host records the appended queues (flx_wf_materials): lazy bump
    } else {
        launch_one(s, st, qs, sc, FLX_Q_DIFFUSE, USE_ALL, 0u);
    }
}

// what is left of flx_wf_materials after the fused logic pass (separate queues): the queues in doneMask are served already, and
// the extension-queue entries of every continuing path are written (so nothing appends here)
void launch_materials_after_fused(hipStream_t s, const State &st, const Queues &qs, const Scene &sc, uint32_t doneMask, int extWritten)
{
    const uint32_t FLX_NO_APPEND_ = extWritten ? FLX_NO_APPEND : 0u;
    const uint32_t all = (1u << FLX_Q_DIFFUSE) | (1u << FLX_Q_GLOSSY) | (1u << FLX_Q_GGX_REFL) | (1u << FLX_Q_GGX_REFR) | (1u << FLX_Q_DELTA);
    if ((doneMask & all) == all) return;
    if (!(doneMask & (1u << FLX_Q_DIFFUSE))) launch_one(s, st, qs, sc, FLX_Q_DIFFUSE, USE_DIFFUSE, FLX_NO_APPEND_);
    if ((doneMask & (all & ~(1u << FLX_Q_DIFFUSE))) != (all & ~(1u << FLX_Q_DIFFUSE))) {
        uint32_t blocks = (st.numTasks + MAT_BLOCK - 1) / MAT_BLOCK + 4;
        if (blocks > MAT_GRID) blocks = MAT_GRID;
        hipLaunchKernelGGL(k_material_rest, dim3(blocks), dim3(MAT_BLOCK), 0, s, st, qs, sc, doneMask | FLX_NO_APPEND_);
    }
}

} // namespace flxd
