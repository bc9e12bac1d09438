// tests/wide_analysis.cpp -- CPU-side analysis of the 4-wide traversal tree (TEST / EXPERIMENT INFRASTRUCTURE, built by tests/conftest.py into
// tests/_build/libwide_analysis.so; the product never loads it).
//   fh_wide_visits[_ex]  the traversal of flx_trace4.h emulated on the host with the device's arithmetic (WRay::setup, wide_node_visit,
//                        wide_leaf_visit: same float operations in the same order, fmaf for the plane evaluations) over caller-supplied
//                        rays, on the tree build_wide() (csrc/flx_wide.h -- the product's own builder) makes: hits per ray + visit counts.
//                        tests/test_wide_emulation.py compares its hits with the oracle's binary traversal (src/bvh.cl:234-373).
//   fh_wide_optimise     the archived topology optimiser (scripts/experiments/flx_wide_opt.h) for scripts/exp_tree_opt.py.
#include <stdint.h>
#include <stdexcept>
#include <string>
#include <vector>
#include <cmath>
#include <cstring>
#include <algorithm>
#include "../fluctus_amd/csrc/flx_wide.h"
#include "../scripts/experiments/flx_wide_opt.h"
#include "../include/flx_math.h"

static thread_local std::string g_err;
#define FH_TRY try {
#define FH_CATCH } catch (const std::exception &e) { g_err = e.what(); return 1; } return 0;
extern "C" {
const char *fh_analysis_last_error() { return g_err.c_str(); }

// ---- CPU-side analysis of the traversal tree (tests/test_host.py, scripts/exp_tree_opt.py): the inner topology re-optimised by
// flx_wide_opt.h, and the visit counts of the 4-wide traversal emulated on the host with the device's arithmetic (flx_trace4.h:
// WRay::setup, wide_node_visit, wide_leaf_visit) on caller-supplied rays.  Not on the product path.
int fh_wide_optimise(const void *nodesv, uint64_t nnodes, int passes, void *outNodes /* nnodes x 48 B */, double *stats8)
{
    FH_TRY
    std::vector<flx_node> res; flxw::OptStats st; const char *err = nullptr;
    if (!flxw::optimise_topology((const flx_node *)nodesv, nnodes, passes, res, &st, &err)) throw std::runtime_error(err ? err : "optimise_topology failed");
    memcpy(outNodes, res.data(), nnodes * sizeof(flx_node));
    if (stats8) { stats8[0] = st.costBefore; stats8[1] = st.costAfter; stats8[2] = (double)st.moved; stats8[3] = (double)st.searched; stats8[4] = (double)st.searchSteps;
                  stats8[5] = st.depthBefore; stats8[6] = st.depthAfter; stats8[7] = st.passes; }
    FH_CATCH
}

namespace {
using namespace flx;
struct SimRay { f3 orig, dir, dinv; float dw[3]; bool neg[3]; };
static inline bool sim_slab(const float *bmin, const float *bmax, const SimRay &r, float tMaxPrev)
{
    f3 tmp = (mk3(bmin[0], bmin[1], bmin[2]) - r.orig) * r.dinv;
    f3 tmaxv = (mk3(bmax[0], bmax[1], bmax[2]) - r.orig) * r.dinv;
    f3 tminv = min3(tmp, tmaxv);
    tmaxv = max3(tmp, tmaxv);
    const float tmin = fmaxf_(fmaxf_(tminv.x, tminv.y), tminv.z), tmax = fminf_(fminf_(tmaxv.x, tmaxv.y), tmaxv.z);
    if (tmax < 0.0f) return false;
    if (tmin > tmax) return false;
    return tmin < tMaxPrev;
}
static inline bool sim_mt(f3 orig, f3 dir, f3 p0, f3 p1, f3 p2, float *tret)
{
    f3 s1 = p1 - p0, s2 = p2 - p0, pvec = cross(dir, s2);
    const float det = dot(s1, pvec);
    if (absf(det) < 1e-12f) return false;
    const float iDet = 1.0f / det;
    f3 tvec = orig - p0;
    const float u = dot(tvec, pvec) * iDet;
    if (u < 0.0f || u > 1.0f) return false;
    f3 qvec = cross(tvec, s1);
    const float v = dot(dir, qvec) * iDet;
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = dot(s2, qvec) * iDet;
    if (t < 0.0f) return false;
    *tret = t;
    return true;
}
}

// rays: n x 8 floats {orig.xyz, tmax, dir.xyz, unused}.  mode 0 closest hit (nearest child first), 1 any hit / last hit slot first, 2 any hit / farthest
// first.  out8 = {node visits, leaf visits, leaf boxes passed, triangle tests, rays with a hit, deepest stack, wide nodes, sum of hit triangle
// indices + 1 (a checksum to compare two trees over the same leaves: closest hit must agree up to ties)}.
int fh_wide_visits_ex(const void *nodesv, uint64_t nnodes, const void *trisv, uint64_t ntris, const uint32_t *indices, uint64_t nidx,
                      const float *rays, uint64_t nrays, int mode, double *out8, int32_t *hitTri, uint32_t *nodeVisitsPerRay);
int fh_wide_visits(const void *nodesv, uint64_t nnodes, const void *trisv, uint64_t ntris, const uint32_t *indices, uint64_t nidx,
                   const float *rays, uint64_t nrays, int mode, double *out8)
{
    return fh_wide_visits_ex(nodesv, nnodes, trisv, ntris, indices, nidx, rays, nrays, mode, out8, nullptr, nullptr);
}
// + per ray: the triangle found (closest hit: the winner; any hit: the first occluder met, in the device's visit order; -1 none) and the
// number of wide-node visits
int fh_wide_visits_ex(const void *nodesv, uint64_t nnodes, const void *trisv, uint64_t ntris, const uint32_t *indices, uint64_t nidx,
                      const float *rays, uint64_t nrays, int mode, double *out8, int32_t *hitTri, uint32_t *nodeVisitsPerRay)
{
    FH_TRY
    flxw::WideTree w; const char *err = nullptr;
    if (!flxw::build_wide((const flx_node *)nodesv, nnodes, (const flx_triangle *)trisv, ntris, indices, nidx, w, &err)) throw std::runtime_error(err ? err : "build_wide failed");
    flxw::reorder_slots(w, mode >> 4);                 // bits 4.. of mode: slot order key (flx_wide.h: reorder_slots)
    mode &= 15;
    const flx_node &r0 = ((const flx_node *)nodesv)[0];
    float m = 0.0f; for (float v : {r0.bmin.x, r0.bmin.y, r0.bmin.z, r0.bmax.x, r0.bmax.y, r0.bmax.z}) m = std::fabs(v) > m ? std::fabs(v) : m;
    const float clampNear = m < 67108864.0f ? 1.2676506e30f : 1.8446744e19f;
    double nv = 0, lv = 0, lp = 0, tt = 0, hits = 0, chk = 0; int deepest = 0;
    #pragma omp parallel for schedule(dynamic, 256) reduction(+ : nv, lv, lp, tt, hits, chk) reduction(max : deepest)
    for (int64_t ri = 0; ri < (int64_t)nrays; ri++) {
        const float *rp = rays + ri * 8;
        SimRay r; r.orig = mk3(rp[0], rp[1], rp[2]); r.dir = mk3(rp[4], rp[5], rp[6]);
        r.dinv = mk3(1.0f / r.dir.x, 1.0f / r.dir.y, 1.0f / r.dir.z);
        const float far = fmaxf_(fmaxf_(absf(r.orig.x), absf(r.orig.y)), absf(r.orig.z));
        const float lim = far < 67108864.0f ? clampNear : 1.8446744e19f;
        const float di[3] = {r.dinv.x, r.dinv.y, r.dinv.z}, og[3] = {r.orig.x, r.orig.y, r.orig.z};
        for (int a = 0; a < 3; a++) { r.dw[a] = fminf_(fmaxf_(di[a], -lim), lim); uint32_t b; memcpy(&b, &r.dw[a], 4); r.neg[a] = (b >> 31) != 0u; }
        float tbest = rp[3]; int tribest = -1;
        const double nv0 = nv;
        std::vector<uint32_t> stack; stack.reserve(64);
        uint32_t cur = w.rootRef;
        bool done = false;
        while (!done) {
            while (!(cur & FLX_WIDE_LEAF_BIT)) {
                nv += 1;
                const flxw::WNode &n = w.nodes[cur];
                const float o[3] = {n.ox, n.oy, n.oz}, sc[3] = {n.sx, n.sy, n.sz};
                const uint32_t ql[3] = {n.qlox, n.qloy, n.qloz}, qh[3] = {n.qhix, n.qhiy, n.qhiz};
                float sd[3], on[3], of[3]; uint32_t qn[3], qf[3];
                for (int a = 0; a < 3; a++) {
                    sd[a] = sc[a] * r.dw[a];
                    const float od = (o[a] - og[a]) * r.dw[a];
                    const float e = 4.76837158e-7f * __builtin_fmaf(absf(sd[a]), 256.0f, absf(od));
                    on[a] = od - e; of[a] = od + e;
                    qn[a] = r.neg[a] ? qh[a] : ql[a]; qf[a] = r.neg[a] ? ql[a] : qh[a];
                }
                uint32_t refs[4] = {n.c0, n.c1, n.c2, n.c3}; float key[4]; bool hit[4];
                for (int c = 0; c < 4; c++) {
                    float tn = -3.0e38f, tf = 3.0e38f;
                    for (int a = 0; a < 3; a++) {
                        const float a_n = __builtin_fmaf((float)((qn[a] >> (8 * c)) & 255u), sd[a], on[a]), a_f = __builtin_fmaf((float)((qf[a] >> (8 * c)) & 255u), sd[a], of[a]);
                        tn = a == 0 ? a_n : fmaxf_(tn, a_n); tf = a == 0 ? a_f : fminf_(tf, a_f);
                    }
                    hit[c] = (tn <= tf) && (tf >= 0.0f) && (tn < tbest); key[c] = tn;
                }
                if (mode == 1) {                       // any hit: last hit slot first, earlier ones pushed in slot order
                    int last = -1; for (int c = 0; c < 4; c++) if (hit[c]) last = c;
                    for (int c = 0; c < last; c++) if (hit[c]) stack.push_back(refs[c]);
                    if (last >= 0) cur = refs[last];
                    else if (stack.empty()) { cur = 0xFFFFFFFFu; }
                    else { cur = stack.back(); stack.pop_back(); }
                } else {
                    const float INF = __builtin_huge_valf();
                    float k[4]; for (int c = 0; c < 4; c++) k[c] = hit[c] ? (mode == 2 ? -key[c] : key[c]) : INF;
                    auto ce = [&](int a, int b) { if (k[b] < k[a]) { std::swap(k[a], k[b]); std::swap(refs[a], refs[b]); } };
                    ce(0, 1); ce(2, 3); ce(0, 2); ce(1, 3); ce(1, 2);
                    for (int c = 3; c >= 1; c--) if (k[c] < INF) stack.push_back(refs[c]);
                    if (k[0] < INF) cur = refs[0];
                    else if (stack.empty()) { cur = 0xFFFFFFFFu; }
                    else { cur = stack.back(); stack.pop_back(); }
                }
                if ((int)stack.size() > deepest) deepest = (int)stack.size();
            }
            if (cur == 0xFFFFFFFFu) break;
            lv += 1;
            const flxw::F4 *lpn = &w.leafdata[cur & FLX_WIDE_OFF_MASK];
            const float bmin[3] = {lpn[0].x, lpn[0].y, lpn[0].z}, bmax[3] = {lpn[1].x, lpn[1].y, lpn[1].z};
            if (sim_slab(bmin, bmax, r, tbest)) {
                lp += 1;
                int cnt; memcpy(&cnt, &lpn[0].w, 4);
                for (int k = 0; k < cnt; k++) {
                    const flxw::F4 &a = lpn[2 + 3 * k], &b = lpn[3 + 3 * k], &c = lpn[4 + 3 * k];
                    tt += 1;
                    float t;
                    if (sim_mt(r.orig, r.dir, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), &t) && t > 0.0f && t < tbest) {
                        if (mode != 0) { memcpy(&tribest, &a.w, 4); done = true; break; }
                        tbest = t; memcpy(&tribest, &a.w, 4);
                    }
                }
            }
            if (done) break;
            if (stack.empty()) break;
            cur = stack.back(); stack.pop_back();
        }
        if (tribest >= 0) { hits += 1; chk += (double)(tribest + 1); }
        if (hitTri) hitTri[ri] = tribest;
        if (nodeVisitsPerRay) nodeVisitsPerRay[ri] = (uint32_t)(nv - nv0);
    }
    out8[0] = nv; out8[1] = lv; out8[2] = lp; out8[3] = tt; out8[4] = hits; out8[5] = deepest; out8[6] = (double)w.nodes.size(); out8[7] = chk;
    FH_CATCH
}


} // extern "C"
