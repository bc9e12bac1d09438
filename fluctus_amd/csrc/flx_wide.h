// flx_wide.h -- the 4-wide, quantised traversal tree: record layouts (host + device) and the host-side collapse of the
// reference's binary node array into it (used by flx_upload_scene).
//
// WHY.  The binary traversal (flx_trace.h, the reference's visit order) is bound by the chain of ~20 DEPENDENT 64-byte
// fetches per ray.  A 4-wide node halves that chain with the SAME request size: one 64-byte line holds
// four child boxes quantised to 8 bits per plane on a per-node power-of-two grid, plus the four child references.
//
// WHAT STAYS EXACT.  The set of LEAVES and the triangles in them are the reference tree's (src/sbvh.cpp / src/bvh.cpp
// leaves, index-list order), only the inner levels are collapsed.  Every quantised box CONTAINS the exact fp32 box of its
// child (and the child's whole subtree, because the reference's boxes are nested -- verified at upload), and the wide slab
// test is conservative with respect to the reference's (flx_trace4.h), so every leaf the reference's traversal reaches is
// reached here.  Each leaf block carries the leaf's EXACT fp32 box; it is tested with the reference's own arithmetic
// (slab(), src/intersect.cl:41-60) before the triangles are, so a leaf's triangles are tested iff the reference tests
// them (for a fixed tMax: any-hit).  Hence k_shadow on this tree is bit-identical to bvh_occluded (src/bvh.cl:312-373);
// k_extend differs from bvh_intersect only in the ORDER leaves are visited in (exact ties in t, and box-vs-triangle
// rounding near-ties -- measured flip rate in DESIGN.md).
#pragma once
#include <stdint.h>
#include <vector>
#include <cmath>
#include <cstring>
#include "../../include/fluctus_wire.h"

namespace flxw {

#define FLX_WIDE_LEAF_BIT 0x80000000u
#define FLX_WIDE_EMPTY    (FLX_WIDE_LEAF_BIT | 0u)   // unused child slot = the dummy leaf at offset 0 of the leaf data (cannot be hit)
#define FLX_WIDE_OFF_MASK 0x7FFFFFFFu
#define FLX_WIDE_COORD_MAX 4.611686e18f          // 2^62: largest |coordinate| of a node box the wide tree accepts (flx_trace4.h: WRay::setup)

// 64 B, 64-B aligned.  Plane k of child c on axis a:  o[a] + q * s[a],  q = byte c of qlo[a] / qhi[a].
struct WNode {
    float ox, oy, oz, sx;            // grid origin, power-of-two scales
    float sy, sz; uint32_t c0, c1;   // child refs: inner = WNode index; leaf = LEAF_BIT | offset of the leaf block in 16-byte units
    uint32_t c2, c3, qlox, qloy;
    uint32_t qloz, qhix, qhiy, qhiz;
};
static_assert(sizeof(WNode) == 64, "WNode must be one 64-byte line");

// Leaf block in the float4 array `leafdata`: header {bmin.xyz, count (int bits)} {bmax.xyz, 0}, then `count` triangles of
// three float4 each: {v0.xyz, triangle index (int bits)} {v1.xyz, 0} {v2.xyz, 0}  (the TriRec of the binary path).
struct F4 { float x, y, z, w; };

struct WideTree {
    std::vector<WNode> nodes;
    std::vector<F4> leafdata;
    uint32_t rootRef = 0;            // WNode 0, or a leaf ref when the whole scene is one leaf
    uint32_t maxStack = 0;           // upper bound of traversal-stack entries any ray can need
    bool nested = true;              // every child box lies inside its parent's (the exactness argument needs it)
    uint32_t maxLeafCount = 0;
};

static inline float pow2f(int e) { return std::ldexp(1.0f, e); }

// Collapse the reference's node array (48-B nodes, left child = i + 1, right child = iStartOrRight, leaf when nPrims > 0;
// src/bvhnode.hpp:50-59) into WNodes.  Returns false with *err set on malformed input.
static inline bool build_wide(const flx_node *nodes, size_t nnodes, const flx_triangle *tris, size_t ntris, const uint32_t *indices, size_t nidx,
                              WideTree &out, const char **err)
{
    auto fail = [&](const char *m) { *err = m; return false; };
    if (!nnodes) return fail("wide tree: empty node array");
    out.nodes.clear(); out.leafdata.clear(); out.nested = true; out.maxLeafCount = 0;
    // ---- leaf blocks, one per leaf node, in node order
    std::vector<uint32_t> leafRef(nnodes, 0);
    {
        size_t total = 0;
        for (size_t i = 0; i < nnodes; i++) if (nodes[i].nPrims) total += 2 + 3 * (size_t)nodes[i].nPrims;
        total += 5;
        if (total >= FLX_WIDE_OFF_MASK) return fail("wide tree: leaf data exceeds the 31-bit offset range");
        out.leafdata.reserve(total + 5);
        // dummy leaf for unused child slots: a point box far outside every scene and one degenerate triangle (det == 0: never hit).
        // An unused slot's quantised box is inverted (lo plane 255, hi plane 0) and fails the node test except for rays more than
        // ~2^28 grid cells away (the conservative shift e); those land here and find nothing.
        {
            int one = 1; float fc; memcpy(&fc, &one, 4);
            int none = -1; float fn; memcpy(&fn, &none, 4);
            const float F = 3.0e38f;
            out.leafdata.push_back({F, F, F, fc}); out.leafdata.push_back({F, F, F, 0.0f});
            out.leafdata.push_back({F, F, F, fn}); out.leafdata.push_back({F, F, F, 0.0f}); out.leafdata.push_back({F, F, F, 0.0f});
        }
        for (size_t i = 0; i < nnodes; i++) {
            const flx_node &n = nodes[i];
            if (!n.nPrims) continue;
            if ((size_t)n.iStartOrRight + n.nPrims > nidx) return fail("wide tree: leaf range outside the index list");
            leafRef[i] = FLX_WIDE_LEAF_BIT | (uint32_t)out.leafdata.size();
            int cnt = n.nPrims; float fc; memcpy(&fc, &cnt, 4);
            out.leafdata.push_back({n.bmin.x, n.bmin.y, n.bmin.z, fc});
            out.leafdata.push_back({n.bmax.x, n.bmax.y, n.bmax.z, 0.0f});
            if ((uint32_t)n.nPrims > out.maxLeafCount) out.maxLeafCount = n.nPrims;
            for (uint32_t k = 0; k < n.nPrims; k++) {
                const uint32_t ti = indices[n.iStartOrRight + k];
                if (ti >= ntris) return fail("wide tree: triangle index out of range");
                const flx_triangle &t = tris[ti];
                const float pc[9] = {t.v0.p.x, t.v0.p.y, t.v0.p.z, t.v1.p.x, t.v1.p.y, t.v1.p.z, t.v2.p.x, t.v2.p.y, t.v2.p.z};
                for (float v : pc) if (!std::isfinite(v)) return fail("wide tree: triangle with a NaN or infinite vertex");
                int idx = (int)ti; float fi; memcpy(&fi, &idx, 4);
                out.leafdata.push_back({t.v0.p.x, t.v0.p.y, t.v0.p.z, fi});
                out.leafdata.push_back({t.v1.p.x, t.v1.p.y, t.v1.p.z, 0.0f});
                out.leafdata.push_back({t.v2.p.x, t.v2.p.y, t.v2.p.z, 0.0f});
            }
        }
    }
    {
        const flx_node &r0 = nodes[0];
        const float rb[6] = {r0.bmin.x, r0.bmin.y, r0.bmin.z, r0.bmax.x, r0.bmax.y, r0.bmax.z};
        for (float v : rb) if (!std::isfinite(v) || std::fabs(v) > FLX_WIDE_COORD_MAX) return fail("wide tree: root box not finite or beyond +-2^62");
    }
    if (nodes[0].nPrims) {               // the whole scene is one leaf
        out.rootRef = leafRef[0]; out.maxStack = 1;
        out.nodes.resize(1); memset(out.nodes.data(), 0, sizeof(WNode));
        return true;
    }
    auto area = [&](uint32_t i) {
        const flx_node &n = nodes[i];
        const double dx = (double)n.bmax.x - n.bmin.x, dy = (double)n.bmax.y - n.bmin.y, dz = (double)n.bmax.z - n.bmin.z;
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<uint8_t> seen(nnodes, 0);
    // ---- which binary inner nodes become wide nodes: the SAH-optimal choice (Ylitie, Karras, Laine 2017, section 3).  The leaves are
    // fixed, so the expected traversal cost is  sum over the binary nodes that become wide nodes of their surface area.
    // cost[n][i-1] = least such sum for the subtree of n represented as a FOREST of at most i roots (i = 1..3); a wide node rooted at n
    // hands its 4 slots to the two subtrees in the cheapest way (k | 4 - k).  Nodes are in DFS order (children after parents).
    std::vector<float> cost;      // 3 per node
    std::vector<uint8_t> split4;  // per inner node: slots given to the LEFT subtree when n roots a wide node (1..3)
    std::vector<uint8_t> splitF;  // per inner node: for the forest of <= 2 / <= 3 roots: low nibble i=2, high nibble i=3; 0 = "use fewer roots"
    {
        cost.assign(nnodes * 3, 0.0f); split4.assign(nnodes, 0); splitF.assign(nnodes, 0);
        for (size_t ii = nnodes; ii-- > 0;) {
            if (nodes[ii].nPrims) continue;                                   // leaf: 0 for every i
            const uint32_t l = (uint32_t)ii + 1, r = nodes[ii].iStartOrRight;
            if (l >= nnodes || r >= nnodes || r <= ii) return fail("wide tree: child index out of range");
            auto C = [&](uint32_t n, int i) { return cost[(size_t)n * 3 + (i - 1)]; };     // i in 1..3
            auto distribute = [&](int j, int *bestk) {                        // j roots over the two children, each >= 1
                float best = 3.0e38f; int bk = 1;
                for (int k = 1; k < j; k++) { const float c = C(l, k > 3 ? 3 : k) + C(r, (j - k) > 3 ? 3 : (j - k)); if (c < best) { best = c; bk = k; } }
                *bestk = bk; return best;
            };
            int k4 = 1;
            const float c1 = (float)area((uint32_t)ii) + distribute(4, &k4);
            split4[ii] = (uint8_t)k4;
            cost[ii * 3 + 0] = c1;
            int k2 = 1, k3 = 1;
            const float d2 = distribute(2, &k2), d3 = distribute(3, &k3);
            float c2 = c1; uint8_t f = 0;
            if (d2 < c2) { c2 = d2; f |= (uint8_t)k2; }
            float c3 = c2; uint8_t f3 = 0;
            if (d3 < c3) { c3 = d3; f3 = (uint8_t)k3; }
            cost[ii * 3 + 1] = c2; cost[ii * 3 + 2] = c3;
            splitF[ii] = (uint8_t)(f | (f3 << 4));
        }
    }
    // the binary nodes that form the forest of at most `i` roots under n, left to right
    auto forest = [&](uint32_t n, int i, uint32_t *outSlots, int &ns, auto &&self) -> bool {
        if (nodes[n].nPrims || i == 1) { outSlots[ns++] = n; return true; }
        const uint32_t l = n + 1, r = nodes[n].iStartOrRight;
        int k = i == 3 ? (splitF[n] >> 4) : 0;
        if (i == 3 && k == 0) return self(n, 2, outSlots, ns, self);         // three roots are no better than two
        if (i == 2) { k = splitF[n] & 15; if (k == 0) { outSlots[ns++] = n; return true; } }
        if (seen[n]) return false;
        seen[n] = 1;                                                          // opened: it will not get a record of its own
        return self(l, k, outSlots, ns, self) && self(r, i - k, outSlots, ns, self);
    };

    // ---- collapse, wide nodes numbered so that the (up to 4) inner children of a node are consecutive records
    struct Item { uint32_t bin; uint32_t wide; uint32_t stackAbove; };      // binary inner node -> wide record; stack entries pending above it
    std::vector<Item> todo;
    out.nodes.resize(1);
    todo.push_back({0u, 0u, 0u});
    seen[0] = 1;
    out.rootRef = 0; out.maxStack = 0;
    while (!todo.empty()) {
        const Item it = todo.back(); todo.pop_back();
        uint32_t slots[4]; int ns = 0;
        {
            const uint32_t l = it.bin + 1, r = nodes[it.bin].iStartOrRight;
            if (l >= nnodes || r >= nnodes || r <= it.bin) return fail("wide tree: child index out of range");
            slots[ns++] = l; slots[ns++] = r;
        }
        {
            const uint32_t l = slots[0], r = slots[1];
            const int k = split4[it.bin];
            ns = 0;
            if (!forest(l, k, slots, ns, forest) || !forest(r, 4 - k, slots, ns, forest)) return fail("wide tree: node reachable twice (cyclic or shared node array)");
        }
        // nesting: every slot's box inside this node's box (transitively, through the opened intermediate nodes)
        const flx_node &P = nodes[it.bin];
        float omin[3] = {P.bmin.x, P.bmin.y, P.bmin.z}, omax[3] = {P.bmax.x, P.bmax.y, P.bmax.z};
        float cmin[4][3], cmax[4][3];
        for (int k = 0; k < ns; k++) {
            const flx_node &c = nodes[slots[k]];
            cmin[k][0] = c.bmin.x; cmin[k][1] = c.bmin.y; cmin[k][2] = c.bmin.z; cmax[k][0] = c.bmax.x; cmax[k][1] = c.bmax.y; cmax[k][2] = c.bmax.z;
            for (int a = 0; a < 3; a++) {
                if (!(cmin[k][a] >= omin[a]) || !(cmax[k][a] <= omax[a])) out.nested = false;
                if (!(cmin[k][a] <= cmax[k][a])) return fail("wide tree: inverted or NaN child box");
                // infinite bounds pass the comparison above; the grid below needs finite extents (log2 of inf is undefined behaviour
                // when converted to int), and FLX_WIDE_COORD_MAX keeps (o - orig) * dinv of the node test finite (flx_trace4.h)
                if (!std::isfinite(cmin[k][a]) || !std::isfinite(cmax[k][a]) || std::fabs(cmin[k][a]) > FLX_WIDE_COORD_MAX || std::fabs(cmax[k][a]) > FLX_WIDE_COORD_MAX)
                    return fail("wide tree: node box not finite or beyond +-2^62");
            }
        }
        // grid: origin = min over the children, per-axis power-of-two scale with o + 255 s >= max over the children
        float o[3], s[3]; uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
        for (int a = 0; a < 3; a++) {
            float lo = cmin[0][a], hi = cmax[0][a];
            for (int k = 1; k < ns; k++) { lo = cmin[k][a] < lo ? cmin[k][a] : lo; hi = cmax[k][a] > hi ? cmax[k][a] : hi; }
            o[a] = lo;
            const long double ext = (long double)hi - (long double)lo;
            int e = ext > 0 ? (int)std::ceil(std::log2((double)(ext / 255.0L))) : -108;
            if (e < -108) e = -108;
            for (;; e++) {               // (re)quantise until every plane fits 8 bits
                const long double sc = std::ldexp(1.0L, e);
                bool ok = true; uint32_t pl = 0, ph = 0;
                for (int k = 0; k < 4 && ok; k++) {
                    if (k >= ns) { pl |= 255u << (8 * k); continue; }              // empty slot: inverted box (lo plane 255, hi plane 0)
                    const long double dl = (long double)cmin[k][a] - lo, dh = (long double)cmax[k][a] - lo;
                    long double ql = std::floor(dl / sc), qh = std::ceil(dh / sc);
                    while (ql > 0 && ql * sc > dl) ql -= 1;                       // REAL o + ql*s <= bmin  and  o + qh*s >= bmax
                    while (qh * sc < dh) qh += 1;
                    if (ql < 0) ql = 0;
                    if (qh > 255) { ok = false; break; }
                    pl |= (uint32_t)ql << (8 * k); ph |= (uint32_t)qh << (8 * k);
                }
                if (ok) { qlo[a] = pl; qhi[a] = ph; s[a] = pow2f(e); break; }
                if (e > 120) return fail("wide tree: box extent out of range");
            }
        }
        WNode w;
        w.ox = o[0]; w.oy = o[1]; w.oz = o[2]; w.sx = s[0]; w.sy = s[1]; w.sz = s[2];
        w.qlox = qlo[0]; w.qloy = qlo[1]; w.qloz = qlo[2]; w.qhix = qhi[0]; w.qhiy = qhi[1]; w.qhiz = qhi[2];
        // child refs; inner children get consecutive new records
        uint32_t refs[4] = {FLX_WIDE_EMPTY, FLX_WIDE_EMPTY, FLX_WIDE_EMPTY, FLX_WIDE_EMPTY};
        const uint32_t pending = it.stackAbove + (uint32_t)(ns - 1);         // a visit pushes at most ns - 1 entries
        if (pending > out.maxStack) out.maxStack = pending;
        for (int k = 0; k < ns; k++) {
            const uint32_t c = slots[k];
            if (nodes[c].nPrims) { refs[k] = leafRef[c]; continue; }
            if (seen[c]) return fail("wide tree: node reachable twice (cyclic or shared node array)");
            seen[c] = 1;
            refs[k] = (uint32_t)out.nodes.size();
            out.nodes.push_back(WNode());
        }
        for (int k = ns - 1; k >= 0; k--)                                      // left subtree processed first (pre-order-ish numbering)
            if (!nodes[slots[k]].nPrims) todo.push_back({slots[k], refs[k], pending});
        w.c0 = refs[0]; w.c1 = refs[1]; w.c2 = refs[2]; w.c3 = refs[3];
        out.nodes[it.wide] = w;
    }
    out.maxStack += 1;
    return true;
}

} // namespace flxw
