// ARCHIVED EXPERIMENT (round 4; not part of the product: built only into tests/_build/libwide_analysis.so for scripts/exp_tree_opt.py).
// Measured on the device (profiles/r04_wide_opt_ab.txt, two passes): node visits per closest-hit ray 10.64 -> 10.55 kitchen, 11.74 -> 11.12
// conference, 25.99 -> 25.64 courtyard; k_trace4r -0.7 % / -3 % / 0 %, k_shadow4 -2.5 % / -3 % / -1.4 %, step +0.4 % / +0.5..3 % / +-0; upload +0.75 s /
// +0.25 s / +25 s.  The round-3 verdict asked for -8 % visits on kitchen AND courtyard: the reference builder's SBVH topology is already within a
// few per cent of what reinsertion finds.  Not shipped.
//
// flx_wide_opt.h -- re-optimisation of the INNER topology of the traversal tree before it is collapsed into 4-wide nodes (flx_wide.h).
//
// WHY.  Both traversal kernels are VALU-issue-bound and pay ~140 VALU instructions per wide-node visit, so the lever that is
// left is the NUMBER of visits.  Round 2 collapsed the reference's binary SBVH topology into wide nodes with the SAH-optimal choice of which
// binary nodes survive; the topology itself was the reference builder's greedy top-down one.
//
// WHAT IS FREE.  The parity contract pins the LEAVES (their exact fp32 boxes, their triangles, the order of the triangles inside a leaf), not
// the inner levels: a leaf's triangles are tested iff the leaf's exact box passes the reference's slab test (flx_trace4.h: wide_leaf_visit),
// and by monotone rounding a leaf box that passes implies that every conservative box around it passes, the reference's own ancestors
// included -- so the set of leaves a ray tests is independent of ANY conservative inner hierarchy over the same leaves.  The any-hit query
// (order-free) stays bit-identical to bvh_occluded (src/bvh.cl:312-373); the closest-hit query visits the same leaves in another order
// and stays inside the tie budget it already has (src/bvh.cl:234-310; tests/test_gpu_wide.py counts the flips).
//
// HOW.  Insertion-based optimisation of the binary hierarchy (Bittner, Hapala, Havran 2013, in the subtree-reinsertion form of Meister &
// Bittner 2018): take a subtree out (its parent is replaced by its sibling), find the position in the rest of the tree where putting it
// back increases the total surface area of the inner nodes least (best-first branch and bound over "induced cost"), insert it there with
// the freed parent record.  The original position is among the candidates, so the cost never rises.  Nodes are processed in descending
// order of surface area, a few passes.  Inner boxes are the exact fp32 union of the leaf boxes below them (min / max are exact), so they are
// nested by construction.  The result is written as a node array in the reference's wire format (48-B nodes, DFS order, left child = i + 1)
// over the SAME index list, and build_wide() runs on it unchanged.
#pragma once
#include <stdint.h>
#include <vector>
#include <queue>
#include <algorithm>
#include <cstring>
#include "../../include/fluctus_wire.h"
#include "../../fluctus_amd/csrc/flx_wide.h"

namespace flxw {

struct OptStats {
    double costBefore = 0.0, costAfter = 0.0;      // sum of inner-node half-areas / root half-area (binary SAH, Cinner = 1, leaves not counted)
    uint64_t moved = 0, searched = 0, searchSteps = 0;
    uint32_t depthBefore = 0, depthAfter = 0;
    int passes = 0;
};

namespace optdetail {
struct ONode {
    float mn[3], mx[3];
    float area;
    int32_t parent, left, right;               // leaf: left = -1, right = index of the leaf in the input node array
};
static inline float half_area(const float *mn, const float *mx)
{
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
}
static inline float union_area(const ONode &a, const ONode &b)
{
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) { mn[k] = a.mn[k] < b.mn[k] ? a.mn[k] : b.mn[k]; mx[k] = a.mx[k] > b.mx[k] ? a.mx[k] : b.mx[k]; }
    return half_area(mn, mx);
}
}

// Returns false (with *err) on malformed input; `out` then is untouched.  passes <= 0: `out` = a copy of the input.
static inline bool optimise_topology(const flx_node *nodes, size_t nnodes, int passes, std::vector<flx_node> &out, OptStats *stats, const char **err)
{
    using namespace optdetail;
    auto fail = [&](const char *m) { *err = m; return false; };
    if (!nnodes) return fail("tree optimiser: empty node array");
    if (passes <= 0 || nodes[0].nPrims) { out.assign(nodes, nodes + nnodes); return true; }
    std::vector<ONode> t(nnodes);
    for (size_t i = 0; i < nnodes; i++) {
        ONode &n = t[i];
        n.parent = -1;
        if (nodes[i].nPrims) {
            n.left = -1; n.right = (int32_t)i;
            n.mn[0] = nodes[i].bmin.x; n.mn[1] = nodes[i].bmin.y; n.mn[2] = nodes[i].bmin.z;
            n.mx[0] = nodes[i].bmax.x; n.mx[1] = nodes[i].bmax.y; n.mx[2] = nodes[i].bmax.z;
            for (int k = 0; k < 3; k++) if (!(n.mn[k] <= n.mx[k])) return fail("tree optimiser: inverted or NaN leaf box");
        } else {
            const uint32_t l = (uint32_t)i + 1, r = nodes[i].iStartOrRight;
            if (l >= nnodes || r >= nnodes || r <= i) return fail("tree optimiser: child index out of range");
            n.left = (int32_t)l; n.right = (int32_t)r;
        }
    }
    // parents + reachability (every node exactly once), then boxes bottom-up: children come after their parent in the input
    {
        std::vector<uint8_t> seen(nnodes, 0);
        seen[0] = 1;
        for (size_t i = 0; i < nnodes; i++) {
            if (t[i].left < 0) continue;
            if (!seen[i]) return fail("tree optimiser: node unreachable from the root");
            for (int32_t c : {t[i].left, t[i].right}) { if (seen[c]++) return fail("tree optimiser: node reachable twice"); t[c].parent = (int32_t)i; }
        }
        for (size_t i = 0; i < nnodes; i++) if (!seen[i]) return fail("tree optimiser: node unreachable from the root");
    }
    auto refit = [&](int32_t i) {
        ONode &n = t[i]; const ONode &a = t[n.left], &b = t[n.right];
        for (int k = 0; k < 3; k++) { n.mn[k] = a.mn[k] < b.mn[k] ? a.mn[k] : b.mn[k]; n.mx[k] = a.mx[k] > b.mx[k] ? a.mx[k] : b.mx[k]; }
        n.area = half_area(n.mn, n.mx);
    };
    for (size_t i = nnodes; i-- > 0;) { if (t[i].left < 0) t[i].area = half_area(t[i].mn, t[i].mx); else refit((int32_t)i); }
    int32_t root = 0;
    auto total_cost = [&]() { double c = 0.0; for (size_t i = 0; i < nnodes; i++) if (t[i].left >= 0) c += t[i].area; return c / (double)t[root].area; };
    auto depth_of = [&]() {
        uint32_t best = 0; std::vector<std::pair<int32_t, uint32_t>> st; st.push_back({root, 1u});
        while (!st.empty()) { auto [i, d] = st.back(); st.pop_back(); if (d > best) best = d; if (t[i].left >= 0) { st.push_back({t[i].left, d + 1}); st.push_back({t[i].right, d + 1}); } }
        return best;
    };
    if (stats) { stats->costBefore = total_cost(); stats->depthBefore = depth_of(); }
    auto refit_up = [&](int32_t i) {                 // recompute boxes from i to the root, stopping when nothing changes any more
        while (i >= 0) {
            ONode &n = t[i]; float mn[3], mx[3]; const ONode &a = t[n.left], &b = t[n.right];
            bool same = true;
            for (int k = 0; k < 3; k++) {
                mn[k] = a.mn[k] < b.mn[k] ? a.mn[k] : b.mn[k]; mx[k] = a.mx[k] > b.mx[k] ? a.mx[k] : b.mx[k];
                same = same && mn[k] == n.mn[k] && mx[k] == n.mx[k];
            }
            if (same) break;
            for (int k = 0; k < 3; k++) { n.mn[k] = mn[k]; n.mx[k] = mx[k]; }
            n.area = half_area(mn, mx);
            i = n.parent;
        }
    };
    struct Cand { float induced; int32_t node; bool operator<(const Cand &o) const { return induced > o.induced; } };      // min-heap on the induced cost
    std::vector<Cand> heap;
    std::vector<int32_t> order(nnodes);
    for (int pass = 0; pass < passes; pass++) {
        for (size_t i = 0; i < nnodes; i++) order[i] = (int32_t)i;
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return t[a].area != t[b].area ? t[a].area > t[b].area : a < b; });
        uint64_t movedThisPass = 0;
        for (int32_t x : order) {
            const int32_t p = t[x].parent;
            if (p < 0) continue;                                  // the root
            const int32_t g = t[p].parent;
            if (g < 0) continue;                                  // child of the root: taking it out would leave the sibling as the whole tree; skipped
            const int32_t s = t[p].left == x ? t[p].right : t[p].left;
            // ---- take x (and the record p) out: s moves up into p's place
            if (t[g].left == p) t[g].left = s; else t[g].right = s;
            t[s].parent = g;
            refit_up(g);
            // ---- best-first search for the position whose insertion cost (direct + induced on the ancestors) is least
            const ONode &X = t[x];
            float bestCost = 3.0e38f; int32_t best = -1;
            heap.clear(); heap.push_back({0.0f, root});
            if (stats) stats->searched++;
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end()); const Cand c = heap.back(); heap.pop_back();
                if (c.induced + X.area >= bestCost) break;        // nothing below any remaining candidate can beat the best: the union is at least X
                if (stats) stats->searchSteps++;
                const ONode &N = t[c.node];
                const float direct = union_area(N, X);
                const float cost = c.induced + direct;
                if (cost < bestCost) { bestCost = cost; best = c.node; }
                if (N.left >= 0) {
                    const float ind = cost - N.area;              // what inserting BELOW this node adds to it
                    if (ind + X.area < bestCost) {
                        heap.push_back({ind, N.left}); std::push_heap(heap.begin(), heap.end());
                        heap.push_back({ind, N.right}); std::push_heap(heap.begin(), heap.end());
                    }
                }
            }
            // ---- put it back: p becomes the parent of (best, x) in best's place
            const int32_t bp = t[best].parent;
            t[p].left = best; t[p].right = x; t[p].parent = bp;
            t[best].parent = p; t[x].parent = p;
            if (bp < 0) root = p; else if (t[bp].left == best) t[bp].left = p; else t[bp].right = p;
            refit(p);
            refit_up(bp);
            if (best != s) movedThisPass++;
        }
        if (stats) { stats->moved += movedThisPass; stats->passes = pass + 1; }
        if (movedThisPass * 200 < nnodes) break;                  // fewer than 0.5 % of the nodes moved: converged
    }
    if (stats) { stats->costAfter = total_cost(); stats->depthAfter = depth_of(); }
    // ---- write the tree in the wire format: DFS, left child = next record; leaves keep their box, index range and count
    std::vector<flx_node> res(nnodes);
    {
        std::vector<std::pair<int32_t, int32_t>> st;               // (node, index of its parent's record or -1)
        std::vector<int32_t> pendingRight;                          // records whose right child index is still to be filled, as a stack
        size_t w = 0;
        struct Frame { int32_t node; int32_t parentRec; bool isRight; };
        std::vector<Frame> stack; stack.push_back({root, -1, false});
        while (!stack.empty()) {
            const Frame f = stack.back(); stack.pop_back();
            const ONode &n = t[f.node];
            const size_t rec = w++;
            if (f.isRight) res[f.parentRec].iStartOrRight = (uint32_t)rec;
            flx_node &o = res[rec];
            memset(&o, 0, sizeof(o));
            if (n.left < 0) { o = nodes[n.right]; o.parent = f.parentRec; continue; }
            o.bmin.x = n.mn[0]; o.bmin.y = n.mn[1]; o.bmin.z = n.mn[2]; o.bmax.x = n.mx[0]; o.bmax.y = n.mx[1]; o.bmax.z = n.mx[2];
            o.parent = f.parentRec; o.nPrims = 0;
            stack.push_back({n.right, (int32_t)rec, true});
            stack.push_back({n.left, (int32_t)rec, false});         // popped first: lands at rec + 1
        }
        if (w != nnodes) return fail("tree optimiser: internal error (node count changed)");
    }
    out.swap(res);
    return true;
}

// Slot order inside a wide node.  The closest-hit query sorts the hit children by entry distance and the far -> near any-hit order does too,
// so for them the slot order is irrelevant; the any-hit order "LAST hit slot first, earlier hits pushed" (flx_trace4.h: ANY_ORDER 0, rays toward
// an area light) descends in SLOT order.  The answer is the same whatever the order (any hit), the number of nodes visited before the first
// occluder is not: putting the child that is most likely to hold an occluder LAST makes it the one taken first.  key: 1 = surface area of the
// child's box (ascending), 2 = number of triangles below the child (ascending), 3 = triangles per unit area (ascending), 0 = leave as built.
// Slots are permuted in place (references and the six plane bytes of each slot); empty slots stay behind the used ones.
static inline void reorder_slots(WideTree &w, int key)
{
    if (key <= 0 || (w.rootRef & FLX_WIDE_LEAF_BIT) || w.nodes.empty()) return;
    const size_t n = w.nodes.size();
    std::vector<double> tris(n, 0.0);
    auto slot_ref = [](const WNode &nd, int c) { return c == 0 ? nd.c0 : c == 1 ? nd.c1 : c == 2 ? nd.c2 : nd.c3; };
    auto byte = [](uint32_t v, int c) { return (v >> (8 * c)) & 255u; };
    for (size_t i = n; i-- > 0;) {                                         // children are numbered after their parent
        WNode &nd = w.nodes[i];
        double cnt[4], area[4]; uint32_t refs[4]; int used = 0;
        for (int c = 0; c < 4; c++) {
            refs[c] = slot_ref(nd, c);
            cnt[c] = 0.0; area[c] = 0.0;
            if (refs[c] == FLX_WIDE_EMPTY) continue;
            used = c + 1;
            if (refs[c] & FLX_WIDE_LEAF_BIT) { int k; memcpy(&k, &w.leafdata[refs[c] & FLX_WIDE_OFF_MASK].w, 4); cnt[c] = k; }
            else cnt[c] = tris[refs[c]];
            const double ex = (double)(byte(nd.qhix, c) - (double)byte(nd.qlox, c)) * nd.sx, ey = (double)(byte(nd.qhiy, c) - (double)byte(nd.qloy, c)) * nd.sy,
                         ez = (double)(byte(nd.qhiz, c) - (double)byte(nd.qloz, c)) * nd.sz;
            area[c] = ex * ey + ey * ez + ez * ex;
            tris[i] += cnt[c];
        }
        int perm[4] = {0, 1, 2, 3};
        auto val = [&](int c) { return key == 1 ? area[c] : key == 2 ? cnt[c] : (area[c] > 0.0 ? cnt[c] / area[c] : 1e300); };
        std::stable_sort(perm, perm + used, [&](int a, int b) { return val(a) < val(b); });
        uint32_t q[6] = {nd.qlox, nd.qloy, nd.qloz, nd.qhix, nd.qhiy, nd.qhiz}, qn[6] = {0, 0, 0, 0, 0, 0}, rn[4];
        for (int c = 0; c < 4; c++) {
            rn[c] = refs[perm[c]];
            for (int k = 0; k < 6; k++) qn[k] |= byte(q[k], perm[c]) << (8 * c);
        }
        nd.c0 = rn[0]; nd.c1 = rn[1]; nd.c2 = rn[2]; nd.c3 = rn[3];
        nd.qlox = qn[0]; nd.qloy = qn[1]; nd.qloz = qn[2]; nd.qhix = qn[3]; nd.qhiy = qn[4]; nd.qhiz = qn[5];
    }
}

} // namespace flxw
