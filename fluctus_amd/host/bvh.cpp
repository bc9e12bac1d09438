// bvh.cpp -- see bvh.hpp.  Own implementation; decision rules follow the cited reference lines so
// that the SBVH of a given mesh is the same tree the reference's builder produces.
#include "bvh.hpp"
#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <parallel/algorithm>      // __gnu_parallel::sort (OpenMP); a strict total order makes its result the serial one
#include <omp.h>
#include <sched.h>
#include <cstdio>

namespace fluctus {

namespace {

struct Ref { uint32_t ind; Box box; };

Box triBox(const flx_triangle &t)
{
    Box b;
    b.expand(&t.v0.p.x); b.expand(&t.v1.p.x); b.expand(&t.v2.p.x);
    return b;
}

// sort key = box.min[d] + box.max[d], ties by triangle index (reference: src/bvh.cpp:258-272)
// `par`: the caller is the only running thread (top of the tree): sort with all cores.  Triangle indices are unique within a node
// (a split triangle's halves go to different children), so (key, ind) is a strict total order and every sorting algorithm, serial
// or parallel, produces the same array.
void sortRefs(std::vector<Ref> &refs, size_t s, size_t e /*inclusive*/, int dim, bool par = false)
{
    auto less = [dim](const Ref &a, const Ref &b) {
        float ca = a.box.mn[dim] + a.box.mx[dim], cb = b.box.mn[dim] + b.box.mx[dim];
        return ca < cb || (ca == cb && a.ind < b.ind);
    };
    if (par && e - s > 50000) __gnu_parallel::sort(refs.begin() + s, refs.begin() + e + 1, less);
    else std::sort(refs.begin() + s, refs.begin() + e + 1, less);
}

struct Split { int i = -1; float pos = 0; int dim = -1; float cost = FLT_MAX; Box left, right; };

struct TreeNode { Box box; int left = -1, right = -1; uint32_t leafStart = 0, leafCount = 0; };

flx_node makeNode(const Box &b, int parent)
{
    flx_node n; std::memset(&n, 0, sizeof(n));
    n.bmin = flx_vec3{b.mn[0], b.mn[1], b.mn[2], 0}; n.bmax = flx_vec3{b.mx[0], b.mx[1], b.mx[2], 0};
    n.parent = parent;
    return n;
}

// ----------------------------------------------------------------------------------- SBVH
struct SbvhBuilder {
    enum { MaxLeaf = 8, MinLeaf = 1, MaxDepth = 64, MaxSpatialDepth = 48, Bins = 128 };   // src/sbvh.hpp:36-43
    const std::vector<flx_triangle> &tris;
    std::vector<Ref> refs;                // stack: a node owns the LAST spec.refs entries
    std::vector<Box> rightBoxes;
    std::vector<TreeNode> tree;
    std::vector<uint32_t> leafInd;        // leaf contents, leaves appended in creation order
    float minOverlap = 0;
    uint32_t depthMax = 0, splits = 0, duplicates = 0, spatialSplits = 0;
    struct Bin { Box bounds; int enter = 0, exit = 0; };
    std::vector<Bin> bins;                // 3 * Bins

    struct Spec { int refs = 0; Box box; };
    bool par = false;                     // this builder works on a top-of-tree node with all cores (parallel sort / binning)

    explicit SbvhBuilder(const std::vector<flx_triangle> &t) : tris(t), bins(3 * Bins) {}

    int leaf(const Spec &s)
    {
        TreeNode n; n.box = s.box; n.leafStart = (uint32_t)leafInd.size(); n.leafCount = (uint32_t)s.refs;
        size_t first = refs.size() - s.refs;
        for (int i = 0; i < s.refs; i++) leafInd.push_back(refs[first + i].ind);   // forward order (see bvh.hpp note)
        refs.resize(first);
        tree.push_back(n);
        return (int)tree.size() - 1;
    }

    // full sweep on 3 axes (reference: src/sbvh.cpp:159-223)
    Split sahSplit(const Spec &s, float nodeSAH)
    {
        Split best; float bestTie = FLT_MAX;
        size_t start = refs.size() - s.refs, end = refs.size() - 1;
        for (int dim = 0; dim < 3; dim++) {
            sortRefs(refs, start, end, dim, par);
            Box rb;
            for (int i = s.refs - 1; i > 0; i--) { rb.expand(refs[start + i].box); rightBoxes[i - 1] = rb; }
            Box lb;
            for (int i = 1; i < s.refs; i++) {
                lb.expand(refs[start + i - 1].box);
                float cl = lb.area() * (float)i;
                float cr = rightBoxes[i - 1].area() * (float)(s.refs - i);
                float cost = nodeSAH + cl + cr;
                float fi = (float)i, fr = (float)(s.refs - i);
                float tie = fi * fi + fr * fr;
                if (cost < best.cost || (cost == best.cost && tie < bestTie)) {
                    best.cost = cost; best.i = i; best.left = lb; best.right = rightBoxes[i - 1]; best.dim = dim; bestTie = tie;
                }
            }
        }
        return best;
    }

    static float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

    // reference: src/sbvh.cpp:410-449
    void splitReference(Ref &L, Ref &R, const Ref &ref, int dim, float coord) const
    {
        L.ind = R.ind = ref.ind; L.box = Box(); R.box = Box();
        const flx_triangle &t = tris[ref.ind];
        const float *v[3] = {&t.v0.p.x, &t.v1.p.x, &t.v2.p.x};
        static const int prev[3] = {2, 0, 1};
        for (int i = 0; i < 3; i++) {
            const float *p2 = v[i], *p1 = v[prev[i]];
            float a = p1[dim], b = p2[dim];
            if (a <= coord) L.box.expand(p1);
            if (a >= coord) R.box.expand(p1);
            if ((a < coord && b > coord) || (a > coord && b < coord)) {
                float tt = std::max(0.0f, std::min(1.0f, (coord - a) / (b - a)));
                float q[3] = {lerpf(p1[0], p2[0], tt), lerpf(p1[1], p2[1], tt), lerpf(p1[2], p2[2], tt)};
                L.box.expand(q); R.box.expand(q);
            }
        }
        L.box.mx[dim] = coord; R.box.mn[dim] = coord;
        L.box.intersect(ref.box); R.box.intersect(ref.box);
    }

    static int toBin(float v) { if (!(v == v) || v < -2147483000.0f) return 0; if (v > 2147483000.0f) return Bins - 1; int i = (int)v; return i < 0 ? 0 : (i > Bins - 1 ? Bins - 1 : i); }

    // chopped binning, 128 bins per axis (reference: src/sbvh.cpp:243-324)
    Split binSplit(const Spec &s, float nodeSAH)
    {
        float origin[3], binSize[3], inv[3];
        for (int k = 0; k < 3; k++) { origin[k] = s.box.mn[k]; binSize[k] = (s.box.mx[k] - origin[k]) * (1.0f / (float)Bins); inv[k] = 1.0f / binSize[k]; }
        for (auto &b : bins) { b.bounds = Box(); b.enter = b.exit = 0; }
        auto accumulate = [&](std::vector<Bin> &into, size_t r0, size_t r1) {
            for (size_t r = r0; r < r1; r++) {
                const Ref &ref = refs[r];
                for (int dim = 0; dim < 3; dim++) {
                    int first = toBin((ref.box.mn[dim] - origin[dim]) * inv[dim]);
                    int last = toBin((ref.box.mx[dim] - origin[dim]) * inv[dim]);
                    if (last < first) last = first;
                    Ref cur = ref;
                    for (int i = first; i < last; i++) {
                        Ref l, rr;
                        float coord = origin[dim] + binSize[dim] * (float)(i + 1);
                        splitReference(l, rr, cur, dim, coord);
                        into[dim * Bins + i].bounds.expand(l.box);
                        cur = rr;
                    }
                    into[dim * Bins + last].bounds.expand(cur.box);
                    into[dim * Bins + first].enter++;
                    into[dim * Bins + last].exit++;
                }
            }
        };
        const size_t r0 = refs.size() - s.refs, r1 = refs.size();
        if (par && s.refs > 50000) {
            // per-thread bins merged with min / max / integer sums: the same bins in any order
            #pragma omp parallel
            {
                std::vector<Bin> local(3 * Bins);
                const int nt = omp_get_num_threads(), t = omp_get_thread_num();
                const size_t chunk = (r1 - r0 + nt - 1) / nt, a = r0 + (size_t)t * chunk, b = std::min(r1, a + chunk);
                if (a < b) accumulate(local, a, b);
                #pragma omp critical
                for (int i = 0; i < 3 * Bins; i++) { bins[i].bounds.expand(local[i].bounds); bins[i].enter += local[i].enter; bins[i].exit += local[i].exit; }
            }
        } else accumulate(bins, r0, r1);
        Split sp;
        for (int dim = 0; dim < 3; dim++) {
            Box rb;
            for (int i = Bins - 1; i > 0; i--) { rb.expand(bins[dim * Bins + i].bounds); rightBoxes[i - 1] = rb; }
            Box lb; int ln = 0, rn = s.refs;
            for (int i = 1; i < Bins; i++) {
                lb.expand(bins[dim * Bins + i - 1].bounds);
                ln += bins[dim * Bins + i - 1].enter;
                rn -= bins[dim * Bins + i - 1].exit;
                float sah = nodeSAH + lb.area() * (float)ln + rightBoxes[i - 1].area() * (float)rn;
                if (sah < sp.cost) { sp.cost = sah; sp.dim = dim; sp.pos = origin[dim] + binSize[dim] * (float)i; }
            }
        }
        return sp;
    }

    // reference: src/sbvh.cpp:225-241
    void partitionObject(Spec &L, Spec &R, const Spec &s, const Split &sp)
    {
        sortRefs(refs, refs.size() - s.refs, refs.size() - 1, sp.dim, par);
        L.refs = sp.i; L.box = sp.left; R.refs = s.refs - sp.i; R.box = sp.right;
    }

    // reference: src/sbvh.cpp:328-407
    void partitionSpatial(Spec &L, Spec &R, const Spec &s, const Split &sp)
    {
        int leftStart = (int)refs.size() - s.refs, leftEnd = leftStart, rightStart = (int)refs.size();
        L.box = Box(); R.box = Box();
        for (int i = leftEnd; i < rightStart; i++) {
            if (refs[i].box.mx[sp.dim] <= sp.pos) { L.box.expand(refs[i].box); std::swap(refs[i], refs[leftEnd++]); }
            else if (refs[i].box.mn[sp.dim] >= sp.pos) { R.box.expand(refs[i].box); std::swap(refs[i--], refs[--rightStart]); }
        }
        while (leftEnd < rightStart) {
            Ref lref, rref;
            splitReference(lref, rref, refs[leftEnd], sp.dim, sp.pos);
            Box lub = L.box, rub = R.box, ldb = L.box, rdb = R.box;
            lub.expand(refs[leftEnd].box); rub.expand(refs[leftEnd].box);
            ldb.expand(lref.box); rdb.expand(rref.box);
            float lac = (float)(leftEnd - leftStart), rac = (float)((int)refs.size() - rightStart);
            float lbc = (float)(leftEnd - leftStart + 1), rbc = (float)((int)refs.size() - rightStart + 1);
            float unsplitLeft = lub.area() * lbc + R.box.area() * rac;
            float unsplitRight = L.box.area() * lac + rub.area() * rbc;
            float dup = ldb.area() * lbc + rdb.area() * rbc;
            float mn = std::min(unsplitLeft, std::min(unsplitRight, dup));
            if (mn == unsplitLeft) { L.box = lub; leftEnd++; }
            else if (mn == unsplitRight) { R.box = rub; std::swap(refs[leftEnd], refs[--rightStart]); }
            else { L.box = ldb; R.box = rdb; refs[leftEnd++] = lref; refs.push_back(rref); }
        }
        L.refs = leftEnd - leftStart; R.refs = (int)refs.size() - rightStart;
    }

    // reference: src/sbvh.cpp:105-157
    int build(Spec &s, int depth)
    {
        depthMax = std::max(depthMax, (uint32_t)depth);
        if (s.refs <= MinLeaf || depth >= MaxDepth) return leaf(s);
        float parentArea = s.box.area();
        float nodeSAH = parentArea * 2 * 1;
        Split obj = sahSplit(s, nodeSAH);
        Split spatial;
        if (depth < MaxSpatialDepth) {
            Box ov = obj.left; ov.intersect(obj.right);
            if (ov.area() >= minOverlap) spatial = binSplit(s, nodeSAH);
        }
        float parentCost = parentArea * (float)s.refs;
        float minCost = std::min(obj.cost, std::min(spatial.cost, parentCost));
        if (minCost == parentCost && s.refs <= MaxLeaf) return leaf(s);
        Spec L, R;
        if (minCost == spatial.cost) { partitionSpatial(L, R, s, spatial); if (L.refs && R.refs) spatialSplits++; }
        if (!L.refs || !R.refs) partitionObject(L, R, s, obj);
        splits++;
        duplicates += (uint32_t)(L.refs + R.refs - s.refs);
        int rn = build(R, depth + 1);       // right first: duplicates live at the end of the ref stack
        int ln = build(L, depth + 1);
        TreeNode n; n.box = s.box; n.left = ln; n.right = rn;
        tree.push_back(n);
        return (int)tree.size() - 1;
    }

    // ---- parallel build.  A node's subtree depends only on its SET of references (every decision sorts them first), so subtrees
    // are independent: the top of the tree -- nodes with more than `big` references -- is split one node after the other with all
    // cores inside each split (parallel sort + binning); every smaller node becomes a JOB, built serially by a private builder, all
    // jobs in parallel.  Same decisions on the same data as the serial recursion, hence the same tree (tests/test_host.py).
    struct Job { std::vector<Ref> refs; Spec spec; int depth; int node; };   // node: placeholder in `tree` to be replaced by the job's root
    std::vector<Job> jobs;

    int buildTop(Spec &s, int depth, size_t big)
    {
        depthMax = std::max(depthMax, (uint32_t)depth);
        if ((size_t)s.refs <= big || s.refs <= MinLeaf || depth >= MaxDepth) {
            Job j; j.spec = s; j.depth = depth;
            j.refs.assign(refs.end() - s.refs, refs.end());
            refs.resize(refs.size() - s.refs);
            tree.push_back(TreeNode()); j.node = (int)tree.size() - 1;
            jobs.push_back(std::move(j));
            return jobs.back().node;
        }
        par = true;
        float parentArea = s.box.area();
        float nodeSAH = parentArea * 2 * 1;
        Split obj = sahSplit(s, nodeSAH);
        Split spatial;
        if (depth < MaxSpatialDepth) {
            Box ov = obj.left; ov.intersect(obj.right);
            if (ov.area() >= minOverlap) spatial = binSplit(s, nodeSAH);
        }
        float parentCost = parentArea * (float)s.refs;
        float minCost = std::min(obj.cost, std::min(spatial.cost, parentCost));
        if (minCost == parentCost && s.refs <= MaxLeaf) { par = false; return leaf(s); }
        Spec L, R;
        if (minCost == spatial.cost) { partitionSpatial(L, R, s, spatial); if (L.refs && R.refs) spatialSplits++; }
        if (!L.refs || !R.refs) partitionObject(L, R, s, obj);
        par = false;
        splits++;
        duplicates += (uint32_t)(L.refs + R.refs - s.refs);
        int rn = buildTop(R, depth + 1, big);
        int ln = buildTop(L, depth + 1, big);
        TreeNode n; n.box = s.box; n.left = ln; n.right = rn;
        tree.push_back(n);
        return (int)tree.size() - 1;
    }

    // build every job with a private serial builder, then graft the subtrees into `tree` / `leafInd`
    void runJobs()
    {
        struct Done { std::vector<TreeNode> tree; std::vector<uint32_t> leafInd; int root; uint32_t depthMax, splits, duplicates, spatialSplits; };
        std::vector<Done> done(jobs.size());
        #pragma omp parallel for schedule(dynamic, 1)
        for (size_t k = 0; k < jobs.size(); k++) {
            SbvhBuilder b(tris);
            b.minOverlap = minOverlap;
            b.refs = std::move(jobs[k].refs);
            b.rightBoxes.resize(std::max(b.refs.size(), (size_t)Bins));
            Spec sp = jobs[k].spec;
            Done &d = done[k];
            d.root = b.build(sp, jobs[k].depth);
            d.tree = std::move(b.tree); d.leafInd = std::move(b.leafInd);
            d.depthMax = b.depthMax; d.splits = b.splits; d.duplicates = b.duplicates; d.spatialSplits = b.spatialSplits;
        }
        for (size_t k = 0; k < jobs.size(); k++) {
            Done &d = done[k];
            const int toff = (int)tree.size(); const uint32_t loff = (uint32_t)leafInd.size();
            for (TreeNode n : d.tree) {
                if (n.left >= 0) { n.left += toff; n.right += toff; } else n.leafStart += loff;
                tree.push_back(n);
            }
            leafInd.insert(leafInd.end(), d.leafInd.begin(), d.leafInd.end());
            tree[jobs[k].node] = tree[toff + d.root];            // the placeholder becomes (a copy of) the job's root node
            depthMax = std::max(depthMax, d.depthMax); splits += d.splits; duplicates += d.duplicates; spatialSplits += d.spatialSplits;
        }
        jobs.clear();
    }

    // DFS, left child = next slot (reference: src/sbvh.cpp:52-73)
    void emit(int t, int parent, std::vector<flx_node> &nodes, std::vector<uint32_t> &indices)
    {
        uint32_t id = (uint32_t)nodes.size();
        nodes.push_back(makeNode(tree[t].box, parent));
        if (tree[t].left < 0) {
            if (tree[t].leafCount > 255) throw std::runtime_error("leaf too large for u8");
            nodes[id].iStartOrRight = (uint32_t)indices.size();
            nodes[id].nPrims = (uint8_t)tree[t].leafCount;
            for (uint32_t i = 0; i < tree[t].leafCount; i++) indices.push_back(leafInd[tree[t].leafStart + i]);
        } else {
            emit(tree[t].left, (int)id, nodes, indices);
            nodes[id].iStartOrRight = (uint32_t)nodes.size();
            emit(tree[t].right, (int)id, nodes, indices);
        }
    }
};

// ----------------------------------------------------------------------------------- SAH / binned
struct ObjBuilder {
    enum { MaxLeaf = 8 };
    std::vector<Ref> refs;
    std::vector<Box> rightBoxes;
    std::vector<flx_node> &nodes;
    uint32_t depthMax = 0, splits = 0;
    bool binned;
    ObjBuilder(std::vector<flx_node> &n, bool b) : nodes(n), binned(b) {}

    static Box boundsOf(const std::vector<Ref> &r, size_t s, size_t e) { Box b; for (size_t i = s; i <= e; i++) b.expand(r[i].box); return b; }

    // reference: src/bvh.cpp:333-407 (areas normalised by the parent area, cost 2*cb + ...)
    size_t sweepSplit(size_t s, size_t e, const Box &box)
    {
        float parentArea = box.area();
        size_t n = e - s + 1; float bestCost = FLT_MAX; size_t bestI = s; int bestDim = 2;
        for (int dim = 0; dim < 3; dim++) {
            sortRefs(refs, s, e, dim);
            Box rb; for (size_t i = 0; i < n; i++) { rb.expand(refs[e - i].box); rightBoxes[i] = rb; }
            Box lb; uint32_t lc = 0;
            for (size_t i = s; i < e; i++) {
                lb.expand(refs[i].box); lc++;
                const Box &r = rightBoxes[e - i - 1];
                float lcost = (float)lc * lb.area() / parentArea, rcost = (float)(n - lc) * r.area() / parentArea;
                float cost = 2.0f + (lcost + rcost);
                if (cost < bestCost) { bestCost = cost; bestI = i; bestDim = dim; }
            }
        }
        if (bestDim != 2) sortRefs(refs, s, e, bestDim);
        if (bestI == s) bestI++; else if (bestI == e) bestI--;   // reference: src/bvh.cpp:394-404
        return bestI;
    }

    size_t binnedSplit(size_t s, size_t e, const Box &)
    {
        const int NB = 32;
        Box cb; for (size_t i = s; i <= e; i++) { float c[3]; for (int k = 0; k < 3; k++) c[k] = 0.5f * (refs[i].box.mn[k] + refs[i].box.mx[k]); cb.expand(c); }
        float bestCost = FLT_MAX; int bestDim = -1, bestBin = -1;
        for (int dim = 0; dim < 3; dim++) {
            float ext = cb.mx[dim] - cb.mn[dim]; if (!(ext > 0)) continue;
            Box bb[NB]; uint32_t cnt[NB] = {0}; float sc = (float)NB / ext;
            for (size_t i = s; i <= e; i++) { float c = 0.5f * (refs[i].box.mn[dim] + refs[i].box.mx[dim]); int b = std::min(NB - 1, std::max(0, (int)((c - cb.mn[dim]) * sc))); bb[b].expand(refs[i].box); cnt[b]++; }
            Box rb[NB]; Box acc; for (int b = NB - 1; b > 0; b--) { acc.expand(bb[b]); rb[b] = acc; }
            Box lb; uint32_t lc = 0, tot = (uint32_t)(e - s + 1);
            for (int b = 0; b < NB - 1; b++) { lb.expand(bb[b]); lc += cnt[b]; if (!lc || lc == tot) continue; float cost = lb.area() * lc + rb[b + 1].area() * (tot - lc); if (cost < bestCost) { bestCost = cost; bestDim = dim; bestBin = b; } }
        }
        if (bestDim < 0) { size_t mid = (s + e) / 2; std::nth_element(refs.begin() + s, refs.begin() + mid, refs.begin() + e + 1, [](const Ref &a, const Ref &b) { return a.ind < b.ind; }); return mid; }
        float ext = cb.mx[bestDim] - cb.mn[bestDim], sc = (float)NB / ext; int d = bestDim, bsel = bestBin; float mn = cb.mn[d];
        auto it = std::stable_partition(refs.begin() + s, refs.begin() + e + 1, [=](const Ref &r) { float c = 0.5f * (r.box.mn[d] + r.box.mx[d]); int b = std::min(NB - 1, std::max(0, (int)((c - mn) * sc))); return b <= bsel; });
        size_t split = (size_t)(it - refs.begin());
        if (split <= s || split > e) split = (s + e + 1) / 2;
        return split - 1;   // last index of the left part
    }

    // reference: src/bvh.cpp:221-256 (left child pushed right after the parent)
    void build(size_t s, size_t e, int parent, uint32_t depth)
    {
        uint32_t id = (uint32_t)nodes.size();
        Box box = boundsOf(refs, s, e);
        nodes.push_back(makeNode(box, parent));
        depthMax = std::max(depthMax, depth);
        size_t n = e - s + 1;
        if (n <= MaxLeaf || depth >= 62) {
            if (n > 255) throw std::runtime_error("leaf too large for u8");
            nodes[id].iStartOrRight = (uint32_t)s; nodes[id].nPrims = (uint8_t)n;
            return;
        }
        size_t i = binned ? binnedSplit(s, e, box) : sweepSplit(s, e, box);
        splits++;
        build(s, i, (int)id, depth + 1);
        nodes[id].iStartOrRight = (uint32_t)nodes.size();
        build(i + 1, e, (int)id, depth + 1);
    }
};

} // namespace

int BVH::usableThreads()
{
    int n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32]; long period = 0;
        if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && quota[0] != 'm' && period > 0) {
            long q = std::atol(quota) / period;
            if (q >= 1 && q < n) n = (int)q;
        }
        std::fclose(f);
    }
    return n < 1 ? 1 : n;
}

void BVH::build(const std::vector<flx_triangle> *tris, Mode mode)
{
    m_tris = tris;
    m_nodes.clear(); m_indices.clear();
    const size_t n = tris->size();
    if (n == 0) throw std::runtime_error("BVH: empty mesh");
    if (mode == Mode::SBVH) {
        SbvhBuilder b(*tris);
        SbvhBuilder::Spec root; root.refs = (int)n;
        b.refs.resize(n);
        for (size_t i = 0; i < n; i++) { b.refs[i].ind = (uint32_t)i; b.refs[i].box = triBox((*tris)[i]); root.box.expand(b.refs[i].box); }
        b.rightBoxes.resize(std::max(n, (size_t)SbvhBuilder::Bins));
        b.minOverlap = root.box.area() * 1e-5f;           // splitAlpha (src/sbvh.hpp:70)
        int r;
        const int threads = sbvhThreads > 0 ? sbvhThreads : usableThreads();
        if (threads <= 1) r = b.build(root, 0);           // the reference's serial recursion
        else {
            const int before = omp_get_max_threads();
            omp_set_num_threads(threads);
            const size_t big = sbvhJobSize > 0 ? sbvhJobSize : std::max((size_t)4096, n / (size_t)(8 * threads));
            r = b.buildTop(root, 0, big);
            b.runJobs();
            omp_set_num_threads(before);
        }
        b.emit(r, -1, m_nodes, m_indices);
        metrics.depth = b.depthMax; metrics.splits = b.splits; metrics.duplicates = b.duplicates; metrics.spatialSplits = b.spatialSplits;
    } else {
        ObjBuilder b(m_nodes, mode == Mode::Binned);
        b.refs.resize(n); b.rightBoxes.resize(n);
        for (size_t i = 0; i < n; i++) { b.refs[i].ind = (uint32_t)i; b.refs[i].box = triBox((*tris)[i]); }
        b.build(0, n - 1, -1, 0);
        m_indices.resize(n);
        for (size_t i = 0; i < n; i++) m_indices[i] = b.refs[i].ind;
        metrics.depth = b.depthMax; metrics.splits = b.splits;
    }
}

void BVH::getSceneBounds(float mn[3], float mx[3]) const
{
    if (m_nodes.empty()) throw std::runtime_error("Cannot get scene bounds from uninitialized BVH");
    mn[0] = m_nodes[0].bmin.x; mn[1] = m_nodes[0].bmin.y; mn[2] = m_nodes[0].bmin.z;
    mx[0] = m_nodes[0].bmax.x; mx[1] = m_nodes[0].bmax.y; mx[2] = m_nodes[0].bmax.z;
}

float BVH::worldRadius() const
{
    float mn[3], mx[3]; getSceneBounds(mn, mx);
    float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) * 0.5f;
}

void BVH::exportTo(const std::string &filename) const
{
    std::ofstream out(filename, std::ios::binary);
    if (!out) return;
    const uint32_t ni = (uint32_t)m_indices.size(), nn = (uint32_t)m_nodes.size();
    out.write((const char *)&ni, 4);
    out.write((const char *)m_indices.data(), (std::streamsize)ni * 4);
    out.write((const char *)&nn, 4);
    std::vector<unsigned char> buf((size_t)nn * 33);
    for (uint32_t i = 0; i < nn; i++) {
        const flx_node &n = m_nodes[i];
        unsigned char *p = &buf[(size_t)i * 33];
        const float box[6] = {n.bmin.x, n.bmin.y, n.bmin.z, n.bmax.x, n.bmax.y, n.bmax.z};
        std::memcpy(p, box, 24); std::memcpy(p + 24, &n.iStartOrRight, 4); std::memcpy(p + 28, &n.parent, 4); p[32] = n.nPrims;
    }
    out.write((const char *)buf.data(), (std::streamsize)buf.size());
}

bool BVH::importFrom(const std::string &filename)
{
    std::ifstream in(filename, std::ios::binary | std::ios::ate);
    if (!in) return false;
    const std::streamoff size = in.tellg();
    in.seekg(0);
    uint32_t ni = 0, header = 0;
    in.read((char *)&ni, 4);
    if (!in || (std::streamoff)(8 + (std::streamoff)ni * 4) > size) return false;
    m_indices.resize(ni);
    in.read((char *)m_indices.data(), (std::streamsize)ni * 4);
    in.read((char *)&header, 4);
    if (!in) return false;
    const std::streamoff rest = size - (8 + (std::streamoff)ni * 4);
    if (rest <= 0 || rest % 33 != 0) return false;
    const size_t nn = (size_t)(rest / 33);                     // not `header`: see bvh.hpp
    std::vector<unsigned char> buf(nn * 33);
    in.read((char *)buf.data(), (std::streamsize)buf.size());
    if (!in) return false;
    m_nodes.assign(nn, flx_node());
    for (size_t i = 0; i < nn; i++) {
        const unsigned char *p = &buf[i * 33];
        flx_node &n = m_nodes[i];
        std::memset(&n, 0, sizeof(n));
        float box[6]; std::memcpy(box, p, 24);
        n.bmin.x = box[0]; n.bmin.y = box[1]; n.bmin.z = box[2]; n.bmax.x = box[3]; n.bmax.y = box[4]; n.bmax.z = box[5];
        std::memcpy(&n.iStartOrRight, p + 24, 4); std::memcpy(&n.parent, p + 28, 4); n.nPrims = p[32];
        // sanity: links must stay inside the arrays
        if (n.nPrims == 0) { if (n.iStartOrRight >= nn || i + 1 >= nn) return false; }
        else if ((size_t)n.iStartOrRight + n.nPrims > ni) return false;
    }
    return true;
}

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const unsigned char *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const unsigned char *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

uint64_t xxh64(const void *data, size_t len, uint64_t seed)
{
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull, P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    const unsigned char *p = (const unsigned char *)data, *end = p + len;
    auto round = [&](uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl64(acc, 31); return acc * P1; };
    auto merge = [&](uint64_t h, uint64_t v) { v = round(0, v); h ^= v; return h * P1 + P4; };
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do { v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24)); p += 32; } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else h = seed + P5;
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p) * P5; h = rotl64(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

uint64_t fileHash(const std::string &filename)
{
    std::ifstream f(filename, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + filename + " for hashing");
    const std::streamoff n = f.tellg();
    std::vector<char> data((size_t)n);
    f.seekg(0);
    if (n) f.read(data.data(), n);
    return xxh64(data.data(), (size_t)n, 0);
}

} // namespace fluctus
