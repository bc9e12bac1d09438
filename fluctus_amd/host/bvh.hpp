// bvh.hpp -- acceleration-structure builders feeding the traversal kernels (host side).
//
// Output format = the reference's flat arrays (reference: src/bvhnode.hpp:50-59, src/bvh.hpp:55-59):
//   nodes[]   48-byte flx_node, DFS order, left child = index+1, inner: rightChild, leaf: iStart+nPrims
//   indices[] triangle indices referenced by the leaves (>= #tris with spatial splits)
// Triangles are never reordered.
//
// Builders:
//   Mode::SBVH   spatial-split BVH (Stich et al. 2009) with the reference's parameters and
//                decision rules (reference: src/sbvh.cpp:105-157 build, :159-223 sahSplit,
//                :243-324 binSplit, :328-407 partitionSpatial, :410-449 splitReference;
//                parameters src/sbvh.hpp:36-43,70).  This is what Tracer::initHierarchy uses
//                (src/tracer.cpp:574-590).
//   Mode::SAH    full-sweep SAH object-split BVH (reference: src/bvh.cpp:221-256 build, :333-407 sahSplit).
//   Mode::Binned our addition for multi-million-triangle scenes: 32-bin SAH object splits,
//                O(n log n), same output format.
#pragma once
#include <vector>
#include <string>
#include <cstdint>
#include "../../include/fluctus_wire.h"

namespace fluctus {

struct Box {
    float mn[3], mx[3];
    Box() { for (int k = 0; k < 3; k++) { mn[k] = 3.402823466e+38f; mx[k] = -3.402823466e+38f; } }
    void expand(const Box &b) { for (int k = 0; k < 3; k++) { if (b.mn[k] < mn[k]) mn[k] = b.mn[k]; if (b.mx[k] > mx[k]) mx[k] = b.mx[k]; } }
    void expand(const float *p) { for (int k = 0; k < 3; k++) { if (p[k] < mn[k]) mn[k] = p[k]; if (p[k] > mx[k]) mx[k] = p[k]; } }
    void intersect(const Box &b) { for (int k = 0; k < 3; k++) { if (b.mn[k] > mn[k]) mn[k] = b.mn[k]; if (b.mx[k] < mx[k]) mx[k] = b.mx[k]; } }
    float area() const { float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2]; return 2 * (dx * dy + dx * dz + dy * dz); }
};

class BVH {
public:
    enum class Mode { SBVH, SAH, Binned };

    BVH() {}
    BVH(const std::vector<flx_triangle> *tris, Mode mode) { build(tris, mode); }
    void build(const std::vector<flx_triangle> *tris, Mode mode);

    // binary cache in the REFERENCE's on-disk format (src/bvh.cpp:102-192): u32 #indices, the indices, u32 count, then
    // 33 bytes per node {bmin xyz, bmax xyz, u32 iStart/rightChild, i32 parent, u8 nPrims}.  The reference writes
    // m_indices.size() into the node-count field (src/bvh.cpp:185) and reads that many nodes back, which truncates
    // trees with more nodes than indices; here the field carries the true node count on export (so the reference loads
    // our files in full) and import sizes the node array from the file length (so its files load in full here).
    void exportTo(const std::string &filename) const;
    bool importFrom(const std::string &filename);

    void getSceneBounds(float mn[3], float mx[3]) const;   // reference: src/bvh.cpp:53-59
    float worldRadius() const;                             // 0.5*|max-min| (src/tracer.cpp:66-67)

    // SBVH build parallelism: threads 0 = all OpenMP threads, 1 = the serial recursion; job size 0 = n / (8 * threads) references.
    // The tree does not depend on either (tests/test_host.py::test_sbvh_parallel_build_is_the_serial_tree).
    int sbvhThreads = 0;
    size_t sbvhJobSize = 0;
    // cores this process may really use: affinity mask capped by the cgroup v2 CPU quota (a container can show 256 CPUs and grant 16)
    static int usableThreads();

    std::vector<flx_node> m_nodes;
    std::vector<uint32_t> m_indices;
    struct { uint32_t depth = 0, splits = 0, duplicates = 0, spatialSplits = 0; } metrics;

private:
    const std::vector<flx_triangle> *m_tris = nullptr;
};

// XXH64 (Collet): the reference keys its caches by XXH64(scene file bytes, seed 0) printed in decimal
// (src/utils.cpp:63-91, src/scene.cpp:46-51, src/tracer.cpp:576,640)
uint64_t xxh64(const void *data, size_t len, uint64_t seed = 0);
uint64_t fileHash(const std::string &filename);          // throws if the file cannot be read

} // namespace fluctus
