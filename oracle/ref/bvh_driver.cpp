// bvh_driver.cpp -- runs the REFERENCE's own object-split BVH builder (src/bvh.cpp + src/bvhnode.cpp, compiled unmodified where they
// lie: they need only the reference's headers, its vendored cl2.hpp / math headers and the OpenCL headers of the image) on a triangle
// array in wire format and writes the hierarchy with the reference's own BVH::exportTo.  TEST INFRASTRUCTURE (oracle/_ref, this
// container only).  No reference code here.  (The SBVH builder, src/sbvh.cpp, cannot be built the same way: it includes
// progressview.hpp -> glad / GLFW / nanogui.)
#include "bvh.hpp"
#include <cstring>
#include <cstdint>
#include <vector>
#include <string>

static_assert(sizeof(RTTriangle) == 160, "RTTriangle is the 160-byte wire triangle");

// mode: 0 = SplitMode::SAH, 1 = ObjectMedian, 2 = SpatialMedian
extern "C" int ref_bvh_build_export(const void *tris160, uint64_t ntris, int mode, const char *path)
{
    try {
        std::vector<RTTriangle> tris;
        tris.reserve(ntris);
        VertexPNT z;
        for (uint64_t i = 0; i < ntris; i++) { tris.emplace_back(z, z, z); std::memcpy(&tris[i], (const char *)tris160 + i * 160, 160); }
        BVH bvh(&tris, mode == 0 ? SplitMode::SAH : mode == 1 ? SplitMode::ObjectMedian : SplitMode::SpatialMedian);
        bvh.exportTo(path);
        return 0;
    } catch (...) { return 1; }
}
