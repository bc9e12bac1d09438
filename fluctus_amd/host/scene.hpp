// scene.hpp -- mesh / material ingest for the wavefront hot path (host side).
//
// Mirrors the public surface of the reference's Scene (reference: src/scene.hpp:26-39:
// loadModel, getTriangles, getMaterials, getTextures, getMaterialTypes) for the formats the
// hot path's configs use: ASCII PLY (src/scene.cpp:422-553) and OBJ+MTL
// (src/scene.cpp:171-301).  Output is the wire format of include/fluctus_wire.h
// (160-byte triangles, 80-byte materials), i.e. exactly what uploadSceneData() ships.
// PBRT/PBF ingest is SURVEY 8(f) N1 ("next").
#pragma once
#include <string>
#include <vector>
#include <cstdint>
#include "../../include/fluctus_wire.h"

namespace fluctus {

struct Texture {            // RGBA8, lower-left origin (reference: src/texture.cpp:16-41)
    std::string name;
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;
};

class Scene {
public:
    Scene();                                        // material 0 = default diffuse (scene.cpp:13-26)
    void loadModel(const std::string &filename);    // dispatch on extension (scene.cpp:53-103)
    void loadPlyModel(const std::string &filename);
    void loadObjWithMaterials(const std::string &filename);
    void loadPBRTModel(const std::string &filename);   // pbrt-v3 text scenes (reference: .pbrt -> .pbf -> loadPBFModel, scene.cpp:74-90, 574-813); pbrt.cpp

    // Deterministic procedural stand-ins for the configs whose assets are missing from the
    // reference checkout; "kitchen" | "conference" | "courtyard".
    void generate(const std::string &kind, uint32_t targetTris, uint32_t seed);

    std::vector<flx_triangle> &getTriangles() { return triangles; }
    std::vector<flx_material> &getMaterials() { return materials; }
    std::vector<Texture> &getTextures() { return textures; }
    uint32_t getMaterialTypes() const { return materialTypes; }
    int addTexture(Texture &&t) { textures.push_back(std::move(t)); return (int)textures.size() - 1; }
    // Scene::tryImportTexture (reference: src/scene.cpp:303-330): reuse a texture already loaded under `name`, else decode the
    // file (PNG / JPEG); -1 when missing or undecodable
    int tryImportTexture(const std::string &path, const std::string &name);
    flx_vec3 getWorldUp() const { return worldUp; }
    int addMaterial(const flx_material &m) { materials.push_back(m); materialTypes |= (uint32_t)m.type; return (int)materials.size() - 1; }

    // packTextures (reference: src/clcontext.cpp:570-611): one byte blob + descriptors
    void packTextures(std::vector<flx_texdesc> &descs, std::vector<uint8_t> &blob) const;

    static int parseShaderType(const std::string &type);   // scene.cpp:171-189

private:
    std::vector<flx_triangle> triangles;
    std::vector<flx_material> materials;
    std::vector<Texture> textures;
    uint32_t materialTypes = 0;
    flx_vec3 worldUp {0.0f, 1.0f, 0.0f, 0.0f};          // reference: Scene::worldUp, set by the PBRT path from the camera frame
};

} // namespace fluctus
