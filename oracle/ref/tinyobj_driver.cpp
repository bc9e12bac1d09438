// tinyobj_driver.cpp -- runs the reference's VENDORED third-party OBJ/MTL parser (include/tiny_obj_loader.h, header-only: it
// builds from its own single file) the way Scene::loadObjWithMaterials calls it (reference: src/scene.cpp:191-215) and hands the
// parsed arrays out, flattened per face corner.  TEST INFRASTRUCTURE (oracle/_ref build, this container only).  No reference code
// here: the conventions the reference applies on top (matId + 1, flat normal when a corner has none, `shader` key -> BSDF type;
// src/scene.cpp:171-189, 236-301) are restated in tests/test_host.py and compared with fluctus_amd/host/scene.cpp's own parser.
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

struct TinyObjResult {
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    uint64_t ntris = 0;
};

extern "C" {

int ref_tinyobj_load(const char *path, const char *mtldir, void **out)
{
    TinyObjResult *r = new TinyObjResult();
    bool ok = tinyobj::LoadObj(&r->attrib, &r->shapes, &r->materials, &r->err, path, mtldir);
    if (!ok) { delete r; return 1; }
    for (auto &s : r->shapes) r->ntris += s.mesh.indices.size() / 3;
    *out = r;
    return 0;
}
void ref_tinyobj_free(void *h) { delete (TinyObjResult *)h; }
int ref_tinyobj_counts(void *h, uint64_t *ntris, uint64_t *nmats, int *hasNormals, int *hasTexCoords)
{
    TinyObjResult *r = (TinyObjResult *)h;
    *ntris = r->ntris; *nmats = r->materials.size();
    *hasNormals = r->attrib.normals.size() > 0; *hasTexCoords = r->attrib.texcoords.size() > 0;
    return 0;
}
// per face corner: position (3), normal (3, zeros when the corner has no normal index), normal index, texcoord (2), texcoord index;
// per face: material id as tinyobj reports it (-1 = none)
int ref_tinyobj_faces(void *h, float *pos, float *nrm, int *nidx, float *uv, int *tidx, int *matid)
{
    TinyObjResult *r = (TinyObjResult *)h;
    uint64_t t = 0;
    for (auto &s : r->shapes) {
        if (s.mesh.indices.size() % 3) return 1;
        for (size_t f = 0; f < s.mesh.indices.size() / 3; f++, t++) {
            for (int v = 0; v < 3; v++) {
                const tinyobj::index_t ind = s.mesh.indices[3 * f + v];
                for (int k = 0; k < 3; k++) pos[(t * 3 + v) * 3 + k] = r->attrib.vertices[3 * ind.vertex_index + k];
                nidx[t * 3 + v] = ind.normal_index; tidx[t * 3 + v] = ind.texcoord_index;
                for (int k = 0; k < 3; k++) nrm[(t * 3 + v) * 3 + k] = ind.normal_index >= 0 && !r->attrib.normals.empty() ? r->attrib.normals[3 * ind.normal_index + k] : 0.0f;
                for (int k = 0; k < 2; k++) uv[(t * 3 + v) * 2 + k] = ind.texcoord_index >= 0 && !r->attrib.texcoords.empty() ? r->attrib.texcoords[2 * ind.texcoord_index + k] : 0.0f;
            }
            matid[t] = s.mesh.material_ids[f];
        }
    }
    return 0;
}
// per material: Kd Ks Ke (9 floats), Ns, Ni; names = 4 x 256 chars: map_Kd, map_Ks, map_bump, unknown parameter "shader"
int ref_tinyobj_materials(void *h, float *vals11, char *names)
{
    TinyObjResult *r = (TinyObjResult *)h;
    for (size_t i = 0; i < r->materials.size(); i++) {
        tinyobj::material_t &m = r->materials[i];
        float *v = vals11 + i * 11;
        for (int k = 0; k < 3; k++) { v[k] = m.diffuse[k]; v[3 + k] = m.specular[k]; v[6 + k] = m.emission[k]; }
        v[9] = m.shininess; v[10] = m.ior;
        const std::string s[4] = {m.diffuse_texname, m.specular_texname, m.bump_texname, m.unknown_parameter["shader"]};
        for (int k = 0; k < 4; k++) { char *d = names + (i * 4 + k) * 256; std::memset(d, 0, 256); std::strncpy(d, s[k].c_str(), 255); }
    }
    return 0;
}

}
