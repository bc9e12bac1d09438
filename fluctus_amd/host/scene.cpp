// scene.cpp -- see scene.hpp.  Own implementation; behaviour follows the cited reference lines.
#include "scene.hpp"
#include "texture.hpp"
#include <fstream>
#include <sstream>
#include <map>
#include <array>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <stdexcept>
#include <algorithm>

namespace fluctus {

namespace {

struct V3 { float x = 0, y = 0, z = 0; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalize(V3 a) { float l = std::sqrt(dot(a, a)); float inv = l > 0 ? 1.0f / l : 0.0f; return a * inv; }
inline flx_vec3 W(V3 v) { return flx_vec3{v.x, v.y, v.z, 0.0f}; }

flx_triangle makeTri(V3 p0, V3 p1, V3 p2, V3 n0, V3 n1, V3 n2, V3 t0, V3 t1, V3 t2, int matId)
{
    flx_triangle t;
    std::memset(&t, 0, sizeof(t));
    t.v0.p = W(p0); t.v1.p = W(p1); t.v2.p = W(p2);
    t.v0.n = W(n0); t.v1.n = W(n1); t.v2.n = W(n2);
    t.v0.t = W(t0); t.v1.t = W(t1); t.v2.t = W(t2);
    t.matId = matId;
    return t;
}

bool endsWith(const std::string &s, const std::string &e)
{
    return s.size() >= e.size() && s.compare(s.size() - e.size(), e.size(), e) == 0;
}

flx_material defaultMaterial()
{
    flx_material m;
    std::memset(&m, 0, sizeof(m));
    m.Kd = flx_vec3{0.64f, 0.64f, 0.64f, 0.0f};
    m.Ni = 1.8f; m.Ns = 700.0f;
    m.map_Kd = m.map_Ks = m.map_N = -1;
    m.type = FLX_BXDF_DIFFUSE;
    return m;
}

} // namespace

Scene::Scene()
{
    addMaterial(defaultMaterial());   // reference: scene.cpp:13-26
}

int Scene::tryImportTexture(const std::string &path, const std::string &name)
{
    for (size_t i = 0; i < textures.size(); i++) if (textures[i].name == name) return (int)i;
    if (!fileExists(path)) return -1;
    try { Texture t = loadTexture(path); t.name = name; return addTexture(std::move(t)); } catch (const std::exception &) { return -1; }
}

int Scene::parseShaderType(const std::string &type)
{
    if (type == "diffuse") return FLX_BXDF_DIFFUSE;
    if (type == "glossy") return FLX_BXDF_GLOSSY;
    if (type == "rough_reflection") return FLX_BXDF_GGX_ROUGH_REFLECTION;
    if (type == "ideal_reflection") return FLX_BXDF_IDEAL_REFLECTION;
    if (type == "rough_dielectric") return FLX_BXDF_GGX_ROUGH_DIELECTRIC;
    if (type == "ideal_dielectric") return FLX_BXDF_IDEAL_DIELECTRIC;
    if (type == "emissive") return FLX_BXDF_EMISSIVE;
    return FLX_BXDF_DIFFUSE;
}

void Scene::loadModel(const std::string &filename)
{
    if (endsWith(filename, "obj")) loadObjWithMaterials(filename);
    else if (endsWith(filename, "ply")) loadPlyModel(filename);
    else if (endsWith(filename, "pbrt")) loadPBRTModel(filename);
    else throw std::runtime_error("Scene::loadModel: unsupported format: " + filename);
}

// ASCII PLY: header elements/properties, per-vertex normals share the vertex index, faces are
// triangles or quads (split 0-1-2 / 2-3-0), texcoords zero.  reference: scene.cpp:422-553,815-861
void Scene::loadPlyModel(const std::string &filename)
{
    std::ifstream in(filename);
    if (!in) throw std::runtime_error("cannot open " + filename);
    struct Element { std::string name; int lines; std::vector<std::string> props; };
    std::vector<Element> elems;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream iss(line);
        std::string s; iss >> s;
        if (s == "element") { Element e; iss >> e.name >> e.lines; elems.push_back(e); }
        else if (s == "property" && !elems.empty()) { std::string ty, nm; iss >> ty >> nm; if (ty == "list") { std::string a, b; iss >> a; nm = a; iss >> b; nm = b; } elems.back().props.push_back(nm); }
        else if (s == "end_header") break;
    }
    std::vector<V3> pos, nrm;
    std::vector<std::array<unsigned, 3>> faces;
    for (const Element &e : elems) {
        if (e.name == "vertex") {
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
            for (int k = 0; k < (int)e.props.size(); k++) {   // last property of a name wins (std::map semantics)
                const std::string &p = e.props[k];
                if (p == "x") ix = k; else if (p == "y") iy = k; else if (p == "z") iz = k;
                else if (p == "nx") inx = k; else if (p == "ny") iny = k; else if (p == "nz") inz = k;
            }
            std::vector<float> vals(e.props.size());
            for (int i = 0; i < e.lines; i++) {
                std::getline(in, line);
                std::istringstream iss(line);
                std::string tok;
                for (size_t k = 0; k < vals.size(); k++) { tok.clear(); iss >> tok; vals[k] = (float)atof(tok.c_str()); }
                auto get = [&](int k) { return k >= 0 ? vals[k] : 0.0f; };
                pos.push_back({get(ix), get(iy), get(iz)});
                if (inx >= 0) nrm.push_back({get(inx), get(iny), get(inz)});
            }
        } else if (e.name == "face") {
            for (int i = 0; i < e.lines; i++) {
                std::getline(in, line);
                std::istringstream iss(line);
                int n = 0; iss >> n;
                if (n == 3) { unsigned a, b, c; iss >> a >> b >> c; faces.push_back({a, b, c}); }
                else if (n == 4) { unsigned a, b, c, d; iss >> a >> b >> c >> d; faces.push_back({a, b, c}); faces.push_back({c, d, a}); }
                else throw std::runtime_error("PLY: unknown polygon type");
            }
        } else {
            for (int i = 0; i < e.lines; i++) std::getline(in, line);
        }
    }
    for (auto &f : faces) {
        V3 p0 = pos[f[0]], p1 = pos[f[1]], p2 = pos[f[2]], n0, n1, n2;
        if (nrm.empty()) n0 = n1 = n2 = normalize(cross(p1 - p0, p2 - p0));
        else { n0 = nrm[f[0]]; n1 = nrm[f[1]]; n2 = nrm[f[2]]; }
        triangles.push_back(makeTri(p0, p1, p2, n0, n1, n2, V3{}, V3{}, V3{}, 0));
    }
}

namespace {

// Minimal MTL reader: the keys the reference consumes through tinyobj (scene.cpp:284-300);
// defaults follow tinyobj's InitMaterial (zeros, shininess 1, ior 1).
struct MtlEntry { std::string name; flx_material m; std::string mapKd, mapKs, mapBump, shader; };

std::vector<MtlEntry> readMtl(const std::string &path)
{
    std::vector<MtlEntry> out;
    std::ifstream in(path);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream iss(line);
        std::string k; iss >> k;
        if (k.empty() || k[0] == '#') continue;
        if (k == "newmtl") {
            MtlEntry e; iss >> e.name;
            std::memset(&e.m, 0, sizeof(e.m));
            e.m.Ns = 1.0f; e.m.Ni = 1.0f; e.m.map_Kd = e.m.map_Ks = e.m.map_N = -1; e.m.type = FLX_BXDF_DIFFUSE;
            out.push_back(e);
            continue;
        }
        if (out.empty()) continue;
        MtlEntry &e = out.back();
        auto rd3 = [&](flx_vec3 &v) { iss >> v.x >> v.y >> v.z; };
        if (k == "Kd") rd3(e.m.Kd); else if (k == "Ks") rd3(e.m.Ks); else if (k == "Ke") rd3(e.m.Ke);
        else if (k == "Ns") iss >> e.m.Ns; else if (k == "Ni") iss >> e.m.Ni;
        else if (k == "map_Kd") iss >> e.mapKd; else if (k == "map_Ks") iss >> e.mapKs;
        // case-sensitive like the reference's vendored tinyobj (include/tiny_obj_loader.h:1220-1229): `map_Bump`, as Country-Kitchen.mtl
        // spells it, is NOT a bump map there (it lands in unknown_parameter), so the reference renders the carrots without one
        else if (k == "map_bump" || k == "bump") iss >> e.mapBump;
        else if (k == "shader") iss >> e.shader;
    }
    return out;
}

} // namespace

// reference: scene.cpp:191-301.  matId = mtl index + 1 (0 = default material); flat normal when
// any vertex normal is missing; texcoord zero when absent.  map_Kd / map_Ks / map_bump images are decoded by
// tryImportTexture (PNG, JPEG; texture.cpp, jpeg.cpp); a missing or undecodable file leaves the slot at -1, and a texture
// registered under the same name with addTexture() beforehand is reused.
void Scene::loadObjWithMaterials(const std::string &filePath)
{
    std::ifstream in(filePath);
    if (!in) throw std::runtime_error("cannot open " + filePath);
    size_t slash = filePath.find_last_of("/\\");
    std::string folder = slash == std::string::npos ? "" : filePath.substr(0, slash + 1);
    std::vector<V3> P, N, T;
    std::map<std::string, int> mtlIndex;
    int matBase = (int)materials.size();   // == 1 for a fresh scene
    int curMat = -1;
    std::string line;
    auto texLookup = [&](const std::string &nm) -> int {
        if (nm.empty()) return -1;
        std::string unix = nm; for (char &ch : unix) if (ch == '\\') ch = '/';
        return tryImportTexture(folder + unix, unix);
    };
    while (std::getline(in, line)) {
        if (line.size() < 2) continue;
        const char *s = line.c_str();
        if (s[0] == 'v' && s[1] == ' ') { V3 v; sscanf(s + 2, "%f %f %f", &v.x, &v.y, &v.z); P.push_back(v); }
        else if (s[0] == 'v' && s[1] == 'n') { V3 v; sscanf(s + 3, "%f %f %f", &v.x, &v.y, &v.z); N.push_back(v); }
        else if (s[0] == 'v' && s[1] == 't') { V3 v; sscanf(s + 3, "%f %f", &v.x, &v.y); T.push_back(v); }
        else if (s[0] == 'f' && s[1] == ' ') {
            struct Idx { int v, t, n; };
            std::vector<Idx> poly;
            std::istringstream iss(line.substr(2));
            std::string tok;
            while (iss >> tok) {
                Idx ix{0, 0, 0};
                const char *c = tok.c_str(); char *end;
                ix.v = (int)strtol(c, &end, 10);
                if (*end == '/') { c = end + 1; if (*c != '/') { ix.t = (int)strtol(c, &end, 10); } else end = (char *)c; if (*end == '/') ix.n = (int)strtol(end + 1, &end, 10); }
                auto fix = [](int i, size_t n) { return i > 0 ? i - 1 : (i < 0 ? (int)n + i : -1); };
                ix.v = fix(ix.v, P.size()); ix.t = fix(ix.t, T.size()); ix.n = fix(ix.n, N.size());
                poly.push_back(ix);
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) {   // fan triangulation
                Idx id[3] = {poly[0], poly[k], poly[k + 1]};
                V3 p[3], n[3], t[3]; bool allN = true;
                for (int v = 0; v < 3; v++) {
                    p[v] = P[id[v].v];
                    if (id[v].n < 0 || N.empty()) { allN = false; n[v] = V3{}; } else n[v] = N[id[v].n];
                    t[v] = (id[v].t >= 0 && !T.empty()) ? T[id[v].t] : V3{};
                }
                if (!allN) n[0] = n[1] = n[2] = normalize(cross(p[1] - p[0], p[2] - p[0]));
                triangles.push_back(makeTri(p[0], p[1], p[2], n[0], n[1], n[2], t[0], t[1], t[2], curMat + 1 == 0 ? 0 : matBase + curMat));
            }
        }
        else if (line.compare(0, 6, "usemtl") == 0) {
            std::istringstream iss(line.substr(6)); std::string nm; iss >> nm;
            auto it = mtlIndex.find(nm);
            curMat = it == mtlIndex.end() ? -1 : it->second;
        }
        else if (line.compare(0, 6, "mtllib") == 0) {
            std::istringstream iss(line.substr(6)); std::string nm; iss >> nm;
            for (MtlEntry &e : readMtl(folder + nm)) {
                e.m.map_Kd = texLookup(e.mapKd); e.m.map_Ks = texLookup(e.mapKs); e.m.map_N = texLookup(e.mapBump);
                e.m.type = parseShaderType(e.shader);
                mtlIndex[e.name] = (int)materials.size() - matBase;
                addMaterial(e.m);
            }
        }
    }
}

void Scene::packTextures(std::vector<flx_texdesc> &descs, std::vector<uint8_t> &blob) const
{
    descs.clear(); blob.clear();
    uint32_t offset = 0;
    for (const Texture &t : textures) {
        flx_texdesc d{offset, t.width, t.height};
        descs.push_back(d);
        blob.insert(blob.end(), t.rgba.begin(), t.rgba.end());
        offset += t.width * t.height * 4;
    }
}

// ---------------------------------------------------------------------------------------
// Procedural stand-ins: deterministic, LCG-seeded.
// ---------------------------------------------------------------------------------------
namespace {

struct Lcg {
    uint32_t s;
    explicit Lcg(uint32_t seed) : s(seed * 2654435761u + 12345u) {}
    uint32_t next() { s = s * 1664525u + 1013904223u; return s; }
    float uni() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
    float range(float a, float b) { return a + (b - a) * uni(); }
};

struct Builder {
    std::vector<flx_triangle> &tris;
    void quadGrid(V3 o, V3 eu, V3 ev, int nu, int nv, int mat, float uvScale = 1.0f)
    {
        V3 n = normalize(cross(eu, ev));
        for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) {
            float u0 = (float)i / nu, u1 = (float)(i + 1) / nu, v0 = (float)j / nv, v1 = (float)(j + 1) / nv;
            V3 p00 = o + eu * u0 + ev * v0, p10 = o + eu * u1 + ev * v0, p11 = o + eu * u1 + ev * v1, p01 = o + eu * u0 + ev * v1;
            V3 t00{u0 * uvScale, v0 * uvScale, 0}, t10{u1 * uvScale, v0 * uvScale, 0}, t11{u1 * uvScale, v1 * uvScale, 0}, t01{u0 * uvScale, v1 * uvScale, 0};
            tris.push_back(makeTri(p00, p10, p11, n, n, n, t00, t10, t11, mat));
            tris.push_back(makeTri(p00, p11, p01, n, n, n, t00, t11, t01, mat));
        }
    }
    void box(V3 c, V3 h, int n, int mat)   // axis-aligned box, n x n quads per face
    {
        V3 X{h.x, 0, 0}, Y{0, h.y, 0}, Z{0, 0, h.z};
        quadGrid(c - X - Y + Z, X * 2, Y * 2, n, n, mat);          // +z
        quadGrid(c + X - Y - Z, X * -2.0f, Y * 2, n, n, mat);      // -z
        quadGrid(c + X - Y + Z, Z * -2.0f, Y * 2, n, n, mat);      // +x
        quadGrid(c - X - Y - Z, Z * 2, Y * 2, n, n, mat);          // -x
        quadGrid(c - X + Y + Z, X * 2, Z * -2.0f, n, n, mat);      // +y
        quadGrid(c - X - Y - Z, X * 2, Z * 2, n, n, mat);          // -y
    }
    // generic parametric surface p(u,v), smooth normals from finite differences of the param
    template <class Fn> void surface(Fn f, int nu, int nv, int mat, bool flip = false)
    {
        std::vector<V3> P((size_t)(nu + 1) * (nv + 1)), Nn(P.size());
        auto at = [&](int i, int j) -> size_t { return (size_t)j * (nu + 1) + i; };
        for (int j = 0; j <= nv; j++) for (int i = 0; i <= nu; i++) P[at(i, j)] = f((float)i / nu, (float)j / nv);
        const float e = 1e-3f;
        for (int j = 0; j <= nv; j++) for (int i = 0; i <= nu; i++) {
            float u = (float)i / nu, v = (float)j / nv;
            V3 du = f(u + e, v) - f(u - e, v), dv = f(u, v + e) - f(u, v - e);
            V3 n = normalize(cross(du, dv));
            if (!(dot(n, n) > 0.5f)) n = normalize(P[at(i, j)] - f(0.5f, 0.5f));
            Nn[at(i, j)] = flip ? n * -1.0f : n;
        }
        for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) {
            size_t a = at(i, j), b = at(i + 1, j), c = at(i + 1, j + 1), d = at(i, j + 1);
            V3 ta{(float)i / nu, (float)j / nv, 0}, tb{(float)(i + 1) / nu, (float)j / nv, 0}, tc{(float)(i + 1) / nu, (float)(j + 1) / nv, 0}, td{(float)i / nu, (float)(j + 1) / nv, 0};
            if (flip) { tris.push_back(makeTri(P[a], P[c], P[b], Nn[a], Nn[c], Nn[b], ta, tc, tb, mat)); tris.push_back(makeTri(P[a], P[d], P[c], Nn[a], Nn[d], Nn[c], ta, td, tc, mat)); }
            else { tris.push_back(makeTri(P[a], P[b], P[c], Nn[a], Nn[b], Nn[c], ta, tb, tc, mat)); tris.push_back(makeTri(P[a], P[c], P[d], Nn[a], Nn[c], Nn[d], ta, tc, td, mat)); }
        }
    }
    void sphere(V3 c, float r, int n, int mat)
    {
        surface([=](float u, float v) { float th = u * 6.2831853f, ph = v * 3.14159265f; return c + V3{std::sin(ph) * std::cos(th), std::cos(ph), std::sin(ph) * std::sin(th)} * r; }, 2 * n, n, mat, false);
    }
    void torus(V3 c, float R, float r, int n, int mat)
    {
        surface([=](float u, float v) { float th = u * 6.2831853f, ph = v * 6.2831853f; float q = R + r * std::cos(ph); return c + V3{q * std::cos(th), r * std::sin(ph), q * std::sin(th)}; }, 2 * n, n, mat, true);
    }
    void cylinder(V3 c, float r, float h, int n, int mat)
    {
        surface([=](float u, float v) { float th = u * 6.2831853f; return c + V3{r * std::cos(th), v * h, r * std::sin(th)}; }, 2 * n, n, mat, true);
        surface([=](float u, float v) { float th = u * 6.2831853f; return c + V3{v * r * std::cos(th), h, v * r * std::sin(th)}; }, 2 * n, std::max(1, n / 4), mat, false);
    }
    void blob(V3 c, float r, int n, int mat, float bump, float freq)   // bumpy sphere (foliage / pottery)
    {
        surface([=](float u, float v) { float th = u * 6.2831853f, ph = v * 3.14159265f; float rr = r * (1.0f + bump * std::sin(freq * th) * std::sin(freq * ph)); return c + V3{std::sin(ph) * std::cos(th), std::cos(ph), std::sin(ph) * std::sin(th)} * rr; }, 2 * n, n, mat, false);
    }
};

Texture makeChecker(uint32_t size, uint32_t cells, uint32_t seed, const std::string &name)
{
    Texture t; t.name = name; t.width = t.height = size; t.rgba.resize((size_t)size * size * 4);
    Lcg rng(seed);
    uint8_t a[3], b[3];
    for (int k = 0; k < 3; k++) { a[k] = (uint8_t)(64 + rng.next() % 160); b[k] = (uint8_t)(32 + rng.next() % 96); }
    for (uint32_t y = 0; y < size; y++) for (uint32_t x = 0; x < size; x++) {
        bool on = (((x * cells) / size) + ((y * cells) / size)) & 1u;
        uint32_t h = (x * 73856093u) ^ (y * 19349663u) ^ seed; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        int noise = (int)(h & 15u) - 8;
        uint8_t *p = &t.rgba[((size_t)y * size + x) * 4];
        for (int k = 0; k < 3; k++) { int v = (on ? a[k] : b[k]) + noise; p[k] = (uint8_t)std::min(255, std::max(0, v)); }
        p[3] = 255;
    }
    return t;
}

flx_material mkMat(int type, V3 kd, V3 ks, float ns, float ni, int mapKd)
{
    flx_material m; std::memset(&m, 0, sizeof(m));
    m.Kd = W(kd); m.Ks = W(ks); m.Ns = ns; m.Ni = ni; m.map_Kd = mapKd; m.map_Ks = -1; m.map_N = -1; m.type = type;
    return m;
}

} // namespace

void Scene::generate(const std::string &kind, uint32_t targetTris, uint32_t seed)
{
    Builder B{triangles};
    Lcg rng(seed);
    auto col = [&]() { return V3{rng.range(0.15f, 0.9f), rng.range(0.15f, 0.9f), rng.range(0.15f, 0.9f)}; };

    if (kind == "kitchen") {
        // material-type mix counted from assets/country_kitchen/Country-Kitchen.mtl:
        // 57 diffuse, 21 glossy, 8 ideal_reflection, 6 rough_reflection, 4 ideal_dielectric = 96
        for (int i = 0; i < 17; i++) addTexture(makeChecker(512, 8 + 2 * (i % 5), seed * 131u + i, "proc_checker_" + std::to_string(i)));
        std::vector<int> mats;
        int texN = 0;
        for (int i = 0; i < 96; i++) {
            int type = i < 57 ? FLX_BXDF_DIFFUSE : i < 78 ? FLX_BXDF_GLOSSY : i < 86 ? FLX_BXDF_IDEAL_REFLECTION
                     : i < 92 ? FLX_BXDF_GGX_ROUGH_REFLECTION : FLX_BXDF_IDEAL_DIELECTRIC;
            int mapKd = (type == FLX_BXDF_DIFFUSE && texN < 17 && (i % 3) == 0) ? texN++ : -1;
            float ns = type == FLX_BXDF_GGX_ROUGH_REFLECTION ? rng.range(30.0f, 3000.0f) : type == FLX_BXDF_GLOSSY ? rng.range(50.0f, 1000.0f) : 96.078431f;
            float ni = type == FLX_BXDF_IDEAL_DIELECTRIC ? 1.5f : type == FLX_BXDF_GLOSSY ? 1.5f : 1.0f;
            V3 ks = type == FLX_BXDF_IDEAL_DIELECTRIC ? V3{0.95f, 0.95f, 0.95f} : V3{0.5f, 0.5f, 0.5f};
            mats.push_back(addMaterial(mkMat(type, col(), ks, ns, ni, mapKd)));
        }
        // interleave so round-robin assignment mixes types
        std::vector<int> order; for (int i = 0; i < 96; i++) order.push_back(mats[(i * 37) % 96]);
        int mi = 0; auto nextMat = [&]() { return order[(mi++) % 96]; };
        // room 6 x 3 x 5 m, open front (+z) and a skylight strip, env light enters through both
        int wallN = 48;
        B.quadGrid(V3{-3, 0, -2.5f}, V3{6, 0, 0}, V3{0, 0, 5}, wallN, wallN, mats[0], 6.0f);            // floor (textured)
        B.quadGrid(V3{-3, 0, -2.5f}, V3{0, 3, 0}, V3{0, 0, 5}, wallN, wallN / 2, mats[1]);              // left wall (+x normal)
        B.quadGrid(V3{3, 0, 2.5f}, V3{0, 3, 0}, V3{0, 0, -5}, wallN, wallN / 2, mats[2]);               // right wall
        B.quadGrid(V3{3, 0, -2.5f}, V3{0, 3, 0}, V3{-6, 0, 0}, wallN, wallN / 2, mats[4]);              // back wall
        B.quadGrid(V3{-3, 3, 0.5f}, V3{6, 0, 0}, V3{0, 0, -3}, wallN, wallN / 2, mats[5]);              // ceiling over the back part
        // counter + table
        B.box(V3{0, 0.45f, -2.0f}, V3{2.8f, 0.45f, 0.35f}, 24, nextMat());
        B.box(V3{0.4f, 0.40f, 0.3f}, V3{1.1f, 0.04f, 0.6f}, 24, nextMat());
        for (int k = 0; k < 4; k++) B.cylinder(V3{0.4f + (k & 1 ? 1.0f : -1.0f), 0, 0.3f + (k & 2 ? 0.5f : -0.5f)}, 0.04f, 0.36f, 12, nextMat());
        // 40 props
        size_t base = triangles.size();
        uint32_t remaining = targetTris > base ? targetTris - (uint32_t)base : 40u * 200u;
        uint32_t perProp = remaining / 40u;
        int n = std::max(4, (int)std::sqrt((double)perProp / 4.0));   // surfaces emit 4*n*n tris
        for (int i = 0; i < 40; i++) {
            float x, y, z;
            if (i < 14) { x = -2.5f + 0.38f * i; y = 0.9f; z = -2.0f; }              // on the counter
            else if (i < 24) { x = -0.5f + 0.2f * (i - 14); y = 0.44f; z = 0.1f + 0.1f * ((i - 14) % 4); }   // on the table
            else { x = rng.range(-2.6f, 2.6f); y = 0.0f; z = rng.range(-1.2f, 2.0f); }  // on the floor
            float r = i < 24 ? rng.range(0.07f, 0.13f) : rng.range(0.15f, 0.32f);
            int m = nextMat();
            switch (i % 4) {
            case 0: B.sphere(V3{x, y + r, z}, r, n, m); break;
            case 1: B.torus(V3{x, y + r * 0.35f, z}, r, r * 0.35f, n, m); break;
            case 2: B.blob(V3{x, y + r, z}, r, n, m, 0.08f, 6.0f); break;
            default: B.cylinder(V3{x, y, z}, r * 0.6f, r * 2.0f, (int)(n * 0.9f), m); break;
            }
        }
    } else if (kind == "conference") {
        // overridden mix: 50% GGX rough reflection (Ns from conference.mtl), 25% glossy, 25% diffuse
        const float nsTable[4] = {32.0f, 302.0f, 602.0f, 3200.0f};
        std::vector<int> mats;
        for (int i = 0; i < 32; i++) {
            int type = (i % 4) < 2 ? FLX_BXDF_GGX_ROUGH_REFLECTION : (i % 4) == 2 ? FLX_BXDF_GLOSSY : FLX_BXDF_DIFFUSE;
            mats.push_back(addMaterial(mkMat(type, col(), V3{0.6f, 0.6f, 0.6f}, nsTable[(i / 4) % 4], type == FLX_BXDF_GLOSSY ? 1.5f : 1.0f, -1)));
        }
        int mi = 0; auto nextMat = [&]() { return mats[(mi++) % mats.size()]; };
        int wallN = 40;
        // closed room 8 x 3 x 6 around the origin (area light at its default position (1,1,0) is inside)
        B.quadGrid(V3{-4, -0.5f, -3}, V3{8, 0, 0}, V3{0, 0, 6}, wallN, wallN, nextMat());
        B.quadGrid(V3{-4, 2.5f, 3}, V3{8, 0, 0}, V3{0, 0, -6}, wallN, wallN, nextMat());
        B.quadGrid(V3{-4, -0.5f, -3}, V3{0, 3, 0}, V3{0, 0, 6}, wallN, wallN / 2, nextMat());
        B.quadGrid(V3{4, -0.5f, 3}, V3{0, 3, 0}, V3{0, 0, -6}, wallN, wallN / 2, nextMat());
        B.quadGrid(V3{4, -0.5f, -3}, V3{0, 3, 0}, V3{-8, 0, 0}, wallN, wallN / 2, nextMat());
        B.quadGrid(V3{-4, -0.5f, 3}, V3{0, 3, 0}, V3{8, 0, 0}, wallN, wallN / 2, nextMat());
        B.box(V3{0, 0.2f, 0}, V3{2.2f, 0.04f, 0.9f}, 32, nextMat());   // table top
        for (int k = 0; k < 4; k++) B.cylinder(V3{(k & 1 ? 2.0f : -2.0f), -0.5f, (k & 2 ? 0.7f : -0.7f)}, 0.06f, 0.66f, 12, nextMat());
        size_t base = triangles.size();
        uint32_t remaining = targetTris > base ? targetTris - (uint32_t)base : 24u * 400u;
        uint32_t perChair = remaining / 24u;
        int n = std::max(3, (int)std::sqrt((double)perChair / (6.0 * 2 * 2 + 4.0 * 4)));   // 2 boxes (12n^2) + blob (4n^2) + 4 legs
        for (int i = 0; i < 24; i++) {
            float side = i < 12 ? -1.0f : 1.0f; int k = i % 12;
            float x = -2.4f + 0.43f * k + (k > 5 ? 0.2f : 0.0f), z = side * 1.5f;
            int m = nextMat(), m2 = nextMat();
            B.box(V3{x, -0.05f, z}, V3{0.18f, 0.03f, 0.18f}, n, m);
            B.box(V3{x, 0.3f, z + side * 0.17f}, V3{0.18f, 0.3f, 0.02f}, n, m);
            B.blob(V3{x, 0.08f, z}, 0.12f, n, m2, 0.05f, 5.0f);
            for (int l = 0; l < 4; l++) B.cylinder(V3{x + (l & 1 ? 0.15f : -0.15f), -0.5f, z + (l & 2 ? 0.15f : -0.15f)}, 0.015f, 0.42f, std::max(3, n / 3), m);
        }
    } else if (kind == "courtyard") {
        // all six BSDF types; open sky
        std::vector<int> mats;
        const int types[6] = {FLX_BXDF_DIFFUSE, FLX_BXDF_GLOSSY, FLX_BXDF_GGX_ROUGH_REFLECTION, FLX_BXDF_IDEAL_REFLECTION, FLX_BXDF_GGX_ROUGH_DIELECTRIC, FLX_BXDF_IDEAL_DIELECTRIC};
        for (int i = 0; i < 4; i++) addTexture(makeChecker(512, 16, seed * 17u + i, "proc_court_" + std::to_string(i)));
        for (int i = 0; i < 48; i++) {
            int type = i < 24 ? FLX_BXDF_DIFFUSE : types[i % 6];
            float ni = (type == FLX_BXDF_GGX_ROUGH_DIELECTRIC || type == FLX_BXDF_IDEAL_DIELECTRIC || type == FLX_BXDF_GLOSSY) ? 1.5f : 1.0f;
            mats.push_back(addMaterial(mkMat(type, col(), V3{0.8f, 0.8f, 0.8f}, rng.range(40.0f, 2000.0f), ni, (i < 4) ? i : -1)));
        }
        int mi = 0; auto nextMat = [&]() { return mats[(mi++) % mats.size()]; };
        B.quadGrid(V3{-20, 0, -20}, V3{40, 0, 0}, V3{0, 0, 40}, 256, 256, mats[0], 20.0f);
        for (int k = 0; k < 4; k++) {                      // arcade walls
            float a = (float)k * 1.5707963f; V3 d{std::cos(a), 0, std::sin(a)}, t{-d.z, 0, d.x};
            B.quadGrid(d * 20.0f - t * 20.0f, t * 40.0f, V3{0, 8, 0}, 192, 48, nextMat());
            for (int c = 0; c < 12; c++) B.cylinder(d * 17.0f + t * (-16.5f + 3.0f * c), 0.35f, 5.0f, 40, nextMat());
        }
        size_t base = triangles.size();
        uint32_t remaining = targetTris > base ? targetTris - (uint32_t)base : 64u * 1000u;
        const int trees = 64;
        uint32_t perTree = remaining / trees;
        // tree = trunk (4n^2+...) + 9 foliage blobs (4n^2 each)
        int n = std::max(4, (int)std::sqrt((double)perTree / 41.0));
        for (int i = 0; i < trees; i++) {
            float x = -14.0f + 4.0f * (i % 8) + rng.range(-0.8f, 0.8f), z = -14.0f + 4.0f * (i / 8) + rng.range(-0.8f, 0.8f);
            float h = rng.range(2.0f, 3.5f);
            B.cylinder(V3{x, 0, z}, 0.18f, h, n / 2 + 2, nextMat());
            int leaf = nextMat();
            for (int b = 0; b < 9; b++) {
                V3 c{x + rng.range(-0.9f, 0.9f), h + rng.range(-0.2f, 1.2f), z + rng.range(-0.9f, 0.9f)};
                B.blob(c, rng.range(0.5f, 0.9f), n, leaf, 0.15f, 9.0f);
            }
        }
    } else {
        throw std::runtime_error("Scene::generate: unknown kind " + kind);
    }
}

} // namespace fluctus
