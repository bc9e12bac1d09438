"""numpy views of the wire structs in include/fluctus_wire.h (sizes asserted against the header's)."""
import numpy as np

VEC3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])
VERTEX = np.dtype([("p", VEC3), ("n", VEC3), ("t", VEC3)])
TRIANGLE = np.dtype([("v0", VERTEX), ("v1", VERTEX), ("v2", VERTEX), ("matId", "<i4"), ("_pad", "<i4", 3)])
NODE = np.dtype([("bmin", VEC3), ("bmax", VEC3), ("parent", "<i4"), ("iStartOrRight", "<u4"),
                 ("nPrims", "u1"), ("_pad", "u1", 7)])
MATERIAL = np.dtype([("Kd", VEC3), ("Ks", VEC3), ("Ke", VEC3), ("Ns", "<f4"), ("Ni", "<f4"),
                     ("map_Kd", "<i4"), ("map_Ks", "<i4"), ("map_N", "<i4"), ("type", "<i4"), ("_pad", "<i4", 2)])
TEXDESC = np.dtype([("offset", "<u4"), ("width", "<u4"), ("height", "<u4")])
AREALIGHT = np.dtype([("right", VEC3), ("up", VEC3), ("N", VEC3), ("pos", VEC3), ("E", VEC3),
                      ("size", "<f4", 2), ("_pad", "<f4", 2)])
CAMERA = np.dtype([("pos", VEC3), ("dir", VEC3), ("up", VEC3), ("right", VEC3), ("fov", "<f4"),
                   ("apertureSize", "<f4"), ("focalDist", "<f4"), ("_pad", "<f4")])
RENDER_PARAMS = np.dtype([("areaLight", AREALIGHT), ("camera", CAMERA), ("exposure", "<f4"), ("tmOperator", "<u4"),
                          ("width", "<u4"), ("height", "<u4"), ("n_tris", "<u4"), ("useEnvMap", "<u4"),
                          ("useAreaLight", "<u4"), ("envMapStrength", "<f4"), ("maxBounces", "<u4"),
                          ("sampleImpl", "<u4"), ("sampleExpl", "<u4"), ("useRoulette", "<u4"),
                          ("wfSeparateQueues", "<u4"), ("worldRadius", "<f4"), ("_pad", "<u4", 2)])
COUNTERS = np.dtype([("raygenQueue", "<u4"), ("extensionQueue", "<u4"), ("shadowQueue", "<u4"),
                     ("diffuseQueue", "<u4"), ("glossyQueue", "<u4"), ("ggxReflQueue", "<u4"),
                     ("ggxRefrQueue", "<u4"), ("deltaQueue", "<u4")])

assert TRIANGLE.itemsize == 160 and NODE.itemsize == 48 and MATERIAL.itemsize == 80
assert TEXDESC.itemsize == 12 and RENDER_PARAMS.itemsize == 240 and COUNTERS.itemsize == 32
assert AREALIGHT.itemsize == 96 and CAMERA.itemsize == 80


class BXDF:
    DIFFUSE = 1 << 1
    GLOSSY = 1 << 2
    GGX_ROUGH_REFLECTION = 1 << 3
    IDEAL_REFLECTION = 1 << 4
    GGX_ROUGH_DIELECTRIC = 1 << 5
    IDEAL_DIELECTRIC = 1 << 6
    EMISSIVE = 1 << 7


class COL:
    ORIG, DIR, SHADOW_ORIG, SHADOW_DIR, T, EI, LAST_BSDF, LAST_EMISSION, LAST_T, P, N, UV = \
        0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44
    PHASE, LAST_PDF_W, PATH_LEN, SEED, LAST_SPECULAR, SHADOW_BLOCKED, BACKFACE, PIXEL_INDEX = \
        46, 47, 48, 49, 50, 51, 52, 53
    FIRST_DIFFUSE, LAST_PDF_DIRECT, LAST_PDF_IMPLICIT, LAST_COS_TH, LAST_PICK_PROB, SHADOW_LEN = \
        54, 55, 56, 57, 58, 59
    HIT_T, HIT_I, AREA_LIGHT_HIT, MAT_ID, NUM = 60, 61, 62, 63, 64


class Q:
    RAYGEN, EXTENSION, SHADOW, DIFFUSE, GLOSSY, GGX_REFL, GGX_REFR, DELTA, NUM = range(9)


def _v3(a, x, y, z, w=0.0):
    a["x"], a["y"], a["z"], a["w"] = x, y, z, w


def default_params(width, height, world_radius=1.0, n_tris=0):
    """RenderParams with the reference's start-up values
    (reference: src/tracer.cpp:38-52 resetParams, :760-797 initCamera/initPostProcessing/initAreaLight)."""
    p = np.zeros((), dtype=RENDER_PARAMS)
    al = p["areaLight"]
    _v3(al["E"], 200.0, 200.0, 200.0)
    _v3(al["right"], 0.0, 0.0, -1.0)
    _v3(al["up"], 0.0, 1.0, 0.0)
    _v3(al["N"], -1.0, 0.0, 0.0, 0.0)
    _v3(al["pos"], 1.0, 1.0, 0.0, 1.0)
    al["size"] = (0.5, 0.5)
    cam = p["camera"]
    _v3(cam["pos"], 0.0, 1.0, 3.5)
    _v3(cam["right"], 1.0, 0.0, 0.0)
    _v3(cam["up"], 0.0, 1.0, 0.0)
    _v3(cam["dir"], 0.0, 0.0, -1.0)
    cam["fov"], cam["apertureSize"], cam["focalDist"] = 60.0, 0.0, 0.5
    p["exposure"], p["tmOperator"] = 1.0, 2
    p["width"], p["height"], p["n_tris"] = width, height, n_tris
    p["useEnvMap"], p["useAreaLight"], p["envMapStrength"] = 0, 1, 1.0
    p["maxBounces"], p["sampleImpl"], p["sampleExpl"], p["useRoulette"] = 10, 1, 1, 0
    p["wfSeparateQueues"], p["worldRadius"] = 0, world_radius
    return p


def look_at(p, pos, target, fov=60.0, up=(0.0, 1.0, 0.0)):
    """Point the camera (pos/dir/up/right kept orthonormal like the reference's camera update)."""
    pos = np.asarray(pos, np.float32)
    d = np.asarray(target, np.float32) - pos
    d /= np.linalg.norm(d)
    r = np.cross(d, np.asarray(up, np.float32))
    r /= np.linalg.norm(r)
    u = np.cross(r, d)
    cam = p["camera"]
    _v3(cam["pos"], *pos)
    _v3(cam["dir"], *d)
    _v3(cam["right"], *r)
    _v3(cam["up"], *u)
    cam["fov"] = fov
    return p
